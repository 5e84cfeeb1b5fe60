#!/bin/bash
# MFMA utilisation of the kernels that use the matrix cores (k_moments: sample scatter; k_nuts_stream: the two
# skinny products of the streaming row pass): rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE.
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/mfma
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/c3 -o run -- python3 $REPO/bench.py --sites 512 --D 32 --n 500 --steps 1 --warmup 1 --cpu-sites 0 > $OUT/c3.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/stream -o run -- python3 $REPO/bench.py --sites 64 --D 128 --n 2000 --cor-input 0 --steps 1 --warmup 0 --cpu-sites 0 > $OUT/stream.log 2>&1
tail -2 $OUT/c3.log | cut -c1-200; tail -2 $OUT/stream.log | cut -c1-200; ls -R $OUT | head

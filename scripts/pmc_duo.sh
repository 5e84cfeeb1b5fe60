#!/bin/bash
# Instruction-mix counters of the sampler kernels on the A/B workload (256 C3-size sites, 3 EP iterations):
# two rocprofv3 --pmc passes (counters only, with the kernel trace).  Outputs under gpurun_out/pmc/.
R=$PWD
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace -d $R/gpurun_out/pmc/a -o run -- python3 $R/scripts/ab_duo.py > $R/gpurun_out/pmc/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -d $R/gpurun_out/pmc/b -o run -- python3 $R/scripts/ab_duo.py > $R/gpurun_out/pmc/b.log 2>&1
tail -n 2 $R/gpurun_out/pmc/a.log $R/gpurun_out/pmc/b.log

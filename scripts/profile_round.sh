#!/bin/bash
# Runs on the GPU box (via gpurun): bench lines, rocprofv3 kernel traces and the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs as MI355X_MICROARCH.md prescribes) for the three
# workloads quoted in DESIGN.md section 6.  Outputs under gpurun_out/round/; summarise locally with
# scripts/profile_summarise.py.
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/round
WHICH=${ONLY:-c2 c3 stream}      # ONLY="stream" re-profiles one workload (outputs of the others are kept locally)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() {   # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d $OUT/${name}_trace -o run -- python3 $REPO/bench.py "$@" --cpu-sites 0 --no-secondary > $OUT/${name}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${name}_fetch -o run -- python3 $REPO/bench.py "$@" --cpu-sites 0 --no-secondary > $OUT/${name}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${name}_write -o run -- python3 $REPO/bench.py "$@" --cpu-sites 0 --no-secondary > $OUT/${name}_write.log 2>&1
}
for w in $WHICH; do
  rm -rf $OUT/${w}_*
  case $w in
  c2)
    python3 $REPO/bench.py --config c2 > $OUT/c2_bench.json 2> $OUT/c2_bench.err
    prof c2 --config c2 --steps 3 --warmup 1 ;;
  c3)
    # the driver's command, for the bench line AND for every profiling pass: the same launches are timed and counted
    python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/c3_bench.json 2> $OUT/c3_bench.err
    prof c3 --gpus 1 --steps 20 --warmup 5
    # instruction mix of the sampler over the same command (SQ counters, two passes of at most 8)
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -d $OUT/c3_mix_a -o run -- python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sites 0 --no-secondary > $OUT/c3_mix_a.log 2>&1
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $OUT/c3_mix_b -o run -- python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sites 0 --no-secondary > $OUT/c3_mix_b.log 2>&1 ;;
  stream)
    python3 $REPO/bench.py --config c5shard --cpu-sites 0 > $OUT/stream_bench.json 2> $OUT/stream_bench.err
    prof stream --config c5shard ;;
  esac
done
ls -R $OUT | head -60

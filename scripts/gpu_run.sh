#!/bin/bash
# The ONE runner for ad-hoc jobs on the GPU box (replaces the round-5 scratch scripts scripts/r5/run*.sh):
#   gpurun --timeout 1200 -- 'bash scripts/gpu_run.sh <tag> <command ...>'
# runs <command> from the repository root with the profiler-friendly environment, and leaves
# gpurun_out/<tag>/{cmd.txt,out.txt,err.txt,rc.txt} behind (gpurun merges gpurun_out/ back into the container).
set -u
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "$*" > "$out/cmd.txt"
"$@" > "$out/out.txt" 2> "$out/err.txt"
rc=$?
echo $rc > "$out/rc.txt"
tail -n 5 "$out/out.txt"
exit $rc

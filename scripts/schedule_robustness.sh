#!/bin/bash
# Robustness check of the kernels' hand-written assembly against the compiler's schedule (round 5: this is how the missing
# wait states in front of logistic_pair_lean's inline assembly were found): build EVERY kernel translation unit under another
# machine scheduler into variants/libepx_sched_<name>.so, then run the parity tests against that library on a GPU box:
#   scripts/schedule_robustness.sh minreg -mllvm -amdgpu-sched-strategy=iterative-minreg          (here: builds)
#   EPX_LIB=$PWD/variants/libepx_sched_minreg.so python -m pytest tests -m gpu -q \
#       --deselect tests/test_gpu_parity.py::test_native_library_is_loaded                          (on the box)
# Results must not depend on the schedule: every draw-by-draw test passes under iterative-minreg and max-ilp.
set -e
cd "$(dirname "$0")/../ep-stan_amd/csrc"
name=$1; shift
mkdir -p build_var ../../variants
objs=""
for tu in dense nuts nuts_duo nuts_stream; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c $tu.hip -o build_var/${tu}_sched_$name.o &
  objs="$objs build_var/${tu}_sched_$name.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libepx_sched_$name.so $objs build/epx_api.o build/epx_comm.o -ldl
echo built variants/libepx_sched_$name.so

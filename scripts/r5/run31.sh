cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
for v in sminreg smaxilp; do
EPX_LIB=$PWD/variants/libepx_$v.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round5.py -m gpu -q --deselect tests/test_gpu_parity.py::test_native_library_is_loaded > gpurun_out/r5/test_$v.log 2>&1; echo "tests ($v: nuts.hip, nuts_stream.hip, dense.hip under the other scheduler) rc=$?"; tail -4 gpurun_out/r5/test_$v.log; grep -n "^FAILED" gpurun_out/r5/test_$v.log | head -20
done

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
export EPX_LIB=$PWD/variants/libepx_minreg.so
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q --deselect tests/test_gpu_parity.py::test_native_library_is_loaded > gpurun_out/r5/test_minreg.log 2>&1; echo "tests (minreg lib) rc=$?"; tail -25 gpurun_out/r5/test_minreg.log
timeout 300 python bench.py --steps 2 --warmup 1 --no-secondary --cpu-sites 0 --parity-sites 0 > /tmp/o.json 2>gpurun_out/r5/minreg_bench.err; echo "bench rc=$?"; tail -5 gpurun_out/r5/minreg_bench.err

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
EPX_LIB=$PWD/variants/libepx_sched_minreg.so timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_native_library_is_loaded --deselect tests/test_gpu_zz_c5_full.py > gpurun_out/r5/test_sched_minreg_all.log 2>&1; echo "whole GPU suite under iterative-minreg (all kernel TUs) rc=$?"; tail -4 gpurun_out/r5/test_sched_minreg_all.log; grep -n "^FAILED" gpurun_out/r5/test_sched_minreg_all.log | head

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -q > gpurun_out/r5/test_round5b.log 2>&1; echo "round5 tests rc=$?"; tail -30 gpurun_out/r5/test_round5b.log

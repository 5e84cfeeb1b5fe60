cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_round5.py -q -m gpu > gpurun_out/r5/test_round5c.log 2>&1; echo "round5 tests rc=$?"; grep -n "^E \|passed\|failed" gpurun_out/r5/test_round5c.log | head -60

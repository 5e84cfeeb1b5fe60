cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -m gpu > gpurun_out/r5/test_round5.log 2>&1; echo "round5 tests rc=$?"; tail -30 gpurun_out/r5/test_round5.log
bash scripts/r5/run3.sh

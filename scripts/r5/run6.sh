cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 600 python scripts/r5/explore_tight.py > gpurun_out/r5/explore_tight.txt 2>&1; cat gpurun_out/r5/explore_tight.txt | tail -12
timeout 900 python -m pytest tests/test_gpu_round5.py -q -m gpu -k "teacher or trace_follows" > gpurun_out/r5/test_round5b.log 2>&1; echo "round5 tests rc=$?"; grep -n "^E \|passed\|failed" gpurun_out/r5/test_round5b.log | head -40

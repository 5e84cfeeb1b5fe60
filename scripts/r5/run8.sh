cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -5 gpurun_out/profile_round.log
ls gpurun_out/round | head -40
cat gpurun_out/round/c3_bench.json | head -c 600

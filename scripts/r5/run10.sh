cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
bash scripts/profile_round.sh > gpurun_out/r5/profile_round.log 2>&1
PROFILES_OUT=$PWD/gpurun_out/prof_r05 python scripts/profile_summarise.py r05 > gpurun_out/r5/profile_summarise.log 2>&1
tail -20 gpurun_out/r5/profile_summarise.log
rm -rf gpurun_out/round
ls -la gpurun_out/prof_r05
du -sh gpurun_out

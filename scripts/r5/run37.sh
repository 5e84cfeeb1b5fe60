cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
ONLY=stream bash scripts/profile_round.sh > gpurun_out/r5/profile_round_streamb.log 2>&1
PROFILES_OUT=$PWD/gpurun_out/prof_r05d python scripts/profile_summarise.py r05 > gpurun_out/r5/profile_summarise_streamb.log 2>&1
tail -5 gpurun_out/r5/profile_summarise_streamb.log
rm -rf gpurun_out/round
ls -la gpurun_out/prof_r05d

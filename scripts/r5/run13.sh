cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
ONLY=c3 bash scripts/profile_round.sh > gpurun_out/r5/profile_round_c3.log 2>&1
PROFILES_OUT=$PWD/gpurun_out/prof_r05 python scripts/profile_summarise.py r05 > gpurun_out/r5/profile_summarise_c3.log 2>&1
tail -5 gpurun_out/r5/profile_summarise_c3.log
rm -rf gpurun_out/round
ls gpurun_out/prof_r05
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu -k "piece or round3 or round4 or round5 or litmus" > gpurun_out/r5/test_pieces.log 2>&1; echo "piece tests rc=$?"; tail -4 gpurun_out/r5/test_pieces.log

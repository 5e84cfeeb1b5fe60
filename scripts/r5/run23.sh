cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 900 python scripts/r5/lockstep_sim.py 12 > gpurun_out/r5/lockstep_sim.txt 2>&1; echo rc=$?; tail -12 gpurun_out/r5/lockstep_sim.txt

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -x -q > gpurun_out/r5/test_team_passes.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r5/test_team_passes.log
for rep in 1 2; do
timeout 600 python bench.py --steps 8 --warmup 3 --no-secondary --cpu-sites 0 --parity-sites 0 > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
python -c "
import json; j=json.load(open('/tmp/o.json')); r=j['roofline']; print('c3 with the team-pass counter rep $rep: %.2f site-updates/s, frac %.4f, launch %.1f ms, %.3f ns per gradient, pass %.0f cycles; team passes %.4g, team pass %.0f cycles, yields %.4f, chains per team pass %.3f' % (j['value'], r['frac'], r['launch_ms'], r['ns_per_gradient'], r['pass_cycles'], r['team_passes_per_launch'], r['team_pass_cycles'], r['passes_lost_to_yields_share'], r['chains_per_team_pass']))" | tee -a gpurun_out/r5/team_passes_bench.txt
done

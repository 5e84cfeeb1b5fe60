cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
( for rep in 1 2; do for y in 8500 0 6000 7000 10000 12000 16000; do
  export EPX_YIELD=$y
  timeout 600 python bench.py --steps 12 --warmup 5 --no-secondary --cpu-sites 0 --parity-sites 0 > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
  python -c "
import json; j=json.load(open('/tmp/o.json')); r=j['roofline']; print('c3 EPX_YIELD=$y rep $rep: %.2f site-updates/s, frac %.4f, launch %.1f ms, %.3f ns per gradient, team pass %.0f cycles, chains per team pass %.3f, yields %.4f' % (j['value'], r['frac'], r['launch_ms'], r['ns_per_gradient'], r['team_pass_cycles'], r['chains_per_team_pass'], r['passes_lost_to_yields_share']))"
done; done ) > gpurun_out/r5/c3_yield_sweep.txt 2>&1
cat gpurun_out/r5/c3_yield_sweep.txt

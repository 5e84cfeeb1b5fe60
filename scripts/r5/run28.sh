cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
( for rep in 1 2; do for lib in default maxilp minreg relaxed nopost; do
  if [ $lib = default ]; then unset EPX_LIB; else export EPX_LIB=$PWD/variants/libepx_$lib.so; fi
  timeout 600 python bench.py --steps 12 --warmup 5 --no-secondary --cpu-sites 0 --parity-sites 0 > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
  python -c "
import json; j=json.load(open('/tmp/o.json')); r=j['roofline']; print('c3 scheduler variant $lib rep $rep: %.2f site-updates/s, frac %.4f, launch %.1f ms, %.3f ns per gradient, team pass %.0f cycles' % (j['value'], r['frac'], r['launch_ms'], r['ns_per_gradient'], r['team_pass_cycles']))"
  timeout 600 python bench.py --config c2 --steps 10 --warmup 3 --no-secondary --cpu-sites 0 --parity-sites 0 > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
  python -c "
import json; j=json.load(open('/tmp/o.json')); r=j['roofline']; print('c2 scheduler variant $lib rep $rep: %.2f site-updates/s, launch %.1f ms, %.3f us per leapfrog of the slowest chain' % (j['value'], r['launch_ms'], j['launch_tail']['last_launch_us_per_leapfrog_of_the_slowest_chain']))"
done; done ) > gpurun_out/r5/scheduler_variants_ab.txt 2>&1
cat gpurun_out/r5/scheduler_variants_ab.txt

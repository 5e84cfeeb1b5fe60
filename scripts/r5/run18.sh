cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
EPX_LIB=$PWD/variants/libepx_prefetch.so timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5/test_prefetch.log 2>&1; echo "tests (prefetch lib) rc=$?"; tail -4 gpurun_out/r5/test_prefetch.log
( for rep in 1 2; do for lib in default prefetch; do
  if [ $lib = prefetch ]; then export EPX_LIB=$PWD/variants/libepx_prefetch.so; else unset EPX_LIB; fi
  timeout 600 python bench.py --steps 8 --warmup 3 --no-secondary --cpu-sites 0 --parity-sites 0 > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
  python -c "
import json; j=json.load(open('/tmp/o.json')); r=j['roofline']; print('c3 stack prefetch $lib rep $rep: %.2f site-updates/s, frac %.4f, launch %.1f ms, %.3f ns per gradient, pass %.0f cycles' % (j['value'], r['frac'], r['launch_ms'], r['ns_per_gradient'], r['pass_cycles']))"
done; done ) > gpurun_out/r5/stack_prefetch_ab.txt 2>&1
cat gpurun_out/r5/stack_prefetch_ab.txt

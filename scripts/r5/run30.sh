cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
EPX_LIB=$PWD/variants/libepx_minreg.so timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q > gpurun_out/r5/test_minreg2.log 2>&1; echo "tests (minreg lib, hazard fixed) rc=$?"; tail -8 gpurun_out/r5/test_minreg2.log
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_parity.py -m gpu -q > gpurun_out/r5/test_default_hazfix.log 2>&1; echo "tests (default lib, hazard fixed) rc=$?"; tail -3 gpurun_out/r5/test_default_hazfix.log
for rep in 1 2; do
timeout 600 python bench.py --steps 12 --warmup 5 --no-secondary --cpu-sites 0 --parity-sites 0 > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
python -c "
import json; j=json.load(open('/tmp/o.json')); r=j['roofline']; print('c3 with the s_nop in the logistic clamp rep $rep: %.2f site-updates/s, frac %.4f, launch %.1f ms, %.3f ns per gradient, team pass %.0f cycles' % (j['value'], r['frac'], r['launch_ms'], r['ns_per_gradient'], r['team_pass_cycles']))"
done

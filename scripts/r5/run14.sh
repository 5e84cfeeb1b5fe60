cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r5/smoke.log
SECONDS=0
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r5/test_gpu_final.log 2>&1; echo "all gpu tests rc=$? wall=${SECONDS}s"; tail -4 gpurun_out/r5/test_gpu_final.log; grep -n "C5 at J" gpurun_out/r5/test_gpu_final.log | cut -c1-400
SECONDS=0
timeout 1200 python bench.py > gpurun_out/r5/bench_final.json 2> gpurun_out/r5/bench_final.err; echo "bench rc=$? wall=${SECONDS}s"
python - <<'PY'
import json
j=json.load(open('gpurun_out/r5/bench_final.json'))
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('ns_per_gradient'), j['roofline'].get('pass_cycles'), j['roofline']['traffic_source'])
print('cpu', j['cpu_baseline']['value'], 'parity ok', j['parity']['ok'])
for s in j.get('secondary', []):
    print(s['config']['name'], s['value'], s['roofline']['frac'], s['roofline']['launch_ms'])
PY

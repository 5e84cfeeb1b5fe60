cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
SECONDS=0
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r5/test_gpu_all5.log 2>&1; echo "all gpu tests rc=$? wall=${SECONDS}s"; tail -12 gpurun_out/r5/test_gpu_all5.log; grep -n "C5 at J" gpurun_out/r5/test_gpu_all5.log
SECONDS=0
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5/bench_full5.json 2> gpurun_out/r5/bench_full5.err; echo "bench rc=$? wall=${SECONDS}s"
python - <<'PY'
import json
j=json.load(open('gpurun_out/r5/bench_full5.json'))
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('ns_per_gradient'), j['roofline'].get('pass_cycles'))
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['strict_build']['site_updates_per_s'])
print('parity ok', j['parity']['ok'], json.dumps(j['parity']['transition_by_transition'])[:900])
for s in j.get('secondary', []):
    print(s['config']['name'], s['value'], s['roofline']['frac'], s['roofline']['launch_ms'])
PY
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5/smoke5.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r5/smoke5.log

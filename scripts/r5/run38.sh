cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
SECONDS=0
timeout 1200 python bench.py > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/bench_default.err; echo "bench (no flags) rc=$? wall=${SECONDS}s"
python - <<'PY'
import json
j=json.load(open('gpurun_out/r5/bench_default.json'))
r=j['roofline']
print(j['metric'], j['value'], j['unit'], j['n_gpus'], j['steps'], j['warmup'], j['ms_per_step'], r['frac'], j['config'])
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['cores'], j['cpu_baseline']['sample'][:120], 'parity', j['parity']['ok'])
for s in j.get('secondary', []):
    print(s['config']['name'], s['value'], s['roofline']['frac'])
PY

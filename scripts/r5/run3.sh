cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
SECONDS=0; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5/bench_full.json 2> gpurun_out/r5/bench_full.err
echo rc=$? wall=${SECONDS}s
tail -5 gpurun_out/r5/bench_full.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r5/bench_full.json'))
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('ns_per_gradient'), j['roofline'].get('pass_cycles'))
print(json.dumps(j.get('cpu_baseline'))[:1500])
print('parity ok', j['parity']['ok'], j['parity']['whole_update'])
for s in j.get('secondary', []):
    print(json.dumps(s)[:1200])
PY

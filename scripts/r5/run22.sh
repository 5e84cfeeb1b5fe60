cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
SECONDS=0
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5/bench_final.json 2> gpurun_out/r5/bench_final.err; echo "bench rc=$? wall=${SECONDS}s"
python - <<'PY'
import json
j=json.load(open('gpurun_out/r5/bench_final.json'))
r=j['roofline']
print(j['value'], j['ms_per_step'], r['frac'], r['ns_per_gradient'], r['pass_cycles'], r['team_passes_per_launch'], r['team_pass_cycles'], r['passes_lost_to_yields_share'], r['chains_per_team_pass'])
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['strict_build']['site_updates_per_s'], 'parity', j['parity']['ok'])
for s in j.get('secondary', []):
    print(s['config']['name'], s['value'], s['roofline']['frac'], s['roofline']['launch_ms'])
PY

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_round5.py -q -m gpu > gpurun_out/r5/test_round5.log 2>&1; echo "round5 tests rc=$?"; tail -40 gpurun_out/r5/test_round5.log
for pp in 16 24 32; do
  EPX_PIECES_PER_SITE=$pp timeout 600 python bench.py --config c5shard --steps 1 --warmup 1 --cpu-sites 0 > gpurun_out/r5/c5_pp$pp.json 2> gpurun_out/r5/c5_pp$pp.err
  python -c "
import json; j=json.load(open('gpurun_out/r5/c5_pp$pp.json')); r=j['roofline']; print('pieces/site $pp', j['value'], r['achieved'], r['frac'], r['launch_ms'], j['launch_tail']['max_over_mean'])"
done
SECONDS=0
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r5/test_gpu_all.log 2>&1; echo "all gpu tests rc=$? wall=${SECONDS}s"; tail -15 gpurun_out/r5/test_gpu_all.log

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
( for rep in 1 2; do for div in 1 4; do
  EPX_PIECE_TAIL_DIV=$div timeout 600 python bench.py --steps 10 --warmup 3 --cpu-sites 0 --no-secondary > /tmp/o.json 2>/tmp/o.err
  python -c "
import json; j=json.load(open('/tmp/o.json')); r=j['roofline']; print('c3 tail_div $div rep $rep: %.2f site-updates/s, %.2f ms/step, frac %.4f, ns/grad %.4f' % (j['value'], j['ms_per_step'], r['frac'], r['ns_per_gradient']))"
done; done
for rep in 1 2; do for div in 1 4; do
  EPX_PIECE_TAIL_DIV=$div timeout 600 python bench.py --config c5shard --steps 1 --warmup 1 --cpu-sites 0 > /tmp/o.json 2>/tmp/o.err
  python -c "
import json; j=json.load(open('/tmp/o.json')); r=j['roofline']; print('c5shard tail_div $div rep $rep: %.3f site-updates/s, %.1f GB/s, frac %.4f, launch %.0f ms' % (j['value'], r['achieved'], r['frac'], r['launch_ms']))"
done; done ) > gpurun_out/r5/piece_tail_ab.txt 2>&1
cat gpurun_out/r5/piece_tail_ab.txt

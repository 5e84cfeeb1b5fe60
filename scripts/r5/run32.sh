cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
( for rep in 1 2; do for lib in default stream_maxilp stream_minreg; do
  if [ $lib = default ]; then unset EPX_LIB; else export EPX_LIB=$PWD/variants/libepx_$lib.so; fi
  timeout 600 python bench.py --config c5shard --steps 1 --warmup 1 --cpu-sites 0 > /tmp/o.json 2>/tmp/o.err
  python -c "
import json; j=json.load(open('/tmp/o.json')); r=j['roofline']; print('c5shard nuts_stream.hip scheduler $lib rep $rep: %.3f site-updates/s, %.1f GB/s, frac %.4f, launch %.0f ms, %.1f us per pass and CU' % (j['value'], r['achieved'], r['frac'], r['launch_ms'], r['ns_per_row_pass_per_cu']/1e3))"
done; done ) > gpurun_out/r5/stream_scheduler_ab.txt 2>&1
cat gpurun_out/r5/stream_scheduler_ab.txt

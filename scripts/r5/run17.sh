cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests -m gpu -x -q -k "stream or c5 or multigroup or layout3 or 3" > gpurun_out/r5/test_stream_shared.log 2>&1; echo "stream tests rc=$?"; tail -5 gpurun_out/r5/test_stream_shared.log
( for rep in 1 2; do for lib in loader shared; do
  if [ $lib = loader ]; then export EPX_LIB=$PWD/variants/libepx_loader.so; else unset EPX_LIB; fi
  timeout 600 python bench.py --config c5shard --steps 1 --warmup 1 --cpu-sites 0 > /tmp/o.json 2>/tmp/o.err
  python -c "
import json; j=json.load(open('/tmp/o.json')); r=j['roofline']; print('c5shard row DMA issued by $lib rep $rep: %.3f site-updates/s, %.1f GB/s, frac %.4f, launch %.0f ms, %.1f us per pass and CU' % (j['value'], r['achieved'], r['frac'], r['launch_ms'], r['ns_per_row_pass_per_cu']/1e3))"
done; done ) > gpurun_out/r5/stream_shared_issue_ab.txt 2>&1
cat gpurun_out/r5/stream_shared_issue_ab.txt

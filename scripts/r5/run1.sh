cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
(
for args in "256 20000 100 7" "256 20000 100 1" "256 20000 100 3" "256 20000 150 7" "256 20000 200 7" "1 20000 100 7" "1 20000 100 1"; do
  echo "== quad_pass $args"; timeout 120 ./variants/quad_pass $args
done
) > gpurun_out/r5/quad_probe.txt 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-sites 0 > gpurun_out/r5/bench_base.json 2> gpurun_out/r5/bench_base.err
tail -c 600 gpurun_out/r5/bench_base.json
cat gpurun_out/r5/quad_probe.txt

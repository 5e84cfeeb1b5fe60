cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
(
for bin in team8_pass team8_pass_l2; do
for args in "256 20000 100 0" "256 20000 100 1" "256 20000 200 0" "256 20000 300 0" "256 20000 50 0" "1 20000 100 0"; do
  echo "== $bin $args"; timeout 120 ./variants/$bin $args
done
done
) > gpurun_out/r5/team8_probe.txt 2>&1
cat gpurun_out/r5/team8_probe.txt

#!/bin/bash
# Same-box, order-controlled A/B of library builds on the C5 shard (bench.py --config c5shard, 1 timed EP iteration after 1):
#   bash scripts/ab_c5.sh <out file> <lib> <lib> ...      (the libraries in the order given)
out=$1; shift
mkdir -p "$(dirname "$out")"
: > "$out"
for lib in "$@"; do
  EPX_LIB=$PWD/$lib python3 bench.py --config c5shard --steps ${AB_STEPS:-1} --warmup ${AB_WARMUP:-1} --cpu-sites 0 ${AB_EXTRA:-} 2>>"$out.err" | python3 -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = o['roofline']
print('%-36s %6.2f site-updates/s  %7.0f ms/step  frac %.4f  %6.1f us per row pass and CU  passes %.4g' % ('$lib', o['value'], o['ms_per_step'], r['frac'], r['ns_per_row_pass_per_cu'] / 1e3, r['row_passes_per_launch']))" >> "$out"
done
cat "$out"

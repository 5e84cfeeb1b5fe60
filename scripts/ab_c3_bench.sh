#!/bin/bash
# Same-box, order-controlled A/B of library builds on the headline workload (C3: bench.py, 6 timed EP iterations after 3):
#   bash scripts/ab_c3_bench.sh <out file> <lib> <lib> ...      (the libraries in the order given)
out=$1; shift
mkdir -p "$(dirname "$out")"
: > "$out"
for lib in "$@"; do
  EPX_LIB=$PWD/$lib python3 bench.py --steps ${AB_STEPS:-6} --warmup ${AB_WARMUP:-3} --cpu-sites 0 --no-secondary 2>>"$out.err" | python3 -c "
import sys, json
o = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = o['roofline']
print('%-36s %7.2f site-updates/s  frac %.4f  ns/gradient %.3f  team_pass_cycles %.0f  pass_cycles %.0f  chains/pass %.3f  yields %.4f' % ('$lib', o['value'], r['frac'], r['ns_per_gradient'], r['team_pass_cycles'] or 0, r['pass_cycles'] or 0, r['chains_per_team_pass'] or 0, r['passes_lost_to_yields_share'] or 0))" >> "$out"
done
cat "$out"

#!/bin/bash
# Same-box, order-controlled A/B of streaming-sampler builds at the C5 site size (256 sites = one per CU, short runs):
#   bash scripts/ab_stream.sh <out file> <lib> <lib> ...     -> every library in turn, the whole list twice
out=$1; shift
mkdir -p "$(dirname "$out")"
: > "$out"
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib (round $rep)" >> "$out"
    EPX_LIB=$PWD/$lib python3 scripts/tick_time_stream.py 256 10 30 >> "$out" 2>&1
  done
done
cat "$out"

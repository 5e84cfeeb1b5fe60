#!/bin/bash
# Builds libepx.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-parameter"
mkdir -p build
pids=()
for f in dense nuts nuts_duo nuts_stream epx_api epx_comm; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ epx_kernels.h -nt build/$f.o ] || [ epx_device.h -nt build/$f.o ] || [ epx_ctx.h -nt build/$f.o ] || [ nuts_common.h -nt build/$f.o ] || [ nuts_state_machine.inc -nt build/$f.o ] || [ nuts_gradient.inc -nt build/$f.o ] || [ nuts_gradient_groups.inc -nt build/$f.o ] || [ epx_stream_tile.h -nt build/$f.o ] || [ epx_pieces.h -nt build/$f.o ] || [ ../../include/epx.h -nt build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libepx.so build/dense.o build/nuts.o build/nuts_duo.o build/nuts_stream.o build/epx_api.o build/epx_comm.o -ldl
echo "built $(cd .. && pwd)/libepx.so"
# The same library with the piece hand-off's release FENCE kept (-DEPX_PIECE_FENCE, epx_pieces.h): the A/B partner of the
# litmus test (tests/test_gpu_round4.py), never benchmarked.  Only the two files that include epx_pieces.h differ.
mkdir -p build_fence ../../variants
pids=()
for f in nuts_duo nuts_stream; do
  if [ ! -f build_fence/$f.o ] || [ build/$f.o -nt build_fence/$f.o ]; then
    $HIPCC $FLAGS -DEPX_PIECE_FENCE -c $f.hip -o build_fence/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
if [ ! -f ../../variants/libepx_fence.so ] || [ ../libepx.so -nt ../../variants/libepx_fence.so ]; then
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../variants/libepx_fence.so build/dense.o build/nuts.o build_fence/nuts_duo.o build_fence/nuts_stream.o build/epx_api.o build/epx_comm.o -ldl
fi
echo "built $(cd ../.. && pwd)/variants/libepx_fence.so"
if [ "$EPX_STAMPS" = "1" ]; then
  # diagnostic variant with in-kernel cycle stamps (scripts/stamps.py); never benchmarked
  mkdir -p build_stamps
  for f in dense nuts nuts_duo nuts_stream epx_api epx_comm; do $HIPCC $FLAGS -DEPX_STAMPS -c $f.hip -o build_stamps/$f.o & done; wait
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../variants/libepx_stamps.so build_stamps/dense.o build_stamps/nuts.o build_stamps/nuts_duo.o build_stamps/nuts_stream.o build_stamps/epx_api.o build_stamps/epx_comm.o -ldl
  echo "built diagnostic variants/libepx_stamps.so"
fi

#!/bin/bash
# A/B helper: variants/libepx_<name>.so = the library with nuts_stream.hip recompiled under extra flags, the C5 shape only
# (-DEPX_STREAM_MIN: NV = 7, DPB = 128; seconds instead of minutes).  The other objects come from csrc/build.
# usage: scripts/build_stream_variant.sh <name> <extra hipcc flags...>
set -e
cd "$(dirname "$0")/../ep-stan_amd/csrc"
name=$1; shift
mkdir -p build_var ../../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DEPX_STREAM_MIN "$@" -c nuts_stream.hip -o build_var/nuts_stream_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libepx_$name.so build/dense.o build/nuts.o build/nuts_duo.o build_var/nuts_stream_$name.o build/epx_api.o build/epx_comm.o -ldl
echo built variants/libepx_$name.so

#!/bin/bash
# A/B helper: variants/libepx_<name>.so = the library with nuts_duo.hip recompiled under extra flags.
# usage: scripts/build_variant.sh <name> <extra hipcc flags...>      (the other objects come from csrc/build)
set -e
cd "$(dirname "$0")/../ep-stan_amd/csrc"
name=$1; shift
mkdir -p build_var ../../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c nuts_duo.hip -o build_var/nuts_duo_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libepx_$name.so build/dense.o build/nuts.o build_var/nuts_duo_$name.o build/nuts_stream.o build/epx_api.o build/epx_comm.o -ldl
echo built variants/libepx_$name.so

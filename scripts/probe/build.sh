#!/bin/bash
# Builds the stand-alone probes of the row streaming engine into variants/ (git-ignored).
set -e
# (stream_probe.hip follows an older interface of epx_stream_tile.h and no longer builds: kept for its round-3 numbers)
cd "$(dirname "$0")/../.."
mkdir -p variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iep-stan_amd/csrc scripts/probe/stream_probe.hip -o variants/stream_probe 2>/dev/null || echo 'stream_probe: not built (stale)'
hipcc --offload-arch=gfx950 -O2 scripts/probe/mfma_layout.hip -o variants/mfma_layout
hipcc --offload-arch=gfx950 -O2 scripts/probe/concurrent.hip -o variants/concurrent
hipcc --offload-arch=gfx950 -O2 scripts/probe/xcu_latency.hip -o variants/xcu_latency
echo "built variants/stream_probe variants/mfma_layout variants/concurrent variants/xcu_latency"
hipcc --offload-arch=gfx950 -O2 scripts/probe/mfma_rate.hip -o variants/mfma_rate
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iep-stan_amd/csrc scripts/probe/team_pass.hip -o variants/team_pass
hipcc --offload-arch=gfx950 -O2 scripts/probe/dispatch_gaps.hip -o variants/dispatch_gaps
hipcc --offload-arch=gfx950 -O2 scripts/probe/lds_rate.hip -o variants/lds_rate
# round 5: the two gate probes of a restructured C3 pass (both measured ABOVE layout 7's 9 560 cycles: profiles/r05_probe_*.txt)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iep-stan_amd/csrc -Iscripts/probe scripts/probe/quad_pass.hip -o variants/quad_pass
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iep-stan_amd/csrc -Iscripts/probe scripts/probe/team8_pass.hip -o variants/team8_pass
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DT8_OM_L2 -Iep-stan_amd/csrc -Iscripts/probe scripts/probe/team8_pass.hip -o variants/team8_pass_l2

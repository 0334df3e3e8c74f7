#!/bin/bash
# Builds the development probes under tools/ for gfx950 (hipcc cross-compiles without a GPU).  Not part of __graft_entry__.build():
# the product is ug_stereomatcher_amd/libugsm.so; these are measurement tools.  The binaries are git-ignored and travel to the GPU box
# with the gpurun snapshot.
set -e
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Iinclude"
hipcc $F tools/kbench.hip -o tools/kbench
hipcc -O2 --offload-arch=gfx950 tools/queue_probe.hip -o tools/queue_probe
[ -f tools/valubench.hip ] && hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/valubench.hip -o tools/valubench
[ -f tools/ldsbench.hip ] && hipcc -O2 --offload-arch=gfx950 tools/ldsbench.hip -o tools/ldsbench
hipcc -O3 --offload-arch=gfx950 tools/gridbar.hip -o tools/gridbar
hipcc -O3 --offload-arch=gfx950 tools/pcie_probe.hip -o tools/pcie_probe
echo "tools built"

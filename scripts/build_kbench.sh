#!/bin/bash
# builds scripts/kbench.cpp against the in-tree library (cross-compiles here; the binary travels under build/ -- not gpurun_out/,
# which is excluded from the snapshot)
set -e
cd "$(dirname "$0")/.."
mkdir -p leann_amd/lib/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 scripts/kbench.cpp -Iinclude -Lleann_amd/lib -lleann_mi355x -lrocblas \
    -Wl,-rpath,'$ORIGIN/..' -o leann_amd/lib/bin/kbench
echo built leann_amd/lib/bin/kbench

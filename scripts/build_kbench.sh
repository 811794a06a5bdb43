#!/bin/bash
# builds scripts/kbench.cpp against the in-tree library (cross-compiles here; the binary travels under build/ -- not gpurun_out/,
# which is excluded from the snapshot)
set -e
cd "$(dirname "$0")/.."
mkdir -p leann_amd/lib/bin
# kbench runs against the DIAGNOSIS build of the library (-DLM_DIAG: stamped / phase-skipping kernel instantiations)
make -j8 -C leann_amd/csrc OUT=../lib/diag EXTRA_DEFS=-DLM_DIAG ../lib/diag/libleann_mi355x.so > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 scripts/kbench.cpp -Iinclude -Lleann_amd/lib/diag -lleann_mi355x -lrocblas \
    -Wl,-rpath,'$ORIGIN/../diag' -o leann_amd/lib/bin/kbench
echo built leann_amd/lib/bin/kbench

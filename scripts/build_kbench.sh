#!/bin/bash
# builds scripts/kbench.cpp against the in-tree library (cross-compiles here; the binary travels under leann_amd/lib/bin -- not gpurun_out/,
# which is excluded from the snapshot).   scripts/build_kbench.sh [suffix "extra defs"]: a second diagnosis build + binary for an A/B of a
# compile-time switch, e.g.  scripts/build_kbench.sh nop0 "-DLM_DMA_NOP=0"  ->  leann_amd/lib/diag_nop0/, leann_amd/lib/bin/kbench_nop0
set -e
cd "$(dirname "$0")/.."
mkdir -p leann_amd/lib/bin
SUF="${1:+_$1}"
# kbench runs against the DIAGNOSIS build of the library (-DLM_DIAG: stamped / phase-skipping kernel instantiations)
make -j8 -C leann_amd/csrc OUT=../lib/diag$SUF EXTRA_DEFS="-DLM_DIAG ${2:-}" ../lib/diag$SUF/libleann_mi355x.so > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 scripts/kbench.cpp -Iinclude -Lleann_amd/lib/diag$SUF -lleann_mi355x -lrocblas \
    -Wl,-rpath,'$ORIGIN/../diag'$SUF -o leann_amd/lib/bin/kbench$SUF
echo built leann_amd/lib/bin/kbench$SUF

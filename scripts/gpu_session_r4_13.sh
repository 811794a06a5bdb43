#!/bin/bash
# Round 4, GPU session 13 (the round's last ~100 GPU-seconds): HBM-side traffic of the two QKV kernels (FETCH_SIZE / WRITE_SIZE passes).
set -u
cd "$(dirname "$0")/.."
timeout -k 5 110 bash scripts/pmc_tail.sh r4s13 bwqkv 2>&1 | tail -60 | cut -c1-200

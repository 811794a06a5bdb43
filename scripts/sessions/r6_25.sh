#!/bin/bash
# Round 6, GPU session 25: HBM-side traffic of the dominant kernel re-taken on the final tree (the bench line's roofline.traffic cited round 4's pass):
# scripts/pmc_tail.sh -- rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (kernel trace only), calibrated in-pass on kbench's streaming probes.
set -u
cd "$(dirname "$0")/../.."
bash scripts/pmc_tail.sh r6s25 2>&1 | tail -60

#!/bin/bash
# Round 3, GPU session 1: (a) the wave-per-query persistent search against the oracle on hardware; (b) the three opt-in pieces that
# had only run in emulation (QKV-in-tail kernel, alternating slot order, one-call forward): measure -> promote or delete.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/s1
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "wave_per_query" > gpurun_out/s1/pytest_wave.log 2>&1; echo "wave tests rc=$? $(tail -1 gpurun_out/s1/pytest_wave.log)"
grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/s1/pytest_wave.log | head -20
# (the remainder of this session was scripts/next_gpu_session.sh of round 2: kbench tailqkv / alternating order / subbatch A/B / latency A/B -- see profiles/r3_session1_optin_kernels_measured.txt)

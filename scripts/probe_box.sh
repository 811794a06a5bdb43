#!/bin/bash
# prints what the GPU box looks like (cores, memory, GPU) -- used to size tests and the CPU baseline
echo "nproc=$(nproc) affinity=$(python -c 'import os;print(len(os.sched_getaffinity(0)))') cpu.max=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
grep -m1 "model name" /proc/cpuinfo; free -g | head -2
/opt/rocm/bin/rocm-smi --showmeminfo vram 2>/dev/null | head -8

#!/bin/bash
# What to run FIRST in the next round's first GPU session (~12 GPU-minutes), in this order.  Everything below was either changed after
# the last hardware run of round 3 (GPU sessions r3-13 .. r3-15) or is a measurement that round ran out of GPU-minutes for.
#   1. `pytest -m gpu` on the tree as it is (the last full-suite run on hardware is r3-13 + r3-14 together; since then: provider close()
#      semantics, graph builder heuristic in its selection form, bench cpu_baseline.traversal_only -- Python only, CPU-tested).
#   2. the driver's bench command under rocprofv3 (reference line of the round + per-kernel table).
#   3. graph builder: selection form vs candidate scan of the neighbour-selection heuristic on a 1M x 384 table (same graph expected,
#      build seconds and torch launch counts are the question).
#   4. C5 at 500k chunks with the general-width library-side provider (round 3: 26.8 q/s over the Python provider, memo off).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/next1; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c2 -- python bench.py --gpus 1 --steps 6 --warmup 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench rc=$?"
find $OUT/prof -name "*kernel_trace.csv" -size +8M -delete; find $OUT/prof -name "*.db" -delete
timeout -k 10 200 python - > $OUT/builder_ab.json 2> $OUT/builder_ab.err <<'PY'
import json, time, numpy as np, torch
from leann_amd import gpu_graph_build as gb
torch.manual_seed(0)
c = torch.nn.functional.normalize(torch.randn(1000, 384, device="cuda"), dim=1)
x = torch.nn.functional.normalize(c[torch.randint(0, 1000, (1_000_000,), device="cuda")] + 0.15 * torch.nn.functional.normalize(torch.randn(1_000_000, 384, device="cuda"), dim=1), dim=1)
res, graphs = {}, {}
for name, fn in (("selection", gb._select_heuristic_selection), ("scan", gb._select_heuristic_scan)):
    gb._select_heuristic = fn
    torch.cuda.synchronize(); t0 = time.time()
    graphs[name] = gb.build_graph_gpu(x, "mips", M=32, ef_construction=200)
    torch.cuda.synchronize(); res[name + "_build_s"] = round(time.time() - t0, 1)
res["identical_graphs"] = all(np.array_equal(getattr(graphs["selection"], n), getattr(graphs["scan"], n)) for n in ("levels", "level_ptr", "node_offsets", "neighbors"))
print(json.dumps(res))
PY
echo "builder A/B: $(cat $OUT/builder_ab.json)"
timeout -k 10 500 python bench.py --config c5 --chunks 500000 --batch 256 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_c5_500k.json 2> $OUT/bench_c5.err; echo "c5 rc=$?"; tail -2 $OUT/bench_c5.err | cut -c1-400

"""bench.py's multi-rank control flow executed BEFORE the driver's 8-GPU run does it: two ranks of the script itself, launched the way
the driver launches them (torch.distributed.run, one rank per "GPU"), over gloo, against the thread-per-lane build of the product
library (tests/hip_emul) with a tiny general-width model and corpus (`--dry-run-emulated`, a test-only switch of bench.py).  What runs:
rank-0 index build + broadcast of X and of the graph, global_batch / PartitionedSearch with its packed all_gather, the all_reduce of
elapsed / recall, the memo-off steps, `rccl_ranks` in the line -- every `world > 1` branch.  Nothing in the line is a measurement
(data = "dry-run").  SURVEY 8(e); the same for the sharded config: scripts/bench_c4.py."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CLANG = Path("/opt/rocm/lib/llvm/bin/clang++")
sys.path.insert(0, str(ROOT / "tests" / "hip_emul"))


@pytest.fixture(scope="module")
def emul_lib(tmp_path_factory, built_libs):
    if not CLANG.exists():
        pytest.skip("needs ROCm's clang++ as a host compiler")
    import build_emul_lib

    return build_emul_lib.build(tmp_path_factory.mktemp("emul_lib_bench"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, timeout):
    env = {k: v for k, v in os.environ.items() if not k.startswith("LEANN_MI355X_")}
    env["PYTHONPATH"] = str(ROOT)
    return subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=timeout)


def _last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


def test_bench_py_control_flow_on_two_ranks(emul_lib, world=2):
    common = ["bench.py", "--gpus", str(world), "--steps", "1", "--warmup", "1", "--chunks", "200", "--batch", "2", "--M", "8", "--efc", "24", "--ef", "16",
              "--no-cpu-baseline", "--no-table-roofline", "--no-parity-check", "--no-latency-rows", "--no-min-ef-step", "--dry-run-emulated", str(emul_lib)]
    if world == 1:
        cmd = [sys.executable] + common
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + common
    r = _run(cmd, 900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = _last_json(r.stdout)
    assert line["dry_run"] is True and line["data"].startswith("dry-run")
    assert line["n_gpus"] == world and line["config"]["rccl_ranks"] == world and line["config"]["queries_per_step"] == 2 * world
    assert line["steps"] == 1 and line["value"] > 0 and line["scaling"] == "weak"
    assert 0.0 <= line["recall_at_10"] <= 1.0
    nm = line["without_call_memo"]
    assert nm["labels_identical_to_the_memo_steps"] is True and nm["value"] > 0  # all_reduce'd over the ranks
    assert line["roofline"]["bound"] in ("mfma", "hbm")
    assert r.stdout.count("\n{") + r.stdout.startswith("{") == 1  # ONE line, rank 0 only


def test_bench_c4_sharded_control_flow_on_two_ranks(emul_lib):
    """scripts/bench_c4.py (BASELINE configs[3]: sharded graph, every rank searches all queries on its shard, ONE packed all_gather +
    the lm_topk_merge kernel inside the timed region) on two gloo ranks over the emulated library."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "scripts/bench_c4.py", "--chunks", "300", "--batch", "2", "--steps", "1", "--warmup", "1", "--M", "8", "--efc", "24", "--ef", "16", "--no-cpu-baseline",
           "--no-parity-check", "--dry-run-emulated", str(emul_lib)]
    r = _run(cmd, 900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = _last_json(r.stdout)
    assert line["dry_run"] is True and line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["config"]["shard_chunks"] == 150
    assert line["value"] > 0 and line["allgather_plus_merge_us"] > 0 and line["exchange_bytes_per_rank"] == 2 * 10 * 12
    assert line["recall_at_10"] >= 0.2  # merged answer over both shards vs the exact top-10 over both shards; a random one-layer model clusters its embeddings (bit-equality of the merge itself: tests/emulated_two_rank.py)

"""The C3 / C4 bench scripts (BASELINE.json configs[2], configs[3]) at a size that runs in seconds: the JSON contract, a sane
recall, and -- for C4 -- the sharded path's merge in the timed region.  The full-size lines are recorded in DESIGN.md."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _run(script, *args):
    out = subprocess.run([sys.executable, str(ROOT / "scripts" / script), *args], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bench_c3_small():
    r = _run("bench_c3.py", "--chunks", "30000", "--batch", "128", "--steps", "2", "--warmup", "1", "--cpu-baseline-queries", "2")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r
    assert r["config"]["baseline_config"] == "c3" and r["value"] > 0
    assert r["recall_at_10"] >= 0.9  # the timed steps run at the smallest complexity of the sweep that reaches the metric's bar
    sw = r["complexity_sweep"]  # one sweep per rerank set: the final candidate list / every expanded node (upstream DiskANN's full_retset)
    assert set(sw) == {"final_list", "expanded_nodes"} and r["rerank_set"] in sw
    assert max(v["recall_at_10"] for v in sw[r["rerank_set"]].values()) >= 0.9
    assert set(r["roofline_traversal"]["us_per_launch_by_workgroup_width"]) == {"256", "512", "1024"}
    assert r["roofline_traversal"]["bound"] == "hbm" and r["roofline_traversal"]["achieved"] > 0
    assert r["roofline"]["bound"] == "mfma" and "k_layer_tail_h384" in r["roofline"]["kernel"] and 0 < r["roofline"]["share_of_timed_region"] <= 1.0
    assert r["cpu_baseline"]["value"] and r["cpu_baseline"]["value"] > 0
    pc = r["parity_check"]  # GPU vs the PQ oracles on the run's own index, queries, L, W and m
    assert pc["pq_order"]["ids_exact"] and pc["pq_order"]["distance_bits_equal"] and pc["pq_order"]["counts_equal"]
    assert pc["deferred_rerank"]["ids_exact"] and pc["deferred_rerank"]["distance_bits_equal"] and pc["deferred_rerank"]["one_provider_call_same_ids"]
    assert pc["table_rerank"]["ids_exact"] and pc["table_rerank"]["distance_bits_equal"] and pc["table_rerank"]["diskann_transcription_agrees"]


def test_bench_c4_one_shard_small():
    r = _run("bench_c4.py", "--chunks", "30000", "--batch", "64", "--steps", "2", "--warmup", "1", "--cpu-baseline-seconds", "5")
    assert r["config"]["baseline_config"] == "c4" and r["rccl_ranks"] == 1 and r["value"] > 0
    assert r["recall_at_10"] >= 0.9
    assert r["allgather_plus_merge_us"] > 0
    assert r["roofline"]["bound"] == "mfma" and r["roofline"]["achieved"] > 0 and "k_layer_tail_h384" in r["roofline"]["kernel"]
    assert r["cpu_baseline"]["value"] and r["cpu_baseline"]["value"] > 0 and r["cpu_baseline"]["kind"] == "port"
    pc = r["parity_check"]  # the shard's own index and queries against the oracle, as bench.py does it for C2
    assert pc["ids_exact"] and pc["ndis_equal"] and pc["max_abs_dist"] == 0.0 and pc["faiss_transcription_agrees"]
    assert pc["recompute"]["ids_exact"] and pc["recompute"]["same_ids_requested_every_round"] and pc["recompute"]["max_abs_dist"] == 0.0


def test_bench_table_provider_variant_small():
    """SURVEY 8(d)'s clustered-Gaussian table-provider variant (scripts/bench_table_provider.py) at a size that runs in seconds."""
    r = _run("bench_table_provider.py", "--chunks", "40000", "--centres", "100", "--batch", "256", "--steps", "2", "--warmup", "1")
    assert r["value"] > 0 and r["recall_at_10"] >= 0.9
    assert r["roofline"]["bound"] == "hbm" and r["roofline"]["achieved"] > 0 and r["per_query"]["provider_rows"] > 0

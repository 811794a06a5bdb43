#!/usr/bin/env python
"""A/B of the encoder forward (MiniLM-L6 shape, fp16, packed sequences, 16384 synthetic chunks) with the opt-in
kernels switched on one at a time and all together.  Prints one JSON object; run on the GPU box:
    python scripts/encoder_ops_bench.py
Switches: LEANN_MI355X_ATTN=2 (lm_attn_v2.hip), LEANN_MI355X_LN=2, LEANN_MI355X_POOL=1, LEANN_MI355X_EMBED=1,
LEANN_MI355X_MLP=1 (lm_mlp_fused.hip; LEANN_MI355X_MLP_VARIANT=2 = cross-slab pipelining), LEANN_MI355X_LINEAR=1
(lm_linear_h384.hip)."""
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from leann_amd.encoder import BertEncoder, config_for
from leann_amd.synth import CorpusSpec, SyntheticCorpus, pad_batch

SWITCHES = {"LEANN_MI355X_ATTN": "2", "LEANN_MI355X_LN": "2", "LEANN_MI355X_POOL": "1", "LEANN_MI355X_EMBED": "1",
            "LEANN_MI355X_MLP": "1", "LEANN_MI355X_LINEAR": "1", "LEANN_MI355X_PACK": "1"}
dev = torch.device("cuda")
cfg = config_for("all-MiniLM-L6-v2")
enc = BertEncoder.random_init(cfg, 0).to(dev, dtype=torch.float16)
ids, lens = pad_batch(*SyntheticCorpus(CorpusSpec(n_chunks=16384)).chunks(), 256)
ti, tl = torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)
flops = sum(cfg.flops_per_chunk(int(t)) for t in lens)


def clear():
    for k in list(SWITCHES) + ["LEANN_MI355X_MLP_VARIANT"]:
        os.environ.pop(k, None)


def timed(label):
    try:
        e = enc.encode_tokens_packed(ti, tl, 262144)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            e = enc.encode_tokens_packed(ti, tl, 262144)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        return e, {"ms": round(dt * 1e3, 2), "chunks_per_s": round(len(lens) / dt), "TFLOPs": round(flops / dt / 1e12, 1)}
    except Exception as ex:  # noqa: BLE001
        return None, {"error": repr(ex)[:300]}


out = {}
clear()
ref, out["default"] = timed("default")
for k, v in SWITCHES.items():
    clear()
    os.environ[k] = v
    e, r = timed(k)
    if e is not None and ref is not None:
        r["max_abs_diff_vs_default"] = float((e - ref).abs().max())
    out[f"{k}={v}"] = r
clear()
os.environ["LEANN_MI355X_MLP"] = "1"
os.environ["LEANN_MI355X_MLP_VARIANT"] = "2"
e, r = timed("mlp2")
if e is not None and ref is not None:
    r["max_abs_diff_vs_default"] = float((e - ref).abs().max())
out["LEANN_MI355X_MLP=1,VARIANT=2"] = r
clear()
os.environ.update(SWITCHES)
os.environ["LEANN_MI355X_MLP_VARIANT"] = "2"
e, r = timed("all")
if e is not None and ref is not None:
    r["max_abs_diff_vs_default"] = float((e - ref).abs().max())
out["all"] = r
print(json.dumps(out, indent=1))

#!/usr/bin/env python
"""A/B of the whole packed encoder forward (MiniLM-L6 shape, fp16, ~262k tokens) under kernel switch sets; one JSON line."""
import json, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from leann_amd.encoder import BertEncoder, config_for
from leann_amd.synth import CorpusSpec, SyntheticCorpus, pad_batch

dev = torch.device("cuda")
cfg = config_for("all-MiniLM-L6-v2")
enc = BertEncoder.random_init(cfg, 0).to(dev, dtype=torch.float16)
ids, lens = pad_batch(*SyntheticCorpus(CorpusSpec(n_chunks=1460)).chunks(), 256)
ti, tl = torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)
flops = sum(cfg.flops_per_chunk(int(t)) for t in lens)
KEYS = ["LEANN_MI355X_LINEAR", "LEANN_MI355X_MLP_VARIANT", "LEANN_MI355X_MLP", "LEANN_MI355X_ATTN"]
SETS = {"default": {}, "linear2": {"LEANN_MI355X_LINEAR": "2"}, "mlp3": {"LEANN_MI355X_MLP_VARIANT": "3"},
        "linear2+mlp3": {"LEANN_MI355X_LINEAR": "2", "LEANN_MI355X_MLP_VARIANT": "3"},
        "library_gemms": {"LEANN_MI355X_MLP": "0"}}
out, ref = {"tokens": int(tl.sum())}, None
for rnd in range(2):  # interleaved rounds
    for name, env in SETS.items():
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        e = enc.encode_tokens_packed(ti, tl, 524288)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            e = enc.encode_tokens_packed(ti, tl, 524288)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        if ref is None:
            ref = e
        r = out.setdefault(name, {"ms": [], "TFLOPs": []})
        r["ms"].append(round(dt * 1e3, 3)); r["TFLOPs"].append(round(flops / dt / 1e12, 1))
        r["max_abs_diff_vs_default"] = float((e - ref).abs().max())
print(json.dumps(out))

#!/usr/bin/env python
"""One search round's worth of recompute (11,264 chunks, ~2 M tokens) through the packed encoder path at different sub-batch
token budgets, and with the fused layer tail on / off.  Prints one JSON line per setting (chunks/s, TFLOP/s)."""
import json, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from leann_amd.encoder import BertEncoder, config_for
from leann_amd.synth import CorpusSpec, SyntheticCorpus, pad_batch

dev = torch.device("cuda")
cfg = config_for("all-MiniLM-L6-v2")
enc = BertEncoder.random_init(cfg, 0).to(dev, dtype=torch.float16)
n = 11264
ids, lens = pad_batch(*SyntheticCorpus(CorpusSpec(n_chunks=n)).chunks(), 256)
ti, tl = torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)
flops = sum(cfg.flops_per_chunk(int(t)) for t in lens)


def run(max_tokens, reps=6):
    for _ in range(2):
        enc.encode_tokens_packed(ti, tl, max_tokens)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        enc.encode_tokens_packed(ti, tl, max_tokens)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for rnd in range(2):
    for tail in ("1", "0"):
        os.environ["LEANN_MI355X_TAIL"] = tail
        for mt in (2730 * 192, 1 << 20, 1 << 22):
            dt = run(mt)
            print(json.dumps({"round": rnd, "fused_layer_tail": tail, "sub_batch_tokens": mt, "ms": round(dt * 1e3, 2), "chunks_per_s": round(n / dt), "TFLOPs": round(flops / dt / 1e12, 1)}), flush=True)

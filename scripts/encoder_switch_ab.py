#!/usr/bin/env python
"""Interleaved A/B of encoder kernel switch sets on one GPU, one process: one search round's worth of recompute (n chunks of the
synthetic corpus' length distribution) through the packed forward.  Usage:
    python scripts/encoder_switch_ab.py <model> <n_chunks> <max_tokens> "K1=V1,K2=V2" "K1=V3" ...     ("-" = no switch: the default path)
Prints one JSON line per (round, switch set): ms, chunks/s, TFLOP/s."""
import json, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from leann_amd.encoder import BertEncoder, config_for
from leann_amd.synth import CorpusSpec, SyntheticCorpus, pad_batch

model, n, max_tokens = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sets = [dict(kv.split("=") for kv in a.split(",")) if a != "-" else {} for a in sys.argv[4:]] or [{}]
dev = torch.device("cuda")
cfg = config_for(model, strict=True)
enc = BertEncoder.random_init(cfg, 0).to(dev, dtype=torch.float16)
ids, lens = pad_batch(*SyntheticCorpus(CorpusSpec(n_chunks=n)).chunks(), 256)
ti, tl = torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)
flops = sum(cfg.flops_per_chunk(int(t)) for t in lens)
base_env = {k: v for k, v in os.environ.items() if k.startswith("LEANN_MI355X_")}


def run(reps=5):
    for _ in range(2):
        e = enc.encode_tokens_packed(ti, tl, max_tokens)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        e = enc.encode_tokens_packed(ti, tl, max_tokens)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, e


ref = None
for rnd in range(2):
    for sw in sets:
        for k in list(os.environ):
            if k.startswith("LEANN_MI355X_") and k not in base_env:
                del os.environ[k]
        os.environ.update(sw)
        dt, e = run()
        if ref is None:
            ref = e
        print(json.dumps({"model": model, "round": rnd, "switches": sw, "ms": round(dt * 1e3, 2), "chunks_per_s": round(n / dt), "TFLOPs": round(flops / dt / 1e12, 1),
                          "max_abs_diff_vs_first_set": round(float((e - ref).abs().max()), 5)}), flush=True)

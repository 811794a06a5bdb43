#!/usr/bin/env python
"""Why are small recompute rounds slow for 768-d models?  Times the packed forward of a bge-base shaped encoder on batches of
~640 chunks with a NEW token count each call (what a search round does) and with a repeated one, and one layer stage by stage."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch.nn.functional as F
from leann_amd.encoder import BertEncoder, config_for, fused_add_layernorm
from leann_amd.synth import CorpusSpec, SyntheticCorpus, pad_batch

dev = torch.device("cuda")
cfg = config_for("BAAI/bge-base-en-v1.5")
enc = BertEncoder.random_init(cfg, 0).to(dev, dtype=torch.float16).eval()
ids, lens = pad_batch(*SyntheticCorpus(CorpusSpec(n_chunks=8192)).chunks(), 256)
ti, tl = torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)


def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3


with torch.no_grad():
    _, ms = t(lambda: enc.encode_tokens_packed(ti[:640], tl[:640], 1 << 21)); print(json.dumps({"call": "first (640 chunks)", "ms": round(ms, 1)}), flush=True)
    for n in (640, 640, 611, 655, 590, 702, 640, 611):
        _, ms = t(lambda: enc.encode_tokens_packed(ti[:n], tl[:n], 1 << 21)); print(json.dumps({"chunks": n, "ms": round(ms, 1)}), flush=True)
    _, ms = t(lambda: enc.encode_tokens_packed(ti, tl, 1 << 21)); print(json.dumps({"chunks": 8192, "ms": round(ms, 1)}), flush=True)
    # one layer, stage by stage, on a fresh token count
    from torch.nn.attention.varlen import varlen_attn
    n = 633
    lens_n = tl[:n]
    cu = torch.zeros(n + 1, dtype=torch.int32, device=dev); cu[1:] = torch.cumsum(lens_n, 0)
    tot = int(cu[-1]); mx = int(lens_n.max())
    x = torch.randn((tot, cfg.hidden), device=dev).half()
    L = enc.layers[0]
    for rep in range(2):
        qkv2, a_ms = t(lambda: L.qkv(x))
        qkv = qkv2.view(tot, 3, L.heads, cfg.hidden // L.heads)
        a, b_ms = t(lambda: varlen_attn(qkv[:, 0], qkv[:, 1], qkv[:, 2], cu, cu, mx, mx).reshape(tot, cfg.hidden))
        y, c_ms = t(lambda: fused_add_layernorm(L.out(a), x, L.ln1))
        z, d_ms = t(lambda: fused_add_layernorm(L.fc2(F.gelu(L.fc1(y))), y, L.ln2))
        print(json.dumps({"layer stages, tokens": tot, "rep": rep, "qkv_ms": round(a_ms, 2), "varlen_attn_ms": round(b_ms, 2), "out_ln_ms": round(c_ms, 2), "mlp_ln_ms": round(d_ms, 2)}), flush=True)

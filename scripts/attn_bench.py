#!/usr/bin/env python
"""A/B of the hd=32 packed-sequence attention kernels on a search-round sized batch (~262k tokens of the
synthetic corpus): revision 1 (default), revision 2 (LEANN_MI355X_ATTN=2, lm_attn_v2.hip) and torch's varlen_attn.
Prints one JSON object; run on the GPU box:  python scripts/attn_bench.py"""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from leann_amd.encoder import fused_attention_hd32
from leann_amd.synth import CorpusSpec, SyntheticCorpus

heads, H = 12, 384
tok, off = SyntheticCorpus(CorpusSpec(n_chunks=1400)).chunks()
lens = torch.from_numpy((off[1:] - off[:-1]).astype("int32")).clamp(max=256)
cu = torch.zeros(lens.numel() + 1, dtype=torch.int32)
cu[1:] = torch.cumsum(lens, 0)
tot, mx = int(cu[-1]), int(lens.max())
g = torch.Generator(device="cpu").manual_seed(0)
qkv = torch.randn((tot, 3 * H), generator=g).half().cuda()
cud = cu.cuda()
flops = float((4 * lens.double() ** 2 * H).sum())


def run(name):
    if name == "torch":
        from torch.nn.attention.varlen import varlen_attn

        q = qkv.view(tot, 3, heads, 32)
        return lambda: varlen_attn(q[:, 0], q[:, 1], q[:, 2], cud, cud, mx, mx).reshape(tot, H)
    os.environ["LEANN_MI355X_ATTN"] = name
    return lambda: fused_attention_hd32(qkv, cud, heads, mx)


out = {"tokens": tot, "sequences": int(lens.numel()), "max_len": mx}
ref = None
for name in ("1", "2", "torch"):
    try:
        fn = run(name)
        o = fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            o = fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 100.0
        if ref is None:
            ref = o.float()
        out[name] = {"us": round(us, 1), "TFLOPs": round(flops / us / 1e6, 1), "max_abs_diff_vs_rev1": float((o.float() - ref).abs().max())}
    except Exception as ex:  # noqa: BLE001
        out[name] = {"error": repr(ex)[:300]}
print(json.dumps(out, indent=1))

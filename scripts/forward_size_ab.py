#!/usr/bin/env python
"""How many tokens should ONE forward of the library-side recompute provider take?  The product default is 5461 x 192 = 1,048,512 tokens (a
round of the 2048-query benchmark step is ~600 k): the activations of such a forward (x, attention output, y: 768 B per token each; Q / K / V:
2304 B) are 2.9 GB -- every kernel of every layer reads its input from HBM.  A forward of 65 k tokens works on ~300 MB, one of 32 k on
~150 MB: inside the 256 MB Infinity Cache.  Less HBM traffic is less power, and the benchmark's timed steps run at the socket's power cap
 -- against that, smaller launches fill the chip worse (a 65 k-token layer tail is two rounds of workgroups on 256 CUs).
This script measures the trade under SUSTAINED load: the same list of chunks through providers with different token budgets, several
seconds per setting, the settings interleaved, clocks and power sampled alongside.  One JSON line per (round, setting).

    python scripts/forward_size_ab.py [--chunks 400000] [--ids 350000] [--rounds 3] [--budgets 1048512,524288,...]
"""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import bench as B
from leann_amd.encoder import BertEncoder, config_for
from leann_amd.recompute import RecomputeProvider
from leann_amd.synth import CorpusSpec, SyntheticCorpus
from leann_amd.token_store import TokenStore

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=400_000)
ap.add_argument("--ids", type=int, default=350_000)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--calls", type=int, default=2, help="calls per measurement (each embeds --ids chunks)")
ap.add_argument("--budgets", default="1048512,524288,262144,131072,98304,65536,49152,32768")
ap.add_argument("--model", default="sentence-transformers/all-MiniLM-L6-v2")
ap.add_argument("--streams", default="1", help="comma-separated: with s > 1 the list is cut in s parts, each embedded by its own provider on its own stream from its own "
                "host thread (the foreign call releases the interpreter lock): one forward's last, partly filled round of workgroups runs beside the next kernel of another")
args = ap.parse_args()

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
corpus = SyntheticCorpus(CorpusSpec(n_chunks=args.chunks, seed=1234))
tok, off = corpus.chunks()
tokens = TokenStore(tok, off, device=0)
cfg = config_for(args.model)
enc = BertEncoder.load(args.model, allow_random=True).to(dev, dtype=torch.float16).eval()
rng = np.random.default_rng(7)
ids_np = rng.permutation(args.chunks)[: args.ids].astype(np.int32)
ids = torch.from_numpy(ids_np).to(dev)
lens = (off[1:] - off[:-1])[ids_np]
n_tokens = int(lens.sum())
flops = float(sum(cfg.flops_per_chunk(int(t)) for t in lens))
budgets = [int(b) for b in args.budgets.split(",")]
n_streams = [int(x) for x in args.streams.split(",")]
providers = {b: RecomputeProvider(enc, tokens, (cfg.hidden + 63) // 64 * 64, dev, batch_size=b // 192) for b in budgets}
extra = {(b, s, i): RecomputeProvider(enc, tokens, (cfg.hidden + 63) // 64 * 64, dev, batch_size=b // 192) for b in budgets for s in n_streams if s > 1 for i in range(s)}
streams = [torch.cuda.Stream(dev) for _ in range(max(n_streams))]


def embed_parts(b, s):
    """The list in s contiguous parts, part i through provider (b, s, i) on stream i from thread i; returns after every part was ISSUED and synchronised."""
    import threading

    cut = [len(ids) * i // s for i in range(s + 1)]
    outs = [None] * s

    def work(i):
        with torch.cuda.stream(streams[i]):
            outs[i] = extra[(b, s, i)].embed_ids(ids[cut[i]: cut[i + 1]])
        streams[i].synchronize()

    th = [threading.Thread(target=work, args=(i,)) for i in range(s)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return outs


ref = None
for b, p in providers.items():  # warm-up (allocations) + the outputs must not depend on the budget beyond the kernels' size classes
    e = p.embed_ids(ids)
    torch.cuda.synchronize()
    assert p.native() is not None, "library-side provider expected"
    if ref is None:
        ref = e
    else:
        print(json.dumps({"budget": b, "max_abs_diff_to_first_budget": float((e - ref).abs().max()), "bit_identical": bool(torch.equal(e, ref))}), flush=True)
    del e
    for s in n_streams:
        if s > 1:
            e = torch.cat(embed_parts(b, s))
            print(json.dumps({"budget": b, "streams": s, "max_abs_diff_to_first_budget": float((e - ref).abs().max())}), flush=True)
            del e
for rnd in range(args.rounds):
    for b, ns in [(b, s) for b in budgets for s in n_streams]:
        p = providers[b] if ns == 1 else extra[(b, ns, 0)]
        st0 = p.native_stats()
        torch.cuda.synchronize()
        smp = B.BoxSampler(0, period_s=0.2).start()
        t0 = time.perf_counter()
        for _ in range(args.calls):
            if ns == 1:
                p.embed_ids(ids)
            else:
                embed_parts(b, ns)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.calls
        box = smp.stop()
        st1 = p.native_stats()
        print(json.dumps({"round": rnd, "forward_tokens_budget": b, "streams": ns, "forwards_per_call_and_stream": (st1["forwards"] - st0["forwards"]) // args.calls, "chunks": args.ids, "tokens": n_tokens,
                          "ms_per_call": round(dt * 1e3, 1), "chunks_per_s": round(args.ids / dt), "us_per_262144_tokens_all_layers": round(dt * 1e6 * 262144 / n_tokens, 1),
                          "TFLOPs": round(flops / dt / 1e12, 1), "sclk_mhz": box.get("sclk_mhz"), "power_w": box.get("power_w")}), flush=True)

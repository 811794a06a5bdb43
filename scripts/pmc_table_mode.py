#!/usr/bin/env python
"""HBM-side (L2-miss) bytes of the stored-embedding search kernel k_search_table -- the kernel BASELINE.json's last sentence is written for
("distance/gather kernel at >= 60 % of gfx950 HBM peak") -- on the bench's own 1M-chunk index, 8192 DISTINCT queries in flight.

Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (scripts/pmc_table_mode.sh): the script issues, in this order,
  1. ONE calibration launch in the kernel's OWN access pattern (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access
     pattern"): lm_dist_gather over a random permutation of ALL 1M rows of the table -- load_row's 16 lanes x 16 B x 6 pieces per 1536-byte
     row, every row exactly once, 1.54 GB (> the 256 MB Infinity Cache) -> known bytes / FETCH_SIZE = the factor for this pattern;
  2. the persistent search, beam 1 and beam 4, ef 64, twice each (library's own choice of the wave / workgroup form).
It prints one JSON line with the per-call evaluation counts and kernel times in call order; the wrapper joins them with the counter rows."""
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from leann_amd import _lib  # noqa: E402
from leann_amd.encoder import BertEncoder, config_for  # noqa: E402
from leann_amd.gpu_graph_build import build_graph_gpu  # noqa: E402
from leann_amd.index import Mi355xIndex  # noqa: E402
from leann_amd.recompute import RecomputeProvider  # noqa: E402
from leann_amd.synth import CorpusSpec, SyntheticCorpus  # noqa: E402
from leann_amd.token_store import TokenStore  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 8192  # queries in flight (round 6: 8192 -> 5.1, 16384 -> 6.0, 32768 -> 6.5 TB/s algorithmic at beam 1)
dev = torch.device("cuda")
corpus = SyntheticCorpus(CorpusSpec(n_chunks=n, seed=1234))
tok, off = corpus.chunks()
enc = BertEncoder.random_init(config_for("all-MiniLM-L6-v2"), 0).to(dev, dtype=torch.float16).eval()
prov = RecomputeProvider(enc, TokenStore(tok, off), 384, dev)
X = torch.empty((n, 384), dtype=torch.float32, device=dev)
for b0 in range(0, n, 32768):
    ids = torch.arange(b0, min(n, b0 + 32768), dtype=torch.int32, device=dev)
    X[b0:b0 + ids.shape[0]] = prov.embed_ids(ids)
g = build_graph_gpu(X, "mips", M=32, ef_construction=200)
qt, qo, _ = corpus.queries(nq, seed=97531)
Q = RecomputeProvider(enc, TokenStore(qt, qo), 384, dev).embed_ids(torch.arange(nq, dtype=torch.int32, device=dev)).contiguous()
assert torch.unique(Q, dim=0).shape[0] == nq, "queries are not distinct"
idx = Mi355xIndex.from_csr(g)
st_ = torch.cuda.current_stream().cuda_stream
idx.set_stream(st_)
idx.attach_table(X)
idx.set_profiling(True)
out = {"n": n, "queries": nq, "distinct_queries": True, "mean_degree0": float(g.level0_degrees().mean()), "calls": []}
# 1. calibration: every row once, random order, the search kernel's own row loads
perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(5)).to(torch.int32)
qidx = torch.zeros(n, dtype=torch.int32, device=dev)
dout = torch.empty(n, dtype=torch.float32, device=dev)
torch.cuda.synchronize()
_lib.check(_lib.load().lm_dist_gather(C.c_void_p(X.data_ptr()), _lib.DTYPE_F32, 384, 0, C.c_void_p(Q.data_ptr()), C.c_void_p(qidx.data_ptr()),
                                      C.c_void_p(perm.data_ptr()), n, C.c_void_p(dout.data_ptr()), C.c_void_p(st_)), "lm_dist_gather")
torch.cuda.synchronize()
out["calibration"] = {"kernel": "k_dist_pairs", "rows": n, "known_bytes": n * (1536 + 4 + 4)}
# 2. the searches
idx.set_option("persistent_table", 1)
idx.set_option("persistent_wave", -1)
for rnd in range(2):
    for beam in (1, 4):
        prm = idx.make_params(ef=64, beam=beam, recompute=False, max_batch=32768)
        d, l = idx.search_device(Q, 10, prm)
        st = idx.stats()
        ms = max(st["update_span_ms"], 1e-9)
        out["calls"].append({"beam": beam, "ef": 64, "ndis": int(st["ndis"]), "nexpand": int(st["nexpand"]), "ms": round(ms, 3),
                             "algorithmic_bytes": int(st["ndis"]) * 1540, "algorithmic_GBps": round(st["ndis"] * 1540 / ms / 1e6, 1)})
print(json.dumps(out))

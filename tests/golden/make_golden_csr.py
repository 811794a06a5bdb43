"""Generates the golden compact-CSR fixtures with the REFERENCE's own writer.

Run in the dev container only (needs /root/reference):
    python tests/golden/make_golden_csr.py
It hand-writes a small index in the ORIGINAL faiss ``IHNf`` layout (the input format of
packages/leann-backend-hnsw/leann_backend_hnsw/convert_to_csr.py:264-301,439-479), then calls the
reference's ``convert_hnsw_graph_to_csr`` (pure numpy/struct, importable without faiss) twice:
prune_embeddings=True -> ref_csr_pruned.index ("null" storage), False -> ref_csr_full.index.
The arrays the graph was made from are stored in golden_graph.npz so the tests can compare.
"""
import contextlib
import importlib.util
import io
import struct
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference/packages/leann-backend-hnsw/leann_backend_hnsw/convert_to_csr.py")


def wvec(f, a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    f.write(struct.pack("<Q", a.shape[0]))
    f.write(a.tobytes())


def main():
    spec = importlib.util.spec_from_file_location("ref_convert_to_csr", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    rng = np.random.default_rng(20250921)
    n, d, M = 37, 8, 4
    x = rng.standard_normal((n, d)).astype(np.float32)
    # levels: geometric-ish, node 11 is the entry point on top
    levels = np.ones(n, np.int32)
    levels[[3, 11, 20, 29]] = 2
    levels[11] = 3
    cum = np.array([0, 2 * M, 3 * M, 4 * M], np.int32)  # faiss cum_nneighbor_per_level
    # padded original layout: every node owns cum[levels[i]] slots, -1 padded
    offsets = np.zeros(n + 1, np.uint64)
    offsets[1:] = np.cumsum(cum[levels])
    nb = -np.ones(int(offsets[-1]), np.int32)
    adj = []
    for i in range(n):
        per = []
        for l in range(levels[i]):
            pool = [j for j in range(n) if j != i and levels[j] > l]
            cap = int(cum[l + 1] - cum[l])
            cnt = int(rng.integers(0, min(cap, len(pool)) + 1)) if l > 0 else int(rng.integers(1, cap + 1))
            sel = rng.choice(pool, size=min(cnt, len(pool)), replace=False).astype(np.int32)
            per.append(sel)
            b = int(offsets[i]) + int(cum[l])
            nb[b : b + sel.shape[0]] = sel
        adj.append(per)
    orig = HERE / "ref_original.index"
    with open(orig, "wb") as f:
        f.write(struct.pack("<I", int.from_bytes(b"IHNf", "little")))
        f.write(struct.pack("<i", d))
        f.write(struct.pack("<q", n))
        f.write(struct.pack("<q", 1 << 20))
        f.write(struct.pack("<q", 1 << 20))
        f.write(struct.pack("<?", True))
        f.write(struct.pack("<i", 1))  # METRIC_L2
        wvec(f, np.array([0.75, 0.1875, 0.0625]), np.float64)
        wvec(f, cum, np.int32)
        wvec(f, levels, np.int32)
        f.write(b"\x00")  # storage_is_compact = False (the fork always writes this flag byte, :411-437)
        wvec(f, offsets, np.uint64)
        wvec(f, nb, np.int32)
        for v in (11, 2, 40, 16, 1):  # entry_point, max_level, efConstruction, efSearch, upper_beam
            f.write(struct.pack("<i", v))
        # flat storage (faiss IndexFlatL2 payload)
        f.write(struct.pack("<I", int.from_bytes(b"IxF2", "little")))
        f.write(struct.pack("<i", d))
        f.write(struct.pack("<q", n))
        f.write(struct.pack("<q", 1 << 20))
        f.write(struct.pack("<q", 1 << 20))
        f.write(struct.pack("<?", True))
        f.write(struct.pack("<i", 1))
        f.write(struct.pack("<Q", n * d))
        f.write(x.tobytes())
    for prune, name in ((True, "ref_csr_pruned.index"), (False, "ref_csr_full.index")):
        with contextlib.redirect_stdout(io.StringIO()):
            ok = ref.convert_hnsw_graph_to_csr(str(orig), str(HERE / name), prune_embeddings=prune)
        assert ok, name
    flat = np.concatenate([a for per in adj for a in per]).astype(np.int32)
    lens = np.array([a.shape[0] for per in adj for a in per], np.int32)
    np.savez(HERE / "golden_graph.npz", x=x, levels=levels, flat=flat, lens=lens, entry_point=11, max_level=2)
    print("wrote", [p.name for p in HERE.iterdir()])


if __name__ == "__main__":
    sys.exit(main())

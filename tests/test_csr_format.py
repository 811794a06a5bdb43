"""Compact-CSR file format: pinned against fixtures written by the REFERENCE's own converter
(tests/golden/make_golden_csr.py -> convert_to_csr.py:182-237,494-548)."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

from leann_amd import csr_format as cf

G = Path(__file__).resolve().parent / "golden"


def _expected():
    z = np.load(G / "golden_graph.npz")
    return z["x"], z["levels"], z["flat"], z["lens"], int(z["entry_point"]), int(z["max_level"])


def _check_graph(g: cf.HnswCsr, with_storage: bool):
    x, levels, flat, lens, ep, ml = _expected()
    g.validate()
    assert g.ntotal == 37 and g.d == 8 and g.metric_type == cf.METRIC_L2
    assert g.entry_point == ep and g.max_level == ml
    assert np.array_equal(g.levels, levels)
    assert np.array_equal(g.neighbors, flat)
    # per (node, level) list lengths in storage order
    got = []
    for i in range(g.ntotal):
        for l in range(int(g.levels[i])):
            got.append(len(g.neighbors_of(i, l)))
    assert got == lens.tolist()
    assert g.ef_construction == 40 and g.ef_search == 16
    if with_storage:
        assert g.storage is not None and np.array_equal(g.storage, x)
    else:
        assert g.storage is None and g.is_pruned


def test_read_reference_written_pruned_file():
    _check_graph(cf.read_index(G / "ref_csr_pruned.index"), with_storage=False)


def test_read_reference_written_full_file():
    _check_graph(cf.read_index(G / "ref_csr_full.index"), with_storage=True)


def test_read_original_layout_equals_reference_conversion():
    """Our vectorised padded->CSR conversion == the reference's per-node loop (convert_to_csr.py:494-548)."""
    a = cf.read_index(G / "ref_original.index")
    b = cf.read_index(G / "ref_csr_full.index")
    _check_graph(a, with_storage=True)
    for f in ("levels", "level_ptr", "node_offsets", "neighbors"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


def test_writer_is_byte_identical_to_reference_writer(tmp_path):
    for name, prune in (("ref_csr_pruned.index", True), ("ref_csr_full.index", False)):
        g = cf.read_index(G / "ref_csr_full.index")
        out = tmp_path / name
        cf.write_index(out, g, prune_embeddings=prune)
        assert out.read_bytes() == (G / name).read_bytes(), name


def test_roundtrip_built_graph(tmp_path, built_libs):
    from leann_amd.hnsw_builder import build_hnsw
    from tests.util import clustered

    x = clustered(500, 32, 3)
    g = build_hnsw(x, "mips", M=8, ef_construction=40)
    g.storage = x
    for prune in (True, False):
        p = tmp_path / f"g{prune}.index"
        cf.write_index(p, g, prune_embeddings=prune)
        r = cf.read_index(p)
        for f in ("levels", "level_ptr", "node_offsets", "neighbors"):
            assert np.array_equal(getattr(r, f), getattr(g, f))
        assert (r.storage is None) == prune
        assert r.entry_point == g.entry_point and r.max_level == g.max_level and r.metric_type == g.metric_type


def test_malformed_files(tmp_path):
    with pytest.raises(FileNotFoundError):
        cf.read_index(tmp_path / "missing.index")
    bad = tmp_path / "bad.index"
    bad.write_bytes(b"XXXX" + b"\0" * 64)
    with pytest.raises(ValueError):
        cf.read_index(bad)
    good = (G / "ref_csr_pruned.index").read_bytes()
    trunc = tmp_path / "trunc.index"
    trunc.write_bytes(good[: len(good) // 2])
    with pytest.raises(ValueError):
        cf.read_index(trunc)


def test_native_reader_error_codes(tmp_path, built_libs):
    """lm_index_read (C++ parser): missing -> LM_ENOENT, malformed -> LM_EFORMAT; a valid file passes
    parsing+validation and only then needs the GPU (LM_EHIP here, where no device is visible)."""
    from leann_amd import _lib

    lib = _lib.load()
    h = C.c_void_p()
    assert lib.lm_index_read(str(tmp_path / "nope.index").encode(), 0, C.byref(h)) == _lib.LM_ENOENT
    bad = tmp_path / "bad.index"
    bad.write_bytes(b"IHNf" + b"\x01" * 40)
    assert lib.lm_index_read(str(bad).encode(), 0, C.byref(h)) == _lib.LM_EFORMAT
    good = (G / "ref_csr_pruned.index").read_bytes()
    (tmp_path / "t.index").write_bytes(good[:700])
    assert lib.lm_index_read(str(tmp_path / "t.index").encode(), 0, C.byref(h)) == _lib.LM_EFORMAT
    if _lib.device_count() == 0:
        for name in ("ref_csr_pruned.index", "ref_csr_full.index", "ref_original.index"):
            rc = lib.lm_index_read(str(G / name).encode(), 0, C.byref(h))
            assert rc == _lib.LM_EHIP, (name, rc, _lib.last_error())
            assert "no HIP device" in _lib.last_error()

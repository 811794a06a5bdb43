"""Readers of a STOCK DiskANN bundle (leann_amd/diskann_files.py) over the committed fixture
tests/golden/stock_diskann/ -- written byte by byte with struct by tests/golden/make_golden_diskann.py, independently of the
module under test -- plus writer/reader round trips, malformed files, and the chunked product-quantiser arithmetic of the oracle."""
import struct
from pathlib import Path

import numpy as np
import pytest

from leann_amd import diskann_files as df

FX = Path(__file__).resolve().parent / "golden" / "stock_diskann"


@pytest.fixture(scope="module")
def expected():
    return np.load(FX / "expected.npz")


def test_small_files_match_what_the_reference_tests_spell_out():
    """tests/test_diskann_partition.py:258-281 of the reference: medoids = u32 1, u32 1, u32 id; norm = u32 1, u32 1, f32 > 0."""
    raw = (FX / "fx_disk.index_medoids.bin").read_bytes()
    assert len(raw) == 12 and struct.unpack("<III", raw)[:2] == (1, 1)
    assert df.read_medoids(FX / "fx_disk.index_medoids.bin").tolist() == [struct.unpack("<III", raw)[2]]
    raw = (FX / "fx_disk.index_max_base_norm.bin").read_bytes()
    assert struct.unpack("<II", raw[:8]) == (1, 1)
    assert df.read_max_base_norm(FX / "fx_disk.index_max_base_norm.bin") == struct.unpack("<f", raw[8:])[0] > 0


def test_stock_bundle_maps_onto_the_library_inputs(expected):
    b = df.load_stock_bundle(FX / "fx", 24, "mips")
    x = expected["x"]
    assert b.medoid == int(expected["medoid"]) and b.max_base_norm == pytest.approx(float(expected["max_norm"]))
    assert np.allclose(b.vectors, x, atol=2e-6)  # the MIPS transform undone: the caller's vectors
    # PQ: 6 stock chunks over 25 stored dimensions -> 8 chunks (2 empty ones: m % 4 == 0) over the caller's 24
    assert b.chunk_offsets.tolist() == [0, 5, 9, 13, 17, 21, 24, 24, 24]
    assert b.codes.shape == (200, 8) and np.array_equal(b.codes[:, :6], expected["codes"]) and not b.codes[:, 6:].any()
    recon = expected["recon"]  # (pivots + centroid)[:, :24] * max_norm
    off = b.chunk_offsets
    for j in range(8):
        lo, hi = int(off[j]), int(off[j + 1])
        assert np.array_equal(b.codebooks[256 * lo : 256 * hi].reshape(256, hi - lo), recon[:, lo:hi])
    # graph: ragged lists, ids as written
    g = b.graph()
    g.validate()
    assert g.entry_point == b.medoid and g.max_level == 0 and g.ntotal == 200
    assert np.array_equal(g.level0_degrees(), expected["degs"])
    for i in (0, 17, 199):
        assert np.array_equal(g.neighbors_of(i, 0), expected["nbrs"][i, : expected["degs"][i]])
    # the reconstruction from the codes approximates the vectors (a working quantiser, not noise)
    approx = np.concatenate([recon[b.codes[:, j], off[j] : off[j + 1]] for j in range(6)], 1)
    assert np.linalg.norm(approx - x) / np.linalg.norm(x) < 0.5


def test_writers_and_readers_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    piv, cen = rng.standard_normal((256, 10)).astype(np.float32), rng.standard_normal(10).astype(np.float32)
    chunk = np.array([0, 3, 6, 10], np.int32)
    df.write_pq_pivots(tmp_path / "a_pq_pivots.bin", piv, cen, chunk)
    p2, c2, k2 = df.read_pq_pivots(tmp_path / "a_pq_pivots.bin")
    assert np.array_equal(p2, piv) and np.array_equal(c2, cen) and np.array_equal(k2, chunk)
    v = rng.standard_normal((37, 10)).astype(np.float32)
    adj = [rng.choice(37, rng.integers(0, 6), replace=False) for _ in range(37)]
    df.write_disk_index(tmp_path / "a_disk.index", v, adj, 5)
    vec, deg, nbr, med = df.read_disk_index(tmp_path / "a_disk.index")
    assert np.array_equal(vec, v) and med == 5 and deg.tolist() == [len(a) for a in adj]
    assert np.array_equal(nbr, np.concatenate(adj).astype(np.int32))
    # records larger than a sector: one node per ceil(len / 4096) sectors
    big = rng.standard_normal((5, 1100)).astype(np.float32)
    df.write_disk_index(tmp_path / "b_disk.index", big, [[1], [2, 3], [], [0], [4]], 2)
    vec, deg, nbr, med = df.read_disk_index(tmp_path / "b_disk.index")
    assert np.array_equal(vec, big) and deg.tolist() == [1, 2, 0, 1, 1] and nbr.tolist() == [1, 2, 3, 0, 4]


def test_malformed_files_are_rejected(tmp_path):
    (tmp_path / "t.bin").write_bytes(struct.pack("<ii", 4, 4) + b"\0" * 10)
    with pytest.raises(df.DiskannFormatError, match="of 16 values"):
        df.read_bin(tmp_path / "t.bin", np.float32)
    (tmp_path / "m.bin").write_bytes(struct.pack("<IIf", 1, 1, -1.0))
    with pytest.raises(df.DiskannFormatError):
        df.read_max_base_norm(tmp_path / "m.bin")
    df.write_pq_pivots(tmp_path / "p_pq_pivots.bin", np.zeros((256, 6), np.float32), np.zeros(6, np.float32), np.array([0, 4, 2, 6]))
    with pytest.raises(df.DiskannFormatError, match="chunk offsets"):
        df.read_pq_pivots(tmp_path / "p_pq_pivots.bin")
    with pytest.raises(df.DiskannFormatError, match="dimension"):
        df.load_stock_bundle(FX / "fx", 16, "mips")  # the bundle is 24-d (25 stored)


def test_oracle_chunked_lookup_table_and_adc(expected):
    """oracle/lm_oracle_pq.c with chunk offsets (the arithmetic lm_pq_attach_chunked must reproduce): LUT[j][c] = -q[lo:hi] . cb, ADC =
    sum over chunks; uniform offsets give the bits of the uniform form."""
    from oracle import oracle as orc

    b = df.load_stock_bundle(FX / "fx", 24, "mips")
    q = expected["x"][3]
    lut, adc = orc.pq_lut_adc(b.codebooks, b.codes, q, 0, np.arange(200), chunk_off=b.chunk_offsets)
    recon = expected["recon"]
    off = b.chunk_offsets
    for j in (0, 5, 6):
        ref = -(recon[:, off[j] : off[j + 1]].astype(np.float64) @ q[off[j] : off[j + 1]].astype(np.float64))
        assert np.allclose(lut[j], ref, atol=1e-5)
    assert np.allclose(adc, lut[np.arange(8)[None, :], b.codes].sum(1), atol=1e-5)
    exact = -(expected["x"] @ q)
    assert np.corrcoef(adc, exact)[0, 1] > 0.9  # ADC ranks like the exact inner product
    rng = np.random.default_rng(1)
    cb = rng.standard_normal((4, 256, 6)).astype(np.float32)
    cd = rng.integers(0, 256, (50, 4)).astype(np.uint8)
    qq = rng.standard_normal(24).astype(np.float32)
    for metric in (0, 1):
        l0, a0 = orc.pq_lut_adc(cb, cd, qq, metric, np.arange(50))
        l1, a1 = orc.pq_lut_adc(cb.reshape(-1), cd, qq, metric, np.arange(50), chunk_off=np.arange(5) * 6)
        assert np.array_equal(l0, l1) and np.array_equal(a0, a1)

"""Wire-compatible embedding server: codec on CPU (checked against the reference's generated
embedding_pb2 when the reference tree is present), request handlers on the GPU."""
import importlib.util
from pathlib import Path

import numpy as np
import pytest

from leann_amd import embedding_server as es

REF_PB2 = Path("/root/reference/packages/leann-backend-diskann/leann_backend_diskann/embedding_pb2.py")


def test_protobuf_codec_roundtrip():
    ids = [0, 1, 127, 128, 300, 2**31, 2**32 - 1]
    assert es.decode_node_embedding_request(es.encode_node_embedding_request(ids)) == ids
    assert es.decode_node_embedding_request(b"") == []
    data = np.arange(12, dtype=np.float32).tobytes()
    buf = es.encode_node_embedding_response(data, [3, 4], [7, 9])
    assert es.decode_node_embedding_response(buf) == (data, [3, 4], [7, 9])
    with pytest.raises(ValueError):
        es.decode_node_embedding_request(b"\x0a\x05\x01")


@pytest.mark.skipif(not REF_PB2.exists(), reason="reference tree not present")
def test_protobuf_codec_matches_reference_generated_module():
    spec = importlib.util.spec_from_file_location("ref_embedding_pb2", REF_PB2)
    pb2 = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(pb2)
    except Exception as e:  # noqa: BLE001 - generated against another protobuf runtime
        pytest.skip(f"reference embedding_pb2 not importable here: {e}")
    ids = [5, 70000, 3, 2**32 - 1]
    req = pb2.NodeEmbeddingRequest()
    req.node_ids.extend(ids)
    assert es.decode_node_embedding_request(req.SerializeToString()) == ids
    r2 = pb2.NodeEmbeddingRequest()
    r2.ParseFromString(es.encode_node_embedding_request(ids))
    assert list(r2.node_ids) == ids
    data = np.arange(8, dtype=np.float32).tobytes()
    resp = pb2.NodeEmbeddingResponse()
    resp.ParseFromString(es.encode_node_embedding_response(data, [2, 4], [9]))
    assert resp.embeddings_data == data and list(resp.dimensions) == [2, 4] and list(resp.missing_ids) == [9]
    ref = pb2.NodeEmbeddingResponse(embeddings_data=data, dimensions=[2, 4], missing_ids=[9])
    assert es.decode_node_embedding_response(ref.SerializeToString()) == (data, [2, 4], [9])


@pytest.mark.gpu
def test_handlers_on_gpu():
    import msgpack
    import torch

    from leann_amd.encoder import BertEncoder, EncoderConfig
    from leann_amd.synth import CorpusSpec, SyntheticCorpus, pad_batch
    from leann_amd.token_store import TokenStore

    cfg = EncoderConfig(hidden=64, layers=2, heads=4, ffn=128, max_pos=256)
    enc = BertEncoder.random_init(cfg, 1).to("cuda", dtype=torch.float16)
    c = SyntheticCorpus(CorpusSpec(n_chunks=500, n_topics=4))
    tok, off = c.chunks()
    svc = es.Mi355xEmbeddingService("test-model", enc, TokenStore(tok, off), None, "mips")
    ids, lens = pad_batch(tok, off, 256)
    ref = enc.encode_tokens(torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda()).cpu().numpy()
    assert msgpack.unpackb(svc.handle_msgpack(msgpack.packb(["__QUERY_MODEL__"]))) == ["test-model"]
    # embeddings by id, with an unknown id -> zero row
    dims, flat = msgpack.unpackb(svc.handle_msgpack(msgpack.packb([[3, 499, 1000]])))
    e = np.asarray(flat, np.float32).reshape(dims)
    assert dims == [3, 64] and np.allclose(e[:2], ref[[3, 499]], atol=2e-3) and np.all(e[2] == 0)
    # distances: -e.q for mips, 1e9 for the unknown id
    q = ref[7].tolist()
    (d,) = msgpack.unpackb(svc.handle_msgpack(msgpack.packb([[7, 8, 12345], q])))
    exp = -(ref[[7, 8]] @ ref[7])
    assert np.allclose(d[:2], exp, atol=5e-3) and d[2] == pytest.approx(1e9)
    svc.distance_metric = "l2"
    (d2,) = msgpack.unpackb(svc.handle_msgpack(msgpack.packb([[[7, 8]], q])))
    assert np.allclose(d2, ((ref[[7, 8]] - ref[7]) ** 2).sum(1), atol=5e-3)
    # malformed request -> shape-correct fallback, never an exception
    assert msgpack.unpackb(svc.handle_msgpack(b"\xc1")) == [[0, 64], []]
    # DiskANN protocol
    data, dims, missing = es.decode_node_embedding_response(svc.handle_diskann(es.encode_node_embedding_request([1, 2, 3])))
    assert dims == [3, 64] and missing == [] and np.allclose(np.frombuffer(data, np.float32).reshape(3, 64), ref[1:4], atol=2e-3)
    assert svc.handle_diskann(b"") == b""
    assert es.decode_node_embedding_response(svc.handle_diskann(es.encode_node_embedding_request([10**6]))) == (b"", [], [])

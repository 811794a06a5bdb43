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


# ---------------------------------------------------------------------------------------------------------------
# transport + command line (f-1: drop-in for `python -m leann_backend_hnsw.hnsw_embedding_server ...`)
# ---------------------------------------------------------------------------------------------------------------
REF_SERVERS = {"hnsw": Path("/root/reference/packages/leann-backend-hnsw/leann_backend_hnsw/hnsw_embedding_server.py"),
               "diskann": Path("/root/reference/packages/leann-backend-diskann/leann_backend_diskann/diskann_embedding_server.py")}


def _reference_flags(path: Path) -> dict:
    """{flag: {"default": ..., "choices": ..., "type": ...}} of every parser.add_argument(...) call in a reference server."""
    import ast

    flags = {}
    for node in ast.walk(ast.parse(path.read_text())):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument":
            name = ast.literal_eval(node.args[0])
            kw = {k.arg: k.value for k in node.keywords}
            flags[name] = {"default": ast.literal_eval(kw["default"]) if "default" in kw else None,
                           "choices": ast.literal_eval(kw["choices"]) if "choices" in kw else None,
                           "type": kw["type"].id if "type" in kw else None}
    return flags


@pytest.mark.parametrize("flavour", ["hnsw", "diskann"])
def test_command_line_is_the_reference_servers(flavour):
    """Every flag of the reference parser (hnsw_embedding_server.py:395-417 / diskann_embedding_server.py:435-462) exists here
    with the same default, choices and type, and nothing is required that the reference does not require."""
    if not REF_SERVERS[flavour].exists():
        pytest.skip("reference tree not present")
    ref = _reference_flags(REF_SERVERS[flavour])
    assert set(ref) == {"--zmq-port", "--passages-file", "--model-name", "--distance-metric", "--embedding-mode"}
    ours = {a.option_strings[0]: a for a in es.build_parser(flavour)._actions if a.option_strings and a.option_strings[0] != "-h"}
    assert set(ours) == set(ref)
    for flag, r in ref.items():
        a = ours[flag]
        assert a.default == r["default"], (flag, a.default, r["default"])
        assert (list(a.choices) if a.choices else None) == r["choices"], flag
        assert (a.type.__name__ if a.type else None) == r["type"], flag
        assert not a.required
    # the exact command EmbeddingServerManager builds (embedding_server_manager.py:151-174) parses
    ns = es.build_parser(flavour).parse_args(["--zmq-port", "5557", "--model-name", "m", "--passages-file", "/x/i.meta.json", "--distance-metric", "mips"])
    assert (ns.zmq_port, ns.model_name, ns.embedding_mode) == (5557, "m", "sentence-transformers")
    with pytest.raises(SystemExit, match="only 'sentence-transformers'"):
        es.main(["--passages-file", "x", "--embedding-mode", "openai"], flavour=flavour)


def test_overlay_packages_resolve_the_reference_module_names():
    """`python -m leann_backend_hnsw.hnsw_embedding_server` with server_overlay/ in front of PYTHONPATH is OUR server (and the
    same for the DiskANN module name): the stock EmbeddingServerManager needs no change."""
    import subprocess
    import sys

    root = Path(__file__).resolve().parent.parent
    env = {"PYTHONPATH": f"{root / 'server_overlay'}:{root}", "PATH": "/usr/bin:/bin"}
    for mod in ("leann_backend_hnsw.hnsw_embedding_server", "leann_backend_diskann.diskann_embedding_server"):
        r = subprocess.run([sys.executable, "-m", mod, "--help"], capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 0 and "(MI355X)" in r.stdout and "--zmq-port" in r.stdout and "--embedding-mode" in r.stdout, r.stderr[-500:]
        r = subprocess.run([sys.executable, "-m", mod, "--zmq-port", "5999", "--model-name", "m", "--embedding-mode", "ollama",
                            "--passages-file", "nope.meta.json"], capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode != 0 and "only 'sentence-transformers'" in r.stderr


def test_zmtp_rep_endpoint_wire_level():
    """leann_amd/zmtp.py against the published ZMTP 3.0 framing, byte by byte: greeting, READY, REQ envelope, short and long
    frames, several requests on one connection, two connections, a partial (split) write."""
    import socket
    import struct
    import threading

    from leann_amd import zmtp

    assert zmtp.GREETING[:10] == b"\xff" + bytes(8) + b"\x7f" and zmtp.GREETING[10:12] == b"\x03\x00" and zmtp.GREETING[12:16] == b"NULL"
    srv = zmtp.RepServer(0, host="127.0.0.1")
    stop = threading.Event()
    th = threading.Thread(target=srv.serve, args=(lambda m: b"echo:" + m, stop), kwargs={"poll_s": 0.05}, daemon=True)
    th.start()
    try:
        s = socket.create_connection(("127.0.0.1", srv.port), timeout=10)
        s.sendall(zmtp.GREETING[:11])  # libzmq sends the greeting in pieces
        s.sendall(zmtp.GREETING[11:] + zmtp._ready(b"REQ"))
        buf = b""
        while len(buf) < 64 + 27:
            buf += s.recv(4096)
        assert buf[:64] == zmtp.GREETING
        assert buf[64] == 0x04 and buf[66:72] == b"\x05READY" and b"Socket-Type" in buf and buf.endswith(b"\x00\x00\x00\x03REP")
        s.sendall(b"\x01\x00" + b"\x00\x05hello")  # delimiter (MORE, size 0) + short frame
        assert s.recv(4096) == b"\x01\x00" + b"\x00\x0aecho:hello"
        big = bytes(range(256)) * 5  # 1280 bytes -> LONG frame both ways
        s.sendall(b"\x01\x00" + b"\x02" + struct.pack(">Q", len(big)) + big[:100])
        s.sendall(big[100:])
        got = b""
        while len(got) < 2 + 9 + 5 + len(big):
            got += s.recv(1 << 16)
        assert got[:2] == b"\x01\x00" and got[2] == 0x02 and struct.unpack(">Q", got[3:11])[0] == len(big) + 5 and got[11:] == b"echo:" + big
        c2 = zmtp.ReqClient(srv.port)
        assert c2.request(b"") == b"echo:" and c2.request(b"x" * 70000) == b"echo:" + b"x" * 70000
        s.sendall(b"\x01\x00\x00\x01z")
        assert s.recv(4096) == b"\x01\x00\x00\x06echo:z"
        c2.close()
        s.close()
        # what a libzmq 4.x peer actually puts on the wire (ZMTP 3.1): minor version 1, a READY command carrying TWO properties
        # (Socket-Type = REQ and an empty Identity), heartbeat PING commands between requests (answered with PONG + the ping's context)
        s3 = socket.create_connection(("127.0.0.1", srv.port), timeout=10)
        g31 = b"\xff" + bytes(8) + b"\x7f" + b"\x03\x01" + b"NULL".ljust(20, b"\x00") + b"\x00" + bytes(31)
        ready = b"\x05READY" + b"\x0bSocket-Type" + struct.pack(">I", 3) + b"REQ" + b"\x08Identity" + struct.pack(">I", 0)
        s3.sendall(g31[:10])
        s3.sendall(g31[10:] + bytes([0x04, len(ready)]) + ready)
        buf = b""
        while len(buf) < 64 + 27:
            buf += s3.recv(4096)
        ping = b"\x04PING" + struct.pack(">H", 300) + b"ctx1"  # TTL (deciseconds) + context
        s3.sendall(bytes([0x04, len(ping)]) + ping)
        pong = s3.recv(4096)
        assert pong == bytes([0x04, 9]) + b"\x04PONG" + b"ctx1", pong
        s3.sendall(b"\x01\x00" + b"\x00\x03abc")
        assert s3.recv(4096) == b"\x01\x00\x00\x08echo:abc"
        s3.close()
    finally:
        stop.set()
        th.join(5)
        srv.close()


@pytest.mark.gpu
def test_serve_loop_over_a_real_socket(tmp_path):
    """service_from_meta(...).serve(port) driven through TCP by a REQ client: msgpack model query, embeddings, distances
    (hnsw protocol) and protobuf embeddings (diskann protocol) -- the whole path the stock backends use."""
    import json
    import threading

    import msgpack

    from leann_amd import zmtp

    texts = [f"passage {i} " + " ".join(f"w{(i * 7 + j) % 40}" for j in range(10)) for i in range(64)]
    base = tmp_path / "srv.leann"
    (tmp_path / "srv.leann.passages.jsonl").write_text("".join(json.dumps({"id": str(i), "text": t}) + "\n" for i, t in enumerate(texts)))
    meta = {"passage_sources": [{"type": "jsonl", "path": "srv.leann.passages.jsonl", "path_relative": "srv.leann.passages.jsonl"}]}
    (tmp_path / "srv.leann.meta.json").write_text(json.dumps(meta))
    svc = es.service_from_meta(str(base) + ".meta.json", "sentence-transformers/all-MiniLM-L6-v2", "mips", allow_random=True)
    for protocol in ("hnsw", "diskann"):
        stop, ready = threading.Event(), threading.Event()
        port = 20000 + (hash((protocol, str(tmp_path))) % 20000)
        th = threading.Thread(target=svc.serve, args=(port, protocol, stop, ready), daemon=True)
        th.start()
        assert ready.wait(30)
        try:
            try:
                import zmq  # a real libzmq REQ socket when pyzmq is installed

                ctx = zmq.Context()
                sk = ctx.socket(zmq.REQ)
                sk.connect(f"tcp://127.0.0.1:{port}")

                def ask(b):
                    sk.send(b)
                    return sk.recv()
            except ImportError:
                cl = zmtp.ReqClient(port)
                ask = cl.request
            if protocol == "hnsw":
                assert msgpack.unpackb(ask(msgpack.packb(["__QUERY_MODEL__"]))) == ["sentence-transformers/all-MiniLM-L6-v2"]
                dims, flat = msgpack.unpackb(ask(msgpack.packb([[3, 9, 63]])))
                e = np.asarray(flat, np.float32).reshape(dims)
                assert dims == [3, 384] and np.allclose(np.linalg.norm(e, axis=1), 1.0, atol=2e-3)
                (d,) = msgpack.unpackb(ask(msgpack.packb([[3, 9], e[0].tolist()])))
                assert d[0] == pytest.approx(-1.0, abs=5e-3) and d[1] > d[0]
            else:
                data, dims, missing = es.decode_node_embedding_response(ask(es.encode_node_embedding_request([3, 9, 63])))
                assert dims == [3, 384] and missing == [] and len(data) == 3 * 384 * 4
        finally:
            stop.set()
            th.join(10)


def test_transport_survives_a_failing_handler_and_refuses_oversized_frames():
    """leann_amd/zmtp.py: a handler exception answers with the empty (error) reply and the serve loop goes on; a frame header
    announcing more than MAX_FRAME_BYTES drops that connection instead of growing the receive buffer."""
    import socket
    import struct
    import threading

    from leann_amd import zmtp

    srv = zmtp.RepServer(0, host="127.0.0.1")
    stop = threading.Event()

    def handler(b):
        if b == b"boom":
            raise RuntimeError("handler failure")
        return b"echo:" + b

    th = threading.Thread(target=srv.serve, args=(handler, stop, 0.05), daemon=True)
    th.start()
    try:
        c = zmtp.ReqClient(srv.port)
        assert c.request(b"boom") == b"" and c.request(b"again") == b"echo:again"
        raw = socket.create_connection(("127.0.0.1", srv.port), timeout=5)
        raw.sendall(zmtp.GREETING + zmtp._ready(b"REQ") + bytes([zmtp.FLAG_LONG]) + struct.pack(">Q", zmtp.MAX_FRAME_BYTES + 1))
        raw.settimeout(5)
        data = b"x"
        while data:  # the server closes the connection (after its own greeting / READY bytes)
            data = raw.recv(4096)
        raw.close()
        assert c.request(b"still serving") == b"echo:still serving"
        c.close()
    finally:
        stop.set()
        th.join(timeout=5)
        srv.close()

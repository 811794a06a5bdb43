"""Wire-compatible MI355X embedding / distance server.

Lets the STOCK LEANN backends (faiss / DiskANN forks on the CPU) use the GPU for the recompute step
with zero code change: same transport (ZeroMQ REP on ``tcp://*:<port>``), same payloads and the same
CLI flags as
  packages/leann-backend-hnsw/leann_backend_hnsw/hnsw_embedding_server.py:97-324,395-428   (msgpack)
  packages/leann-backend-diskann/leann_backend_diskann/diskann_embedding_server.py:223-364 (protobuf)
but the work behind a request is the HBM token store + in-process encoder + ``lm_dist_gather``:

  ["__QUERY_MODEL__"]                -> [model_name]
  [str, ...]                         -> [[f, ...] x n]                        (text embeddings)
  [[id, ...], [q_0 .. q_{D-1}]]      -> [[d_0 .. d_{n-1}]]  float32, 1e9 for unknown ids
                                        (l2: sum (e-q)^2 ; mips/cosine: -e.q   :195-200)
  [[id, ...]] | [id, ...]            -> [[n, D], flat n*D float32], zero rows for unknown ids
  protobuf NodeEmbeddingRequest      -> NodeEmbeddingResponse{embeddings_data, dimensions=[n,D], missing_ids}

The request handlers are pure ``bytes -> bytes`` functions (tested without a socket); ``serve()``
needs ``pyzmq`` (not installed in the build image -- it raises a clear error then).
"""

from __future__ import annotations

import argparse
import json
import logging
import os
import sys
import threading
from pathlib import Path
from typing import Optional

import numpy as np

logger = logging.getLogger(__name__)

LARGE_DISTANCE = 1e9  # hnsw_embedding_server.py:184


# ---------------------------------------------------------------------------------------------
# minimal protobuf codec for embedding.proto (packages/leann-backend-diskann/third_party/embedding.proto:5-13)
# ---------------------------------------------------------------------------------------------
def _varint(v: int) -> bytes:
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf: bytes, pos: int):
    shift, val = 0, 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7
        if shift > 70:
            raise ValueError("varint too long")


def decode_node_embedding_request(buf: bytes) -> list[int]:
    """NodeEmbeddingRequest{repeated uint32 node_ids = 1} (packed or unpacked)."""
    ids, pos = [], 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if field == 1 and wt == 2:
            ln, pos = _read_varint(buf, pos)
            end = pos + ln
            if end > len(buf):
                raise ValueError("truncated packed field")
            while pos < end:
                v, pos = _read_varint(buf, pos)
                ids.append(v)
        elif field == 1 and wt == 0:
            v, pos = _read_varint(buf, pos)
            ids.append(v)
        else:
            raise ValueError(f"unexpected field {field} / wire type {wt}")
    return ids


def encode_node_embedding_request(ids) -> bytes:
    body = b"".join(_varint(int(i)) for i in ids)
    return (b"\x0a" + _varint(len(body)) + body) if len(body) else b""


def encode_node_embedding_response(data: bytes, dims, missing) -> bytes:
    out = bytearray()
    if data:
        out += b"\x0a" + _varint(len(data)) + data
    if len(dims):
        body = b"".join(_varint(int(d)) for d in dims)
        out += b"\x12" + _varint(len(body)) + body
    if len(missing):
        body = b"".join(_varint(int(m)) for m in missing)
        out += b"\x1a" + _varint(len(body)) + body
    return bytes(out)


def decode_node_embedding_response(buf: bytes):
    data, dims, missing, pos = b"", [], [], 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 2:
            ln, pos = _read_varint(buf, pos)
            chunk = buf[pos : pos + ln]
            pos += ln
            if field == 1:
                data = bytes(chunk)
            else:
                p = 0
                while p < len(chunk):
                    v, p = _read_varint(chunk, p)
                    (dims if field == 2 else missing).append(v)
        elif wt == 0:
            v, pos = _read_varint(buf, pos)
            (dims if field == 2 else missing).append(v)
        else:
            raise ValueError("unexpected wire type")
    return data, dims, missing


# ---------------------------------------------------------------------------------------------
# service
# ---------------------------------------------------------------------------------------------
class Mi355xEmbeddingService:
    """GPU-side request handlers.  ``tokens`` is a TokenStore (passage i == node id i)."""

    def __init__(self, model_name: str, encoder, tokens, tokenizer=None, distance_metric: str = "mips", device=None):
        import torch

        from .recompute import RecomputeProvider

        self.model_name = model_name
        self.encoder = encoder
        self.tokens = tokens
        self.tokenizer = tokenizer
        self.distance_metric = distance_metric.lower()
        self.device = device or torch.device("cuda", 0)
        self.dim = encoder.cfg.hidden
        self.dp = (self.dim + 63) // 64 * 64
        self.provider = RecomputeProvider(encoder, tokens, self.dp, self.device)

    # ---- primitives ------------------------------------------------------------------------------
    def embed_ids(self, node_ids):
        """-> (emb float32 [n, D] torch (zeros for unknown ids), known mask np.bool_[n])"""
        import torch

        ids = np.asarray(list(node_ids), dtype=np.int64)
        known = (ids >= 0) & (ids < self.tokens.n)
        out = torch.zeros((ids.shape[0], self.dim), dtype=torch.float32, device=self.device)
        if known.any():
            t = torch.from_numpy(ids[known].astype(np.int32)).to(self.device)
            out[torch.from_numpy(np.nonzero(known)[0]).to(self.device)] = self.provider.embed_ids(t)[:, : self.dim]
        return out, known

    def embed_texts(self, texts):
        import torch

        if self.tokenizer is None:
            raise RuntimeError("no tokenizer attached: text requests are unavailable")
        seqs = self.tokenizer.encode_batch(list(texts))
        T = max(len(s) for s in seqs)
        ids = torch.zeros((len(seqs), T), dtype=torch.int32)
        for i, s in enumerate(seqs):
            ids[i, : len(s)] = torch.tensor(s, dtype=torch.int32)
        lens = torch.tensor([len(s) for s in seqs], dtype=torch.int32)
        with torch.no_grad():
            return self.encoder.encode_tokens(ids.to(self.device), lens.to(self.device)).float()

    def distances(self, node_ids, query: np.ndarray) -> np.ndarray:
        """hnsw_embedding_server.py:148-211: l2 -> sum (e-q)^2, mips/cosine -> -e.q; unknown id -> 1e9.
        Computed by the lm_dist_gather kernel (canonical reduction)."""
        import ctypes as C

        import torch

        from . import _lib

        emb, known = self.embed_ids(node_ids)
        n = emb.shape[0]
        res = np.full(n, LARGE_DISTANCE, dtype=np.float32)
        if n == 0:
            return res
        e = torch.zeros((n, self.dp), dtype=torch.float32, device=self.device)
        e[:, : self.dim] = emb
        q = torch.zeros((1, self.dp), dtype=torch.float32, device=self.device)
        q[0, : self.dim] = torch.from_numpy(np.asarray(query, dtype=np.float32)).to(self.device)
        rows = torch.arange(n, dtype=torch.int32, device=self.device)
        qidx = torch.zeros(n, dtype=torch.int32, device=self.device)
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        metric = _lib.METRIC_L2 if self.distance_metric == "l2" else _lib.METRIC_INNER_PRODUCT
        _lib.check(_lib.load().lm_dist_gather(C.c_void_p(e.data_ptr()), _lib.DTYPE_F32, self.dp, metric, C.c_void_p(q.data_ptr()),
                                              C.c_void_p(qidx.data_ptr()), C.c_void_p(rows.data_ptr()), n, C.c_void_p(out.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "lm_dist_gather")
        d = out.cpu().numpy()
        res[known] = d[known]
        return res

    # ---- wire handlers ------------------------------------------------------------------------------
    def handle_msgpack(self, request_bytes: bytes) -> bytes:
        """The HNSW server's dispatch (hnsw_embedding_server.py:119-284), shape-correct fallbacks
        on errors (:289-313)."""
        import msgpack

        kind, length = "unknown", 0
        try:
            request = msgpack.unpackb(request_bytes)
            if isinstance(request, list) and len(request) == 1 and request[0] == "__QUERY_MODEL__":
                return msgpack.packb([self.model_name])
            if isinstance(request, list) and request and all(isinstance(x, str) for x in request):
                kind, length = "text", len(request)
                return msgpack.packb(self.embed_texts(request).cpu().numpy().astype(np.float64).tolist())
            if isinstance(request, list) and len(request) == 2 and isinstance(request[0], list) and isinstance(request[1], list):
                node_ids = request[0]
                if len(node_ids) == 1 and isinstance(node_ids[0], list):
                    node_ids = node_ids[0]
                kind, length = "distance", len(node_ids)
                d = self.distances(node_ids, np.asarray(request[1], dtype=np.float32))
                return msgpack.packb([d.tolist()], use_single_float=True)
            if isinstance(request, list) and len(request) == 1 and isinstance(request[0], list):
                node_ids = request[0]
            elif isinstance(request, list):
                node_ids = request
            else:
                node_ids = []
            kind, length = "embedding", len(node_ids)
            emb, _ = self.embed_ids(node_ids)
            e = emb.cpu().numpy()
            if not np.isfinite(e).all():
                return msgpack.packb([[0, self.dim], []], use_single_float=True)
            return msgpack.packb([[len(node_ids), self.dim], e.reshape(-1).tolist()], use_single_float=True)
        except Exception as ex:  # noqa: BLE001 - the reference replies shape-correct sentinels
            logger.error(f"request failed: {ex}")
            if kind == "distance":
                safe = [[LARGE_DISTANCE] * length]
            elif kind == "embedding":
                safe = [[length, self.dim], [0.0] * (length * self.dim)]
            elif kind == "text":
                safe = []
            else:
                safe = [[0, self.dim], []]
            return msgpack.packb(safe, use_single_float=True)

    def handle_diskann(self, message: bytes) -> bytes:
        """The DiskANN server's dispatch (diskann_embedding_server.py:246-353): protobuf first,
        msgpack list[str] fallback, empty message -> empty reply, errors -> empty response."""
        if not message:
            return b""
        try:
            node_ids = decode_node_embedding_request(message)
            emb, known = self.embed_ids(node_ids)
            if not known.all():
                raise RuntimeError("passage id not found")  # the reference treats this as fatal -> empty response
            e = np.ascontiguousarray(emb.cpu().numpy(), dtype=np.float32)
            return encode_node_embedding_response(e.tobytes(), [e.shape[0], e.shape[1]], [])
        except Exception:  # noqa: BLE001
            try:
                import msgpack

                request = msgpack.unpackb(message)
                if isinstance(request, list) and all(isinstance(x, str) for x in request):
                    return msgpack.packb(self.embed_texts(request).cpu().numpy().tolist())
            except Exception:  # noqa: BLE001
                pass
            return encode_node_embedding_response(b"", [], [])

    # ---- transport ------------------------------------------------------------------------------------
    def serve(self, zmq_port: int, protocol: str = "hnsw", shutdown_event: Optional[threading.Event] = None,
              ready_event: Optional[threading.Event] = None) -> None:
        """The REP loop of the reference servers (hnsw_embedding_server.py:97-324, diskann_embedding_server.py:223-364): bind
        tcp://*:port, one request -> one reply, poll with a 1 s timeout so that a shutdown request is noticed.  With pyzmq
        installed this is a zmq.REP socket; without it (this image, the GPU box) the same wire protocol is spoken directly by
        leann_amd/zmtp.py -- REQ clients (the faiss / DiskANN forks, searcher_base.py:130-160) cannot tell the difference."""
        shutdown_event = shutdown_event or threading.Event()
        handler = self.handle_msgpack if protocol == "hnsw" else self.handle_diskann
        try:
            import zmq
        except ImportError:
            zmq = None
        if zmq is None:
            from .zmtp import RepServer

            srv = RepServer(zmq_port)
            logger.info(f"embedding server (native ZMTP REP) listening on tcp://*:{srv.port}")
            if ready_event is not None:
                ready_event.set()
            try:
                srv.serve(handler, shutdown_event)
            finally:
                srv.close()
            return
        ctx = zmq.Context()
        sock = ctx.socket(zmq.REP)
        sock.bind(f"tcp://*:{zmq_port}")
        sock.setsockopt(zmq.RCVTIMEO, 1000)
        sock.setsockopt(zmq.SNDTIMEO, 1000)
        sock.setsockopt(zmq.LINGER, 0)
        if ready_event is not None:
            ready_event.set()
        try:
            while not shutdown_event.is_set():
                try:
                    msg = sock.recv()
                except zmq.Again:
                    continue
                sock.send(handler(msg))
        finally:
            sock.close(0)
            ctx.term()


def service_from_meta(passages_file: str, model_name: str, distance_metric: str = "mips", device_index: int = 0,
                      allow_random: bool = False):
    """Build the service the way the reference servers do from ``--passages-file <index>.meta.json``
    (hnsw_embedding_server.py:60-87): passages JSONL -> tokens in HBM."""
    import torch

    from .encoder import BertEncoder
    from .token_store import TokenStore
    from .tokenizer import load_tokenizer

    meta_path = Path(passages_file)
    meta = json.loads(meta_path.read_text(encoding="utf-8"))
    texts = []
    for src in meta.get("passage_sources", []):
        for cand in (meta_path.parent / src.get("path_relative", ""), Path(src.get("path", "")), meta_path.parent / src.get("path", "")):
            if cand.is_file():
                with open(cand, encoding="utf-8") as f:
                    texts += [json.loads(line).get("text", "") for line in f if line.strip()]
                break
    dev = torch.device("cuda", device_index)
    enc = BertEncoder.load(model_name, allow_random=allow_random).to(dev, dtype=torch.float16).eval()
    index_path = str(meta_path)[: -len(".meta.json")] if str(meta_path).endswith(".meta.json") else str(meta_path)
    tok = load_tokenizer(model_name, min(enc.cfg.max_seq_length, enc.cfg.max_pos), index_path, texts, enc.cfg.vocab_size,
                         allow_stand_in=enc.weights_source == "random")
    tokens = TokenStore.from_lists(tok.encode_batch(texts), device=device_index)
    return Mi355xEmbeddingService(model_name, enc, tokens, tok, distance_metric, dev)


EMBEDDING_MODES = ["sentence-transformers", "openai", "mlx", "ollama"]  # the reference parsers' choices


def build_parser(flavour: str = "hnsw") -> argparse.ArgumentParser:
    """EXACTLY the command line of the reference servers -- the stock EmbeddingServerManager builds
    ``python -m <backend module> --zmq-port P --model-name M [--passages-file F] [--embedding-mode E] [--distance-metric D]``
    (embedding_server_manager.py:151-174).  hnsw_embedding_server.py:395-417: --distance-metric is free-form, default "mips";
    diskann_embedding_server.py:435-462: choices l2 / mips / cosine, default "l2"."""
    ap = argparse.ArgumentParser(description=f"{'HNSW' if flavour == 'hnsw' else 'DiskANN'} Embedding service (MI355X)")
    ap.add_argument("--zmq-port", type=int, default=5555, help="ZMQ port to run on")
    ap.add_argument("--passages-file", type=str, help="Metadata JSON file containing passage sources (<index>.meta.json)")
    ap.add_argument("--model-name", type=str, default="sentence-transformers/all-mpnet-base-v2", help="Embedding model name")
    if flavour == "hnsw":
        ap.add_argument("--distance-metric", type=str, default="mips", help="Distance metric to use")
    else:
        ap.add_argument("--distance-metric", type=str, default="l2", choices=["l2", "mips", "cosine"],
                        help="Distance metric for similarity computation")
    ap.add_argument("--embedding-mode", type=str, default="sentence-transformers", choices=EMBEDDING_MODES, help="Embedding backend mode")
    return ap


def main(argv=None, flavour: str = "hnsw"):
    """Entry point of the launch shims (server_overlay/leann_backend_hnsw/hnsw_embedding_server.py and
    server_overlay/leann_backend_diskann/diskann_embedding_server.py) and of ``python -m leann_amd.embedding_server``."""
    import signal

    ap = build_parser(flavour)
    if flavour == "hnsw" and argv is None and "--protocol" in sys.argv:  # our own module entry: allow choosing the wire protocol
        ap.add_argument("--protocol", type=str, default="hnsw", choices=["hnsw", "diskann"])
    args = ap.parse_args(argv)
    if args.embedding_mode != "sentence-transformers":
        # openai / ollama are remote HTTP services and mlx is Apple-silicon only (embedding_compute.py:25-68): nothing for a GPU
        # to recompute.  Fail loudly instead of serving embeddings from a different model.
        raise SystemExit(f"--embedding-mode {args.embedding_mode}: only 'sentence-transformers' models are served by the MI355X server")
    if not args.passages_file:
        raise SystemExit("--passages-file <index>.meta.json is required")
    logging.basicConfig(level=os.environ.get("LEANN_LOG_LEVEL", "WARNING"))
    # random weights only on explicit request (synthetic corpora): the reference CLI has no such flag, hence the environment
    svc = service_from_meta(args.passages_file, args.model_name, args.distance_metric,
                            allow_random=os.environ.get("LEANN_MI355X_ALLOW_RANDOM_WEIGHTS", "0") == "1")
    stop = threading.Event()
    for sig in (signal.SIGTERM, signal.SIGINT):  # EmbeddingServerManager.stop_server() terminates the process (SIGTERM)
        try:
            signal.signal(sig, lambda *_: stop.set())
        except ValueError:  # not the main thread (tests)
            pass
    svc.serve(args.zmq_port, getattr(args, "protocol", "diskann" if flavour == "diskann" else "hnsw"), stop)


if __name__ == "__main__":
    main()

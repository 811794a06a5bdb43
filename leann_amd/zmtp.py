"""Minimal ZeroMQ wire protocol (ZMTP 3.x, NULL security) REP endpoint over plain TCP sockets.

The reference's embedding servers are ``zmq.REP`` sockets (hnsw_embedding_server.py:97-110, diskann_embedding_server.py:223-236)
and their clients -- the faiss / DiskANN forks' C++ code and ``searcher_base._compute_embedding_via_server`` (searcher_base.py:
130-160) -- are ``REQ`` sockets.  pyzmq is not part of this image, and a drop-in server must not add a dependency the GPU box
lacks, so the REP side of the protocol is spoken directly (RFC 23/ZMTP 3.0, https://rfc.zeromq.org/spec/23/):

  greeting   64 bytes: signature FF 00*8 7F | version 03 00 | mechanism "NULL" padded to 20 | as-server 00 | 31 filler bytes
  handshake  one READY command each way: frame flags 04 (COMMAND) | size | 05 "READY" | properties (name, 4-byte BE length, value),
             here Socket-Type = REP (the peer must be REQ or DEALER)
  messages   frames: flags (01 MORE, 02 LONG, 04 COMMAND) | size (1 byte, or 8 bytes BE when LONG) | body.
             A REQ message is an empty delimiter frame followed by the body frame(s); the REP reply mirrors the envelope.

``RepServer.serve(handler)`` is the whole server: single-threaded like the reference's REP loop, any number of client
connections, one outstanding request per connection (the REQ/REP lock-step).  ``ReqClient`` is the matching client used by the
tests (and a convenience for probing a running server without pyzmq).  When pyzmq *is* installed the embedding server uses it
instead (leann_amd/embedding_server.py: serve)."""

from __future__ import annotations

import selectors
import socket
import struct
import threading
from typing import Callable, Optional

GREETING = b"\xff" + b"\x00" * 8 + b"\x7f" + b"\x03\x00" + b"NULL".ljust(20, b"\x00") + b"\x00" + b"\x00" * 31
assert len(GREETING) == 64
FLAG_MORE, FLAG_LONG, FLAG_COMMAND = 1, 2, 4


def _frame(body: bytes, more: bool = False, command: bool = False) -> bytes:
    flags = (FLAG_MORE if more else 0) | (FLAG_COMMAND if command else 0)
    if len(body) > 255:
        return bytes([flags | FLAG_LONG]) + struct.pack(">Q", len(body)) + body
    return bytes([flags, len(body)]) + body


def _ready(socket_type: bytes) -> bytes:
    name = b"Socket-Type"
    body = b"\x05READY" + bytes([len(name)]) + name + struct.pack(">I", len(socket_type)) + socket_type
    return _frame(body, command=True)


class ProtocolError(RuntimeError):
    pass


class _Conn:
    """Incremental parser of one peer's byte stream."""

    def __init__(self, sock: socket.socket):
        self.sock = sock
        self.buf = bytearray()
        self.greeted = False
        self.ready = False
        self.frames: list[bytes] = []  # frames of the message being received

    def feed(self, data: bytes) -> list[list[bytes]]:
        """Returns the complete messages (lists of frame bodies, delimiter removed) contained in the stream so far."""
        self.buf += data
        out = []
        if not self.greeted:
            if len(self.buf) < 64:
                return out
            g = bytes(self.buf[:64])
            del self.buf[:64]
            if g[0] != 0xFF or g[9] != 0x7F or g[10] < 3:
                raise ProtocolError("not a ZMTP 3.x peer")
            if g[12:32].rstrip(b"\x00") != b"NULL":
                raise ProtocolError(f"unsupported security mechanism {g[12:32].rstrip(bytes(1))!r}")
            self.greeted = True
        while True:
            if len(self.buf) < 2:
                return out
            flags = self.buf[0]
            if flags & FLAG_LONG:
                if len(self.buf) < 9:
                    return out
                size, hdr = struct.unpack(">Q", bytes(self.buf[1:9]))[0], 9
            else:
                size, hdr = self.buf[1], 2
            if size > MAX_FRAME_BYTES:  # a peer may not make the receive buffer grow without bound
                raise ProtocolError(f"frame of {size} bytes exceeds the {MAX_FRAME_BYTES}-byte limit")
            if len(self.buf) < hdr + size:
                return out
            body = bytes(self.buf[hdr : hdr + size])
            del self.buf[: hdr + size]
            if flags & FLAG_COMMAND:
                if body[1:6] == b"READY":
                    self.ready = True
                elif body[1:6] == b"ERROR":
                    raise ProtocolError(f"peer sent ERROR: {body[6:]!r}")
                elif body[1:5] == b"PING":  # ZMTP 3.1 heartbeat: answer with PONG + the ping's context
                    ctx = body[7:]
                    self.sock.sendall(_frame(b"\x04PONG" + ctx, command=True))
                continue
            self.frames.append(body)
            if not flags & FLAG_MORE:
                msg, self.frames = self.frames, []
                if b"" in msg:  # REQ envelope: everything up to and including the first empty frame is routing
                    msg = msg[msg.index(b"") + 1 :]
                out.append(msg)


MAX_FRAME_BYTES = 256 << 20  # largest request the reference's clients send is a list of node ids / a handful of texts
CLIENT_IO_TIMEOUT_S = 30.0    # one stalled peer may not freeze the (single-threaded) server for longer than this


class RepServer:
    def __init__(self, port: int, host: str = "0.0.0.0"):
        self.lsock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.lsock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.lsock.bind((host, port))
        self.lsock.listen(16)
        self.lsock.setblocking(False)
        self.port = self.lsock.getsockname()[1]

    def serve(self, handler: Callable[[bytes], bytes], shutdown_event: Optional[threading.Event] = None, poll_s: float = 0.2) -> None:
        """Single-frame request -> handler(bytes) -> single-frame reply, until ``shutdown_event`` is set."""
        shutdown_event = shutdown_event or threading.Event()
        sel = selectors.DefaultSelector()
        sel.register(self.lsock, selectors.EVENT_READ, None)
        try:
            while not shutdown_event.is_set():
                for key, _ in sel.select(timeout=poll_s):
                    if key.data is None:
                        c, _addr = self.lsock.accept()
                        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        c.settimeout(CLIENT_IO_TIMEOUT_S)
                        c.sendall(GREETING + _ready(b"REP"))
                        sel.register(c, selectors.EVENT_READ, _Conn(c))
                        continue
                    conn: _Conn = key.data
                    try:
                        data = conn.sock.recv(1 << 20)
                        if not data:
                            raise ConnectionResetError
                        for msg in conn.feed(data):
                            try:
                                reply = handler(msg[0] if msg else b"")
                            except Exception:  # noqa: BLE001 - a failing request must not end the serve loop: empty reply = the reference servers' error answer
                                import logging

                                logging.getLogger(__name__).exception("request handler failed")
                                reply = b""
                            conn.sock.sendall(_frame(b"", more=True) + _frame(reply))
                    except (OSError, ProtocolError):
                        sel.unregister(conn.sock)
                        conn.sock.close()
        finally:
            for key in list(sel.get_map().values()):
                try:
                    key.fileobj.close()
                except OSError:
                    pass
            sel.close()

    def close(self) -> None:
        try:
            self.lsock.close()
        except OSError:
            pass


class ReqClient:
    """The matching REQ side (tests / probing)."""

    def __init__(self, port: int, host: str = "127.0.0.1", timeout: float = 30.0):
        self.sock = socket.create_connection((host, port), timeout=timeout)
        self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        self.conn = _Conn(self.sock)
        self.sock.sendall(GREETING + _ready(b"REQ"))

    def request(self, payload: bytes) -> bytes:
        self.sock.sendall(_frame(b"", more=True) + _frame(payload))
        while True:
            data = self.sock.recv(1 << 20)
            if not data:
                raise ConnectionResetError("server closed the connection")
            msgs = self.conn.feed(data)
            if msgs:
                return msgs[0][0] if msgs[0] else b""

    def close(self) -> None:
        self.sock.close()

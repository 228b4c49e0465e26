"""A small streaming server over the real-time boundary.

The reference's live front-end is a customtkinter GUI whose sounddevice callback (evaluations/real-time-gui.py:1204-1358) feeds
`custom_infer` one audio block at a time, re-prefilling lazily when the reference or the block size changes (:32-49).  The GUI toolkit
and the sound device are callers of that loop, not part of the engine; this module is the same loop behind a socket, so that any
client (a sound-card bridge, a VoIP hop, a test) can stream blocks through one GPU-resident session:

    python -m streamvoiceanon_amd.stream_server --port 5577 [--config_path ... --checkpoint_path ...]

Protocol (little endian, one TCP connection = one `RealtimeSession`):
    client -> server   frame  = u32 kind | u32 nbytes | payload
        kind 1 REF    payload = utf-8 name, 0 byte, float32 samples at 44.1 kHz     (a new reference: next block re-prefills, :36-47)
        kind 2 CONF   payload = float32 alpha | i32 block_frame | i32 n_frame_delay (the three GUI settings a preset writes, :645-660)
        kind 3 BLOCK  payload = float32[2048 * block_frame] source samples
        kind 4 BYE
    server -> client   for every BLOCK: u32 3 | u32 nbytes | float32[2048 * block_frame] converted samples (zeros while the delay fills)
                       for REF / CONF:  u32 kind | u32 0
                       on error:        u32 0xffffffff | u32 nbytes | utf-8 message (the connection stays open)
    Frame sizes are bounded per kind (CONF = 12 bytes, BLOCK <= 64 frames of 2048 samples, REF <= 60 s of 44.1 kHz audio + a 1 KiB name);
    an oversize or unknown frame is answered with an error and the connection is closed (its payload cannot be skipped safely).

One connection is served at a time per process (the reference's GUI is single-session too); the engine and its weights stay
resident between connections.
"""
import argparse
import socket
import struct

import numpy as np

from .realtime import GuiSettings, RealtimeSession

KIND_REF, KIND_CONF, KIND_BLOCK, KIND_BYE, KIND_ERR = 1, 2, 3, 4, 0xFFFFFFFF
MAX_BLOCK_FRAMES = 64
MAX_PAYLOAD = {KIND_REF: 1024 + 1 + 4 * 44100 * 60, KIND_CONF: 12, KIND_BLOCK: 4 * 2048 * MAX_BLOCK_FRAMES, KIND_BYE: 0}


class ProtocolError(Exception):
    """A frame the server cannot even read past (unknown kind, oversize payload): answered, then the connection is dropped."""


def _recv_exact(conn, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the connection")
        buf += chunk
    return bytes(buf)


def _send(conn, kind, payload=b""):
    conn.sendall(struct.pack("<II", kind, len(payload)) + payload)


def serve_connection(conn, model_set):
    """The audio loop of real-time-gui.py:1313-1330 for one client: every BLOCK goes through RealtimeSession.run_block."""
    sess, settings = RealtimeSession(), GuiSettings()
    ref_name, ref_wav = "", None
    while True:
        kind, nbytes = struct.unpack("<II", _recv_exact(conn, 8))
        if kind not in MAX_PAYLOAD or nbytes > MAX_PAYLOAD[kind]:
            msg = f"frame kind {kind} with {nbytes} bytes refused (limits: {MAX_PAYLOAD})"
            _send(conn, KIND_ERR, ("ProtocolError: " + msg).encode("utf-8"))
            raise ProtocolError(msg)
        payload = _recv_exact(conn, nbytes) if nbytes else b""
        try:
            if kind == KIND_BYE:
                return
            if kind == KIND_REF:
                z = payload.index(b"\0")
                ref_name = payload[:z].decode("utf-8")
                ref_wav = np.frombuffer(payload[z + 1:], dtype="<f4").astype(np.float32)
                if ref_wav.size < 2048 * 3:
                    raise ValueError("reference shorter than three frames")
                _send(conn, KIND_REF)
            elif kind == KIND_CONF:
                alpha, bf, nd = struct.unpack("<fii", payload)
                if not (1 <= bf <= MAX_BLOCK_FRAMES and nd >= 1):
                    raise ValueError(f"block_frame must be 1..{MAX_BLOCK_FRAMES} and n_frame_delay >= 1")
                settings = GuiSettings(alpha=float(alpha), block_frame=int(bf), n_frame_delay=int(nd))
                _send(conn, KIND_CONF)
            elif kind == KIND_BLOCK:
                if ref_wav is None:
                    raise ValueError("send a reference (kind 1) before the first block")
                if len(payload) % 4:
                    raise ValueError("block payload is not a whole number of float32 samples")
                block = np.frombuffer(payload, dtype="<f4").astype(np.float32)
                out = sess.run_block(model_set, ref_wav, ref_name, block, settings)
                _send(conn, KIND_BLOCK, np.ascontiguousarray(out, dtype="<f4").tobytes())
        except (ValueError, AssertionError, RuntimeError, NotImplementedError, struct.error, UnicodeDecodeError) as ex:
            _send(conn, KIND_ERR, f"{type(ex).__name__}: {ex}".encode("utf-8"))


def serve(model_set, host="127.0.0.1", port=5577, max_connections=None, ready=None):
    """Accept connections one after the other.  `ready` (threading.Event-like) is set once the socket listens; the bound port is
    returned through `ready.port` when port = 0."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((host, port))
        srv.listen(1)
        if ready is not None:
            ready.port = srv.getsockname()[1]
            ready.set()
        served = 0
        while max_connections is None or served < max_connections:
            conn, _ = srv.accept()
            with conn:
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                try:
                    serve_connection(conn, model_set)
                except (ConnectionError, ProtocolError):
                    pass
                except Exception as ex:          # a misbehaving client must not take the resident engine down with it
                    print(f"stream_server: connection dropped after {type(ex).__name__}: {ex}", flush=True)
            served += 1


class Client:
    """Minimal client of the protocol (tests, bridges)."""

    def __init__(self, host="127.0.0.1", port=5577):
        self.sock = socket.create_connection((host, port))
        self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)

    def _roundtrip(self, kind, payload):
        _send(self.sock, kind, payload)
        k, n = struct.unpack("<II", _recv_exact(self.sock, 8))
        body = _recv_exact(self.sock, n) if n else b""
        if k == KIND_ERR:
            raise RuntimeError(body.decode("utf-8"))
        return body

    def set_reference(self, name, wav):
        self._roundtrip(KIND_REF, name.encode("utf-8") + b"\0" + np.ascontiguousarray(wav, dtype="<f4").tobytes())

    def configure(self, alpha=0.7, block_frame=1, n_frame_delay=2):
        self._roundtrip(KIND_CONF, struct.pack("<fii", alpha, block_frame, n_frame_delay))

    def convert(self, block):
        return np.frombuffer(self._roundtrip(KIND_BLOCK, np.ascontiguousarray(block, dtype="<f4").tobytes()), dtype="<f4").copy()

    def close(self):
        try:
            _send(self.sock, KIND_BYE)
        finally:
            self.sock.close()


def main(argv=None):
    from .infer_arvc import InferenceWrapper

    ap = argparse.ArgumentParser(description="StreamVoiceAnon streaming server (MI355X engine)")
    ap.add_argument("--config_path", type=str, default="configs/config_firefly_arvcasr_8192_delay0_8.yaml")
    ap.add_argument("--checkpoint_path", type=str, default=None)
    ap.add_argument("--host", type=str, default="127.0.0.1")
    ap.add_argument("--port", type=int, default=5577)
    ap.add_argument("--fp16", action="store_true", help="fp16 AR weights + KV cache (the reference's InferenceWrapper(fp16=True))")
    args = ap.parse_args(argv)
    model_set = InferenceWrapper(args.config_path, args.checkpoint_path, fp16=args.fp16)
    print(f"listening on {args.host}:{args.port}", flush=True)
    serve(model_set, args.host, args.port)


if __name__ == "__main__":
    main()

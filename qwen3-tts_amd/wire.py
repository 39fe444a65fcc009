"""The reference's worker wire protocol for the Vocoder role (src/protocol.rs, src/worker_client.rs).

Two things live here:
  * ctypes bindings of the C codec (include/hipvoc_wire.h, csrc/wire.cpp) that the ``hipvoc-worker``
    process uses;
  * ``WorkerClient``: a host-side mirror of src/worker_client.rs:14-66 (connect / call / framing), built on
    the independent ``msgpack`` package, emitting what rmp-serde's compact ``to_vec`` emits: structs as
    arrays, enums externally tagged, Vec<u8> as an array of ints.  Tests drive the worker with it.
"""
from __future__ import annotations

import ctypes as C
import socket
import struct
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import hipvoc as _hv

VOCODE, PING, INIT, OTHER, VOCODE_BATCH = 0, 1, 2, 3, 4
E_MALFORMED, E_CAP = -1, -2
MAX_MESSAGE = 100_000_000          # src/worker/mod.rs:666

WIRE_EXPORTS = ["hipvoc_wire_parse_request", "hipvoc_wire_parse_vocode_batch", "hipvoc_wire_encode_vocode",
                "hipvoc_wire_encode_vocode_batch", "hipvoc_wire_encode_simple"]

_bound = False


def _lib():
    global _bound
    L = _hv.load_library()
    if not _bound:
        sz, vp = C.c_size_t, C.c_void_p
        L.hipvoc_wire_parse_request.argtypes = [vp, sz, vp, sz, C.POINTER(sz), C.POINTER(C.c_uint64)]
        L.hipvoc_wire_parse_vocode_batch.argtypes = [vp, sz, vp, sz, vp, vp, sz, C.POINTER(sz)]
        L.hipvoc_wire_encode_vocode.restype = sz
        L.hipvoc_wire_encode_vocode.argtypes = [vp, sz, vp, sz]
        L.hipvoc_wire_encode_vocode_batch.restype = sz
        L.hipvoc_wire_encode_vocode_batch.argtypes = [vp, vp, sz, vp, sz]
        L.hipvoc_wire_encode_simple.restype = sz
        L.hipvoc_wire_encode_simple.argtypes = [C.c_int, C.c_char_p, C.c_char_p, vp, sz]
        _bound = True
    return L


# ------------------------------------------------------------------ C codec (what the worker runs)
def parse_request(msg: bytes) -> Tuple[int, bytes, int]:
    """-> (kind, code bytes, n_tokens); kind < 0: malformed."""
    L = _lib()
    buf = np.frombuffer(msg, dtype=np.uint8) if len(msg) else np.zeros(1, dtype=np.uint8)
    clen, nt = C.c_size_t(0), C.c_uint64(0)
    kind = L.hipvoc_wire_parse_request(buf.ctypes.data, len(msg), None, 0, C.byref(clen), C.byref(nt))
    if kind != VOCODE:
        return kind, b"", 0
    out = np.zeros(max(clen.value, 1), dtype=np.uint8)
    kind = L.hipvoc_wire_parse_request(buf.ctypes.data, len(msg), out.ctypes.data, out.size, C.byref(clen), C.byref(nt))
    return kind, out[: clen.value].tobytes(), int(nt.value)


def parse_vocode_batch(msg: bytes) -> Tuple[int, List[bytes], List[int]]:
    L = _lib()
    buf = np.frombuffer(msg, dtype=np.uint8) if len(msg) else np.zeros(1, dtype=np.uint8)
    n = C.c_size_t(0)
    rc = L.hipvoc_wire_parse_vocode_batch(buf.ctypes.data, len(msg), None, 0, None, None, 0, C.byref(n))
    if rc != VOCODE_BATCH:
        return rc, [], []
    B = n.value
    codes = np.zeros(len(msg) + 1, dtype=np.uint8)
    off = np.zeros(B + 1, dtype=np.uint64)
    nts = np.zeros(max(B, 1), dtype=np.uint64)
    rc = L.hipvoc_wire_parse_vocode_batch(buf.ctypes.data, len(msg), codes.ctypes.data, codes.size, off.ctypes.data,
                                          nts.ctypes.data, B, C.byref(n))
    if rc != VOCODE_BATCH:
        return rc, [], []
    return rc, [codes[int(off[j]): int(off[j + 1])].tobytes() for j in range(B)], [int(nts[j]) for j in range(B)]


def encode_vocode_response(pcm: np.ndarray) -> bytes:
    L = _lib()
    pcm = np.ascontiguousarray(pcm, dtype="<i2")
    p = pcm.ctypes.data if pcm.size else None
    need = L.hipvoc_wire_encode_vocode(p, pcm.size, None, 0)
    out = np.zeros(need, dtype=np.uint8)
    got = L.hipvoc_wire_encode_vocode(p, pcm.size, out.ctypes.data, out.size)
    assert got == need
    return out.tobytes()


def encode_vocode_batch_response(pcms: Sequence[np.ndarray]) -> bytes:
    L = _lib()
    arrs = [np.ascontiguousarray(p, dtype="<i2") for p in pcms]
    so = np.zeros(len(arrs) + 1, dtype=np.uint64)
    if arrs:
        so[1:] = np.cumsum([a.size for a in arrs])
    cat = np.concatenate(arrs) if arrs and so[-1] else np.zeros(1, dtype="<i2")
    need = L.hipvoc_wire_encode_vocode_batch(cat.ctypes.data, so.ctypes.data, len(arrs), None, 0)
    out = np.zeros(need, dtype=np.uint8)
    L.hipvoc_wire_encode_vocode_batch(cat.ctypes.data, so.ctypes.data, len(arrs), out.ctypes.data, out.size)
    return out.tobytes()


def encode_simple_response(ok: bool, variant: str = "Init", error: Optional[str] = None) -> bytes:
    L = _lib()
    e = error.encode() if error is not None else None
    need = L.hipvoc_wire_encode_simple(int(ok), variant.encode(), e, None, 0)
    out = np.zeros(need, dtype=np.uint8)
    L.hipvoc_wire_encode_simple(int(ok), variant.encode(), e, out.ctypes.data, out.size)
    return out.tobytes()


# ------------------------------------------------------------------ client (mirror of worker_client.rs)
def _pack(obj) -> bytes:
    import msgpack
    return msgpack.packb(obj, use_bin_type=True)


def request_ping() -> bytes:
    return _pack("Ping")


def request_init(role: str, models_dir: str) -> bytes:
    return _pack({"Init": [role, models_dir]})


def request_vocode(codes: np.ndarray, n_tokens: Optional[int] = None) -> bytes:
    """Request::Vocode (src/protocol.rs:51-56; built as in pipeline.rs via encode_i64, worker_client.rs:80-83)."""
    c = np.ascontiguousarray(codes, dtype="<i8").reshape(-1, 16)
    n = c.shape[0] if n_tokens is None else n_tokens
    return _pack({"Vocode": [list(c.tobytes()), n]})


def request_vocode_batch(seqs: Sequence[np.ndarray]) -> bytes:
    arrs = [np.ascontiguousarray(s, dtype="<i8").reshape(-1, 16) for s in seqs]
    return _pack({"VocodeBatch": [[list(a.tobytes()) for a in arrs], [a.shape[0] for a in arrs]]})


class WorkerError(RuntimeError):
    pass


class WorkerClient:
    def __init__(self, host: str, port: int, timeout: float = 120.0):
        self.sock = socket.create_connection((host, port), timeout=timeout)
        self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)

    def close(self):
        try:
            self.sock.close()
        except OSError:
            pass

    def send_raw(self, payload: bytes):
        self.sock.sendall(struct.pack(">I", len(payload)) + payload)

    def _read_exact(self, n: int) -> bytes:
        chunks = []
        while n:
            b = self.sock.recv(min(n, 1 << 20))
            if not b:
                raise ConnectionError("worker disconnected")
            chunks.append(b)
            n -= len(b)
        return b"".join(chunks)

    def read_response(self):
        """-> (status, data, error) as decoded msgpack objects (Response = [status, data, error])."""
        import msgpack
        (n,) = struct.unpack(">I", self._read_exact(4))
        obj = msgpack.unpackb(self._read_exact(n), raw=False, strict_map_key=False)
        if isinstance(obj, dict):
            return obj["status"], obj["data"], obj.get("error")
        return obj[0], obj[1], obj[2] if len(obj) > 2 else None

    def call(self, payload: bytes):
        self.send_raw(payload)
        status, data, err = self.read_response()
        if status != "ok":
            raise WorkerError(err or "unknown")
        return data

    def ping(self) -> bool:
        return self.call(request_ping()) == "Ping"

    def vocode(self, codes: np.ndarray) -> np.ndarray:
        data = self.call(request_vocode(codes))
        audio, n_samples = data["Vocode"]
        pcm = np.frombuffer(bytes(audio), dtype="<i2")
        assert pcm.size == n_samples
        return pcm

    def vocode_batch(self, seqs: Sequence[np.ndarray]) -> List[np.ndarray]:
        data = self.call(request_vocode_batch(seqs))
        audio, n_samples = data["VocodeBatch"]
        out = [np.frombuffer(bytes(a), dtype="<i2") for a in audio]
        assert [o.size for o in out] == list(n_samples)
        return out

    def vocode_streamed(self, codes: np.ndarray, chunk_size: int = 64) -> np.ndarray:
        """The orchestrator's vocoder phase (src/pipeline.rs:586-607, 636-673): one Vocode request per full
        ``chunk_size`` tokens is SENT as soon as the tokens exist, the remainder is queued last, and only then are the
        responses read, in submission order, and their samples concatenated -- several requests are in flight on the
        connection at once."""
        codes = np.ascontiguousarray(codes, dtype=np.int64).reshape(-1, 16)
        n = codes.shape[0]
        sent = 0
        pending = 0
        while n - sent >= chunk_size:                       # "send partial batch when we have enough tokens"
            self.send_raw(request_vocode(codes[sent: sent + chunk_size]))
            sent += chunk_size
            pending += 1
        if sent < n:                                        # "queue remaining tail"
            self.send_raw(request_vocode(codes[sent:]))
            pending += 1
        parts = []
        for _ in range(pending):                            # "collect responses in submission order"
            status, data, err = self.read_response()
            if status != "ok":
                raise WorkerError(err or "unknown")
            audio, n_samples = data["Vocode"]
            pcm = np.frombuffer(bytes(audio), dtype="<i2")
            assert pcm.size == n_samples
            parts.append(pcm)
        return np.concatenate(parts) if parts else np.zeros(0, dtype=np.int16)


class VocoderQueue:
    """Host-side mirror of integration/pipeline_vocode_batch.rs (the orchestrator's batched vocoder requests, SURVEY.md
    8f-1): one thread owns the vocoder connection; ``submit`` never blocks and returns a future; whenever the connection is
    free, EVERYTHING that is waiting -- chunks of one utterance and of other concurrent utterances -- goes out as one
    ``VocodeBatch`` (a lone submission as the reference's plain ``Vocode``).  Callers keep the order of their own audio by
    waiting for their futures in submission order, as pipeline.rs:659-673 reads its responses."""

    def __init__(self, host: str, port: int, max_batch: int = 256):
        import queue
        import threading
        self._q = queue.Queue()
        self._client = WorkerClient(host, port)
        self._max = max_batch
        self.batches_sent = 0
        self.singles_sent = 0
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()

    def submit(self, codes: np.ndarray):
        from concurrent.futures import Future
        f = Future()
        self._q.put((np.ascontiguousarray(codes, dtype=np.int64).reshape(-1, 16), f))
        return f

    def close(self):
        self._q.put(None)
        self._th.join(timeout=30)
        self._client.close()

    def _run(self):
        import queue
        while True:
            item = self._q.get()
            if item is None:
                return
            batch = [item]
            stop = False
            while len(batch) < self._max:
                try:
                    nxt = self._q.get_nowait()
                except queue.Empty:
                    break
                if nxt is None:
                    stop = True
                    break
                batch.append(nxt)
            try:
                if len(batch) == 1:
                    self.singles_sent += 1
                    batch[0][1].set_result(self._client.vocode(batch[0][0]))
                else:
                    self.batches_sent += 1
                    for (_, f), pcm in zip(batch, self._client.vocode_batch([c for c, _ in batch])):
                        f.set_result(pcm)
            except Exception as e:  # noqa: BLE001
                for _, f in batch:
                    if not f.done():
                        f.set_exception(e)
            if stop:
                return

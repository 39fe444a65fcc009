"""The worker wire codec (include/hipvoc_wire.h, csrc/wire.cpp) against an INDEPENDENT msgpack implementation
(the `msgpack` package) and hand-assembled known-answer bytes.

What the expected bytes encode is rmp-serde 1.3's compact `to_vec` form of src/protocol.rs's types as reasoned
from the MessagePack spec and the crate's documented behaviour (struct = array, externally tagged enums, unit
variant = str, Vec<u8> = array of ints, None = nil).  There is no Rust toolchain in the image, so agreement with
rmp-serde itself is NOT pinned here ("wire parity unpinned").  No GPU needed: the codec is host code.
"""
import struct

import msgpack
import numpy as np
import pytest


@pytest.fixture(scope="module")
def wire(pkg):
    return pkg.wire


def codes_bytes(n, seed=0):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 2048, size=(n, 16), dtype=np.int64).astype("<i8").tobytes()


def test_known_answer_bytes(wire):
    # Request::Ping -> fixstr "Ping"
    assert wire.request_ping() == bytes([0xA4]) + b"Ping"
    assert wire.parse_request(bytes([0xA4]) + b"Ping")[0] == wire.PING
    # Response{ok, Ping, None} -> [ "ok", "Ping", nil ]
    assert wire.encode_simple_response(True, "Ping") == bytes([0x93, 0xA2]) + b"ok" + bytes([0xA4]) + b"Ping" + bytes([0xC0])
    # Response{error, Init, Some("x")}  (src/worker/mod.rs:251-255)
    assert wire.encode_simple_response(False, "Init", "x") == bytes([0x93, 0xA5]) + b"error" + bytes([0xA4]) + b"Init" + bytes([0xA1]) + b"x"
    # Request::Vocode{codes:[1,0x80,0xff], n_tokens:0} -> {"Vocode": [[1, cc 80, cc ff], 0]}
    msg = bytes([0x81, 0xA6]) + b"Vocode" + bytes([0x92, 0x93, 0x01, 0xCC, 0x80, 0xCC, 0xFF, 0x00])
    assert wire.parse_request(msg) == (wire.VOCODE, bytes([1, 0x80, 0xFF]), 0)
    # ResponseData::Vocode{audio:[0x00,0x80] (one sample -32768), n_samples:1}
    got = wire.encode_vocode_response(np.array([-32768], dtype=np.int16))
    assert got == bytes([0x93, 0xA2]) + b"ok" + bytes([0x81, 0xA6]) + b"Vocode" + bytes([0x92, 0x92, 0x00, 0xCC, 0x80, 0x01, 0xC0])


@pytest.mark.parametrize("n", [0, 1, 7, 125])
def test_encoder_matches_independent_msgpack(wire, n):
    rng = np.random.default_rng(n)
    pcm = rng.integers(-32768, 32768, size=n * 1920 if n < 100 else 4000, dtype=np.int16)
    ours = wire.encode_vocode_response(pcm)
    ref = msgpack.packb(["ok", {"Vocode": [list(pcm.astype("<i2").tobytes()), int(pcm.size)]}, None])
    assert ours == ref
    status, data, err = msgpack.unpackb(ours, raw=False)
    assert status == "ok" and err is None and data["Vocode"][1] == pcm.size
    assert bytes(data["Vocode"][0]) == pcm.astype("<i2").tobytes()


def test_batch_encoder_matches_independent_msgpack(wire):
    rng = np.random.default_rng(5)
    pcms = [rng.integers(-32768, 32768, size=k, dtype=np.int16) for k in (3, 0, 70000)]
    ours = wire.encode_vocode_batch_response(pcms)
    ref = msgpack.packb(["ok", {"VocodeBatch": [[list(p.astype("<i2").tobytes()) for p in pcms], [int(p.size) for p in pcms]]}, None])
    assert ours == ref


@pytest.mark.parametrize("n", [0, 1, 64, 125])
def test_parser_accepts_every_form_serde_accepts(wire, n):
    cb = codes_bytes(n, n)
    forms = [
        msgpack.packb({"Vocode": [list(cb), n]}),                                   # compact (what to_vec emits)
        msgpack.packb({"Vocode": {"codes": list(cb), "n_tokens": n}}),              # named (to_vec_named)
        msgpack.packb({"Vocode": {"n_tokens": n, "codes": list(cb)}}),              # field order is free in a map
        msgpack.packb({"Vocode": [cb, n]}, use_bin_type=True),                       # serde_bytes / bin
    ]
    assert forms[0] == wire.request_vocode(np.frombuffer(cb, dtype="<i8"))
    for f in forms:
        assert wire.parse_request(f) == (wire.VOCODE, cb, n)


def test_parser_classifies_other_roles_and_rejects_garbage(wire):
    assert wire.parse_request(wire.request_init("vocoder", "/models"))[0] == wire.INIT
    other = [
        {"TalkerStep": [None, 0.9, 1.05, 0.0]},
        {"TalkerStep": [list(b"\x00" * 16), 0.9, 1.05, 0.0]},
        {"CodePredict": [list(b"\x01" * 8), 5, 0.8]},
        {"TokenizeAndEmbed": ["hello", "english", None]},
        {"TokenizeAndEmbed": ["hello", "english", {"Inline": list(b"\x02" * 8)}]},
        {"LoadVoice": [list(b"\x00" * 128), 1, "ref"]},
        {"TalkerPrefill": [list(b"\x00" * 64)]},
    ]
    for o in other:
        assert wire.parse_request(msgpack.packb(o))[0] == wire.OTHER, o
    bad = [b"", b"\xc1", msgpack.packb("Nope"), msgpack.packb({"Nope": [1]}), msgpack.packb({"Vocode": [1, 2]}),
           msgpack.packb({"Vocode": [[300], 1]}), msgpack.packb({"Vocode": [[1, 2]]}), msgpack.packb(["Vocode"]),
           msgpack.packb({"Vocode": [[1], 1], "Ping": None}), msgpack.packb({"Vocode": [[1], -1]}),
           msgpack.packb("Ping") + b"\x00", msgpack.packb({"Vocode": [[1], 1]}) + b"\x00", msgpack.packb("Vocode")]
    for b in bad:
        assert wire.parse_request(b)[0] == wire.E_MALFORMED, b


def test_parser_survives_truncation_and_bit_flips(wire):
    msg = wire.request_vocode(np.frombuffer(codes_bytes(5, 1), dtype="<i8"))
    for cut in range(len(msg)):
        assert wire.parse_request(msg[:cut])[0] == wire.E_MALFORMED
    rng = np.random.default_rng(0)
    for _ in range(2000):
        m = bytearray(msg)
        for _ in range(int(rng.integers(1, 4))):
            m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
        kind, cb, n = wire.parse_request(bytes(m))      # must not crash; any verdict is fine
        assert kind in (wire.VOCODE, wire.PING, wire.INIT, wire.OTHER, wire.VOCODE_BATCH, wire.E_MALFORMED)
    # a huge declared array length with no data behind it must not allocate / crash
    assert wire.parse_request(bytes([0x81, 0xA6]) + b"Vocode" + bytes([0x92, 0xDD, 0xFF, 0xFF, 0xFF, 0xFF]))[0] == wire.E_MALFORMED


def test_vocode_batch_round_trip(wire):
    seqs = [np.frombuffer(codes_bytes(n, n), dtype="<i8").reshape(-1, 16) for n in (3, 0, 10)]
    msg = wire.request_vocode_batch(seqs)
    assert wire.parse_request(msg)[0] == wire.VOCODE_BATCH
    rc, codes, nts = wire.parse_vocode_batch(msg)
    assert rc == wire.VOCODE_BATCH and nts == [3, 0, 10]
    assert codes == [s.astype("<i8").tobytes() for s in seqs]
    assert wire.parse_vocode_batch(wire.request_ping())[0] == wire.E_MALFORMED
    assert wire.parse_vocode_batch(msgpack.packb({"VocodeBatch": [[[1]], [1, 2]]}))[0] == wire.E_MALFORMED


def test_framing_constants(wire):
    assert wire.MAX_MESSAGE == 100_000_000                       # src/worker/mod.rs:666
    assert struct.pack(">I", 5) == b"\x00\x00\x00\x05"          # u32 big-endian length (worker_client.rs:34-35)


def test_wire_codec_survives_fuzzing_under_asan_ubsan(tmp_path):
    """tests/native/fuzz_wire.cpp: truncations and ~200k mutations of valid requests plus noise through both parsers
    and the encoders, built with -fsanitize=address,undefined (host code only; the worker feeds these parsers bytes
    straight from a socket)."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "fuzz_wire")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                        os.path.join(root, "tests", "native", "fuzz_wire.cpp"),
                        os.path.join(root, "qwen3-tts_amd", "csrc", "wire.cpp")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fuzz_wire OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_shared_known_answers_fixture(wire):
    """tests/golden/wire_kat.json: the same bytes integration/wire_kat_test.rs asserts against rmp-serde on a box with
    cargo.  Here: the C codec produces / parses them, and the independent msgpack package agrees."""
    import json
    import os
    kat = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wire_kat.json")))
    b = {k: bytes.fromhex(v) for k, v in kat.items() if not k.startswith("_")}
    assert wire.request_ping() == b["request_ping"]
    assert wire.request_init("vocoder", "/models") == b["request_init_vocoder_models"]
    assert wire.parse_request(b["request_ping"])[0] == wire.PING
    assert wire.parse_request(b["request_init_vocoder_models"])[0] == wire.INIT
    assert wire.parse_request(b["request_vocode_codes_01_80_ff_n0"]) == (wire.VOCODE, bytes([1, 0x80, 0xFF]), 0)
    assert wire.encode_simple_response(True, "Ping") == b["response_ok_ping"]
    assert wire.encode_simple_response(False, "Init", "x") == b["response_error_init_x"]
    assert wire.encode_vocode_response(np.array([-32768], dtype=np.int16)) == b["response_ok_vocode_audio_00_80_n1"]
    assert msgpack.packb("Ping") == b["request_ping"]
    assert msgpack.packb({"Init": ["vocoder", "/models"]}) == b["request_init_vocoder_models"]
    assert msgpack.packb({"Vocode": [[1, 0x80, 0xFF], 0]}) == b["request_vocode_codes_01_80_ff_n0"]
    assert msgpack.packb(["ok", "Ping", None]) == b["response_ok_ping"]
    assert msgpack.packb(["error", "Init", "x"]) == b["response_error_init_x"]
    assert msgpack.packb(["ok", {"Vocode": [[0, 0x80], 1]}, None]) == b["response_ok_vocode_audio_00_80_n1"]
    # the VocodeBatch extension (appended variants)
    rc, codes, nts = wire.parse_vocode_batch(b["request_vocode_batch_codes_0180_ff_n00"])
    assert rc == wire.VOCODE_BATCH and codes == [bytes([1, 0x80]), bytes([0xFF])] and nts == [0, 0]
    assert wire.parse_request(b["request_vocode_batch_codes_0180_ff_n00"])[0] == wire.VOCODE_BATCH
    got = wire.encode_vocode_batch_response([np.array([-32768], dtype=np.int16), np.zeros(0, dtype=np.int16)])
    assert got == b["response_ok_vocode_batch_audio_0080_empty_n10"]


def test_hostile_item_counts_allocate_nothing(wire):
    """a VocodeBatch whose array32 header claims 1e8 items must not make the parser reserve memory for them;
    a well-formed batch of a million EMPTY items is counted, and refused by max_items, without per-item allocations."""
    import ctypes as C
    import resource
    L = wire._lib()
    head = b"\x81" + msgpack.packb("VocodeBatch") + b"\x92"
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    lying = head + b"\xdd" + struct.pack(">I", 100_000_000) + b"\x90" * 8
    assert wire.parse_vocode_batch(lying)[0] == wire.E_MALFORMED
    n_items = 1_000_000
    many = head + b"\xdd" + struct.pack(">I", n_items) + b"\x90" * n_items + b"\xdd" + struct.pack(">I", n_items) + b"\x00" * n_items
    buf = np.frombuffer(many, dtype=np.uint8)
    n = C.c_size_t(0)
    assert L.hipvoc_wire_parse_vocode_batch(buf.ctypes.data, len(many), None, 0, None, None, 0, C.byref(n)) == wire.VOCODE_BATCH
    assert n.value == n_items
    off = np.zeros(9, dtype=np.uint64)
    nts = np.zeros(8, dtype=np.uint64)
    out = np.zeros(16, dtype=np.uint8)
    rc = L.hipvoc_wire_parse_vocode_batch(buf.ctypes.data, len(many), out.ctypes.data, out.size, off.ctypes.data, nts.ctypes.data, 8, C.byref(n))
    assert rc == wire.E_CAP and n.value == n_items
    after = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert after - before < 64 * 1024, f"parser grew the process by {(after - before) / 1024:.0f} MiB"      # ru_maxrss is in KiB
    # capacity: codes that do not fit are refused, never written past the buffer
    seqs = [np.frombuffer(codes_bytes(2, 1), dtype="<i8").reshape(-1, 16)] * 3
    msg = np.frombuffer(wire.request_vocode_batch(seqs), dtype=np.uint8)
    small = np.full(300, 0xEE, dtype=np.uint8)
    rc = L.hipvoc_wire_parse_vocode_batch(msg.ctypes.data, msg.size, small.ctypes.data, 256, off.ctypes.data, nts.ctypes.data, 8, C.byref(n))
    assert rc == wire.E_CAP and np.all(small[256:] == 0xEE)

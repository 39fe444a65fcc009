"""Voice-profile files as code sources (qwen3-tts_amd/voice.py mirrors src/voice_loader.rs:7-60)."""
import json
import os

import numpy as np
import pytest


def test_json_roundtrip_and_ref_text(pkg, synth, tmp_path):
    codes = synth.make_codes(37, 5, synth.FULL)
    p = str(tmp_path / "v.json")
    pkg.voice.save_voice_json(p, codes, "hello there")
    got, text = pkg.voice.load_voice_file(p)
    assert text == "hello there"
    assert got.dtype == np.int64 and got.shape == (37, 16)
    assert np.array_equal(got, codes)
    # no ref_text key / non-string ref_text -> None (voice_loader.rs:12 `.as_str()`)
    with open(p, "w") as f:
        json.dump({"codec_tokens": codes.tolist(), "ref_text": 3}, f)
    assert pkg.voice.load_voice_file(p)[1] is None


def test_json_flatten_ignores_row_shape_but_needs_whole_frames(pkg, tmp_path):
    # the reference flattens the nested lists without checking row lengths (voice_loader.rs:15)
    p = str(tmp_path / "v.json")
    with open(p, "w") as f:
        json.dump({"codec_tokens": [list(range(8)), list(range(8, 32))]}, f)
    got, _ = pkg.voice.load_voice_file(p)
    assert got.shape == (2, 16) and np.array_equal(got.reshape(-1), np.arange(32))
    for bad in ({"codec_tokens": [[1, 2, 3]]}, {"codec_tokens": [[1.5] * 16]}, {"codec_tokens": "x"}, {"tokens": []}, [1]):
        with open(p, "w") as f:
            json.dump(bad, f)
        with pytest.raises(ValueError):
            pkg.voice.load_voice_file(p)


def test_npy_auto_transpose(pkg, synth, tmp_path):
    codes = synth.make_codes(40, 9, synth.FULL)
    a, b = str(tmp_path / "a.npy"), str(tmp_path / "b.npy")
    np.save(a, codes)
    np.save(b, np.ascontiguousarray(codes.T))          # [16, N], N > 16: transposed back (voice_loader.rs:21-26)
    for p in (a, b):
        got, text = pkg.voice.load_voice_file(p)
        assert text is None and np.array_equal(got, codes)
    sq = synth.make_codes(16, 2, synth.FULL)             # [16, 16] is taken as [N, 16] (ncols > 16 is false)
    np.save(a, sq)
    assert np.array_equal(pkg.voice.load_voice_file(a)[0], sq)
    np.save(a, codes.astype(np.int32))
    with pytest.raises(ValueError):
        pkg.voice.load_voice_file(a)


def test_pt_and_unknown_refused(pkg, tmp_path):
    for name in ("v.pt", "v.bin"):
        p = str(tmp_path / name)
        open(p, "wb").write(b"\x80\x02")
        with pytest.raises(ValueError):
            pkg.voice.load_voice_file(p)


def test_read_wav_formats(pkg, tmp_path):
    """The WAV reader behind encode_wav (speech_tokenizer.rs:681-696 via hound): 16 / 24 / 32-bit PCM scaled by
    2^(bits-1), 8-bit unsigned, float32 as is; stereo stays interleaved; junk chunks are skipped."""
    import struct
    import wave
    read_wav = pkg.encoder.read_wav
    rng = np.random.default_rng(1)
    x16 = rng.integers(-32768, 32767, size=480, dtype=np.int16)
    p = str(tmp_path / "a.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000); w.writeframes(x16.tobytes())
    s, rate, ch = read_wav(p)
    assert (rate, ch) == (16000, 2) and s.dtype == np.float32
    assert np.array_equal(s, x16.astype(np.float32) / np.float32(32768))

    def riff(fmt_tag, channels, rate, bits, payload, extra=b""):
        fmt = struct.pack("<HHIIHH", fmt_tag, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits)
        body = b"WAVE" + extra + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(payload)) + payload
        return b"RIFF" + struct.pack("<I", len(body)) + body
    f32 = rng.standard_normal(100).astype("<f4")
    open(p, "wb").write(riff(3, 1, 24000, 32, f32.tobytes(), extra=b"LIST" + struct.pack("<I", 3) + b"abc\x00"))
    s, rate, ch = read_wav(p)
    assert (rate, ch) == (24000, 1) and np.array_equal(s, f32)
    v24 = rng.integers(-(1 << 23), (1 << 23) - 1, size=50)
    b24 = b"".join(int(v & 0xFFFFFF).to_bytes(3, "little") for v in v24)
    open(p, "wb").write(riff(1, 1, 48000, 24, b24))
    s, rate, _ = read_wav(p)
    assert rate == 48000 and np.array_equal(s, (v24.astype(np.float32) / np.float32(1 << 23)))
    v32 = rng.integers(-(1 << 31), (1 << 31) - 1, size=50).astype("<i4")
    open(p, "wb").write(riff(1, 1, 8000, 32, v32.tobytes()))
    assert np.array_equal(read_wav(p)[0], v32.astype(np.float32) / np.float32(2 ** 31))
    u8 = rng.integers(0, 255, size=40, dtype=np.uint8)
    open(p, "wb").write(riff(1, 1, 8000, 8, u8.tobytes()))
    assert np.array_equal(read_wav(p)[0], (u8.astype(np.int32) - 128).astype(np.float32) / np.float32(128))
    for bad in (b"RIFX" + b"\0" * 40, riff(2, 1, 8000, 4, b"\0" * 8), riff(1, 0, 8000, 16, b"\0" * 8), b"RIFF\0\0\0\0WAVE"):
        open(p, "wb").write(bad)
        with pytest.raises(ValueError):
            read_wav(p)

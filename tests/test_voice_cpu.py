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

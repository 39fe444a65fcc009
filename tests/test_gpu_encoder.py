"""GPU parity tests of the speech-tokenizer ENCODER path through the C ABI
(include/hipvoc_encoder.h) against oracle/enc_oracle.c.

Bars:
  * floating-point stages (SEANet convs, transformer, downsample, input projections): max-abs error <= 1e-3 of the
    stage's peak and SNR >= 60 dB against the fp32 oracle, the vocoder path's fp32 gates;
  * the integer part -- nearest-codeword search -- is BIT-EXACT: the oracle's quantiser applied to the GPU's own
    latent must give exactly the GPU's codes (same distance expression, k-ascending fmaf chains, first minimum);
  * end to end against the oracle's codes: a code can only differ where the two candidates' distances are within
    round-off of each other (a near-tie flipped by the 1e-6 differences upstream); every differing frame is checked
    for that, and the agreement rate is stated.
"""
import os
import tempfile

import numpy as np
import pytest

from conftest import enc_oracle_for, enc_weights_for, load_golden, snr_db

pytestmark = pytest.mark.gpu

_ENC = {}


@pytest.fixture(scope="module", autouse=True)
def _torch_hip_first():
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    yield


def enc_for(pkg, synth, cfg_name):
    if cfg_name not in _ENC:
        cfg = getattr(synth, cfg_name)
        path = os.path.join(tempfile.gettempdir(), f"hipenc_test_{cfg_name}_0.safetensors")
        synth.save_safetensors(path, enc_weights_for(synth, cfg_name, 0), cfg.metadata())
        _ENC[cfg_name] = pkg.HipSpeechTokenizer.load(path, device=0)
    return _ENC[cfg_name]


def check_codes_against_oracle(o, cfg, audio, got, ids):
    """got == oracle codes except at near-ties; returns the agreement rate."""
    ref = o.encode(audio)
    assert got.shape == ref.shape
    agree = float((got == ref).all(axis=1).mean()) if len(ref) else 1.0
    bad = np.nonzero((got != ref).any(axis=1))[0]
    if bad.size:
        psem, pac = o.stage(audio, ids["proj_sem"]), o.stage(audio, ids["proj_ac"])
        n_sem = min(cfg.n_semantic, got.shape[1])
        for t in bad:
            q = int(np.nonzero(got[t] != ref[t])[0][0])          # first differing layer of the frame
            group, layer = (0, q) if q < n_sem else (1, q - n_sem)
            x = (psem if group == 0 else pac)[:, t].astype(np.float32).copy()
            first = 0 if group == 0 else n_sem
            for l in range(layer):                               # residual before the differing layer (same codes so far)
                x = x - o.embed(group, l)[ref[t, first + l]]
            d = o.distances(group, layer, x)
            gap = float(d[got[t, q]] - d[ref[t, q]])
            assert gap <= 1e-3 * max(1.0, float(abs(d[ref[t, q]]))), (t, q, gap)
    return agree


def stage_ids(cfg):
    nb = len(cfg.ratios)
    return {"seanet": 1 + 2 * nb, "transformer": 1 + 2 * nb + cfg.num_layers, "latent": 2 + 2 * nb + cfg.num_layers,
            "proj_sem": 3 + 2 * nb + cfg.num_layers, "proj_ac": 4 + 2 * nb + cfg.num_layers}


@pytest.mark.parametrize("cfg_name,n", [("ENC_SMALL", 1920 * 9 + 300), ("ENC_ODD", 1920 * 6 + 1)])
def test_every_stage_matches_the_oracle(pkg, synth, eo, cfg_name, n):
    cfg = getattr(synth, cfg_name)
    e = enc_for(pkg, synth, cfg_name)
    o = enc_oracle_for(eo, synth, cfg_name, 0)
    audio = synth.make_audio(n, 17)
    assert e.n_stages == o.n_stages
    for s in range(e.n_stages):
        ref = o.stage(audio, s)
        got = e.stage(audio, s)
        assert got.shape == ref.shape, s
        peak = max(1.0, float(np.abs(ref).max()))
        assert np.abs(got - ref).max() <= 1e-3 * peak, (s, float(np.abs(got - ref).max()))
        assert snr_db(ref, got) >= 60.0, (s, snr_db(ref, got))


@pytest.mark.parametrize("cfg_name", ["ENC_SMALL", "ENC_ODD"])
def test_quantiser_is_bit_exact_on_the_gpu_latent(pkg, synth, eo, cfg_name):
    cfg = getattr(synth, cfg_name)
    e = enc_for(pkg, synth, cfg_name)
    o = enc_oracle_for(eo, synth, cfg_name, 0)
    audio = synth.make_audio(1920 * 50 + 77, 23)
    codes = e.encode(audio)
    assert codes.dtype == np.int64 and codes.shape == (50, min(16, cfg.n_semantic + cfg.n_acoustic))
    lat = e.stage(audio, stage_ids(cfg)["latent"])
    assert np.array_equal(o.quantize(lat), codes)
    assert codes.min() >= 0 and codes.max() < cfg.codebook_size


@pytest.mark.parametrize("cfg_name", ["ENC_SMALL", "ENC_ODD"])
def test_codes_match_oracle_up_to_near_ties(pkg, synth, eo, cfg_name):
    cfg = getattr(synth, cfg_name)
    e = enc_for(pkg, synth, cfg_name)
    o = enc_oracle_for(eo, synth, cfg_name, 0)
    audio = synth.make_audio(1920 * 60, 31)
    agree = check_codes_against_oracle(o, cfg, audio, e.encode(audio), stage_ids(cfg))
    assert agree >= 0.9, agree


def test_matches_upstream_mimi_golden(pkg, synth, eo):
    """The committed vectors of transformers' MimiModel (tools/gen_encoder_golden.py) against the GPU path."""
    g = load_golden("encoder_mimi_small")
    cfg = synth.ENC_SMALL
    e = enc_for(pkg, synth, "ENC_SMALL")
    ids = stage_ids(cfg)
    for name in ("seanet", "transformer", "latent"):
        got = e.stage(g["audio"], ids[name])
        assert np.abs(got - g[name]).max() <= 1e-3 * max(1.0, np.abs(g[name]).max()), name
    o = enc_oracle_for(eo, synth, "ENC_SMALL", 0)
    got = e.encode(g["audio"])
    if not np.array_equal(got, g["codes"]):
        check_codes_against_oracle(o, cfg, g["audio"], got, ids)


def test_lengths_errors_and_determinism(pkg, synth, eo):
    cfg = synth.ENC_SMALL
    e = enc_for(pkg, synth, "ENC_SMALL")
    assert e.samples_per_frame == 1920 and e.n_codebooks == 16
    for n in (1920, 1921, 3839, 3840, 5000):
        a = synth.make_audio(n, n)
        c1 = e.encode(a)
        assert c1.shape == (n // 1920, 16) and e.n_frames(n) == n // 1920
        assert np.array_equal(c1, e.encode(a))
    with pytest.raises(pkg.HipVocError):
        e.encode(synth.make_audio(1919, 0))
    with pytest.raises(pkg.HipVocError):
        e.encode(np.zeros(0, dtype=np.float32))
    # the handle keeps working after an error, and the vocoder entry points refuse it
    assert e.encode(synth.make_audio(1920 * 2, 1)).shape == (2, 16)
    out = np.empty(16, dtype=np.float32)
    codes = np.zeros(16, dtype=np.int64)
    import ctypes as C
    n = C.c_size_t(0)
    assert e._L.hipvoc_decode(e._h, codes.ctypes.data, 1, out.ctypes.data, out.size, C.byref(n)) < 0


def test_batch_is_bitwise_equal_to_single_clips_and_causal(pkg, synth):
    e = enc_for(pkg, synth, "ENC_SMALL")
    clips = [synth.make_audio(n, 100 + i) for i, n in enumerate((1920 * 7 + 5, 1920 * 3, 1920 * 12 + 1900, 1920))]
    outs = e.encode_batch(clips)
    for a, got in zip(clips, outs):
        assert np.array_equal(got, e.encode(a))
    # causality: a prefix's frames do not change when audio is appended
    long = clips[2]
    assert np.array_equal(e.encode(long[: 1920 * 5]), outs[2][:5])


def test_encode_samples_mono_mix_and_resample(pkg, synth, eo):
    e = enc_for(pkg, synth, "ENC_SMALL")
    rng = np.random.default_rng(5)
    x16 = (0.3 * rng.standard_normal(16000 * 2)).astype(np.float32)
    assert np.array_equal(pkg.encoder.resample(x16, 16000, 24000), eo.resample(x16, 16000, 24000))
    stereo = np.stack([x16, -0.5 * x16], axis=1).reshape(-1)
    mono = ((x16 + (-0.5 * x16)) / np.float32(2)).astype(np.float32)
    want = e.encode(eo.resample(mono, 16000, 24000))
    assert np.array_equal(e.encode_samples(stereo, 16000, channels=2), want)


def test_encode_then_decode_round_trip_runs(pkg, synth):
    """Encoder output is valid vocoder input ([n,16] int64 in range): the two halves of the tokenizer compose.  (With
    random weights the reconstruction is meaningless; this checks shapes, ranges and lengths only.)"""
    cfg = synth.ENC_FULL
    path = os.path.join(tempfile.gettempdir(), "hipenc_test_ENC_FULL_0.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, synth.make_encoder_weights(cfg, 0), cfg.metadata())
    e = pkg.HipSpeechTokenizer.load(path, device=0)
    audio = synth.make_audio(24000 * 3, 9)
    codes = e.encode(audio)
    assert codes.shape == (37, 16) and codes.min() >= 0 and codes.max() < 2048
    dpath = os.path.join(tempfile.gettempdir(), "hipvoc_test_FULL_0.safetensors")
    if not os.path.exists(dpath):
        synth.save_safetensors(dpath, synth.make_weights(synth.FULL, 0), synth.FULL.metadata())
    v = pkg.HipVocoder.load(dpath, device=0, unit_frames=64)
    wav = v.synthesize(codes)
    assert wav.size == 37 * 1920 and np.isfinite(wav).all()


def test_full_size_model_parity_and_device_entry_point(pkg, synth, eo):
    """The real dimensions (56 M parameters, 2048-entry codebooks, window 250) on a 3 s clip: float stages within the
    fp32 gates, quantiser bit-exact on the GPU latent, codes equal to the oracle's up to near-ties; the
    device-resident entry point gives the same codes as the host one."""
    import torch
    cfg = synth.ENC_FULL
    path = os.path.join(tempfile.gettempdir(), "hipenc_test_ENC_FULL_0.safetensors")
    W = enc_weights_for(synth, "ENC_FULL", 0)
    if not os.path.exists(path):
        synth.save_safetensors(path, W, cfg.metadata())
    e = pkg.HipSpeechTokenizer.load(path, device=0)
    o = enc_oracle_for(eo, synth, "ENC_FULL", 0)
    audio = synth.make_audio(24000 * 3, 41)
    ids = stage_ids(cfg)
    for name in ("seanet", "transformer", "latent", "proj_sem", "proj_ac"):
        ref, got = o.stage(audio, ids[name]), e.stage(audio, ids[name])
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-3 * max(1.0, float(np.abs(ref).max())), name
        assert snr_db(ref, got) >= 60.0, (name, snr_db(ref, got))
    codes = e.encode(audio)
    assert np.array_equal(o.quantize(e.stage(audio, ids["latent"])), codes)
    agree = check_codes_against_oracle(o, cfg, audio, codes, ids)
    assert agree >= 0.9, agree
    B, n = 3, audio.size
    d_audio = torch.from_numpy(np.stack([audio, synth.make_audio(n, 42), audio])).cuda()
    d_codes = torch.empty((B, e.n_frames(n), 16), dtype=torch.int64, device="cuda")
    e.encode_device(d_audio.data_ptr(), B, n, n, d_codes.data_ptr())
    e.sync()
    out = d_codes.cpu().numpy()
    assert np.array_equal(out[0], codes) and np.array_equal(out[2], codes)
    assert np.array_equal(out[1], e.encode(synth.make_audio(n, 42)))


def test_load_and_call_errors_are_reported_not_fatal(pkg, synth, tmp_path):
    """Error behaviour at the boundary: a bad weights file makes load return NULL with a message (the binding raises),
    a too-small output buffer is HIPVOC_E_CAP, and the process keeps going."""
    import ctypes as C
    cfg = synth.ENC_ODD
    W = dict(enc_weights_for(synth, "ENC_ODD", 0))
    p = str(tmp_path / "bad.safetensors")
    for victim in ("encoder.encoder.layers.3.conv.weight", "encoder.downsample.conv.weight",
                   "encoder.quantizer.semantic_residual_vector_quantizer.input_proj.weight",
                   "encoder.encoder_transformer.layers.0.mlp.fc2.weight"):
        Wb = {k: v for k, v in W.items() if k != victim}
        synth.save_safetensors(p, Wb, cfg.metadata())
        with pytest.raises(pkg.HipVocError):
            pkg.HipSpeechTokenizer.load(p, device=0)
    Wb = dict(W)
    k = "encoder.encoder.layers.1.block.3.conv.weight"
    Wb[k] = Wb[k][:, :-1]                               # 1x1 conv whose input width no longer matches conv3's output
    synth.save_safetensors(p, Wb, cfg.metadata())
    with pytest.raises(pkg.HipVocError):
        pkg.HipSpeechTokenizer.load(p, device=0)
    with pytest.raises(pkg.HipVocError):
        pkg.HipSpeechTokenizer.load(str(tmp_path / "missing.safetensors"), device=0)
    with pytest.raises(pkg.HipVocError):
        pkg.HipSpeechTokenizer.load(p, device=99)
    e = enc_for(pkg, synth, "ENC_ODD")
    a = synth.make_audio(1920 * 4, 2)
    out = np.empty(3 * e.n_codebooks, dtype=np.int64)    # room for 3 frames, 4 are produced
    n = C.c_int64(0)
    rc = e._L.hipvoc_enc_encode(e._h, a.ctypes.data, a.size, out.ctypes.data, out.size, C.byref(n))
    assert rc == -3 and b"capacity" in e._L.hipvoc_last_error(e._h)      # HIPVOC_E_CAP
    assert e.encode(a).shape == (4, e.n_codebooks)


def test_full_size_batch_properties(pkg, synth):
    """The bench's workload shape (many 10 s clips in one device-resident pass, full-size model), checked through
    size-independent properties: every clip's codes equal encoding that clip alone (host path), a clip's prefix
    encodes to the prefix of its codes (causality), and a second pass is bit-identical (determinism)."""
    import torch
    cfg = synth.ENC_FULL
    path = os.path.join(tempfile.gettempdir(), "hipenc_test_ENC_FULL_0.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, enc_weights_for(synth, "ENC_FULL", 0), cfg.metadata())
    e = pkg.HipSpeechTokenizer.load(path, device=0)
    B, n = 24, 24000 * 10
    base = [synth.make_audio(n, 700 + i) for i in range(4)]
    audio = np.stack([base[i % 4] for i in range(B)])
    d_audio = torch.from_numpy(audio).cuda()
    T = e.n_frames(n)
    assert T == 125
    d_codes = torch.empty((B, T, 16), dtype=torch.int64, device="cuda")
    e.encode_device(d_audio.data_ptr(), B, n, n, d_codes.data_ptr())
    e.sync()
    first = d_codes.cpu().numpy()
    assert first.min() >= 0 and first.max() < cfg.codebook_size
    for i in range(4):
        alone = e.encode(base[i])
        for j in range(i, B, 4):
            assert np.array_equal(first[j], alone), (i, j)
    assert np.array_equal(e.encode(base[1][: 1920 * 40 + 123]), first[1][:40])
    d_codes.zero_()
    e.encode_device(d_audio.data_ptr(), B, n, n, d_codes.data_ptr())
    e.sync()
    assert np.array_equal(d_codes.cpu().numpy(), first)
    # codes are not degenerate: many distinct entries of every codebook are used across the batch
    for q in (0, 1, 15):
        assert np.unique(first[:4, :, q]).size > 50


def test_encode_wav_end_to_end_writes_a_voice_profile(pkg, synth, eo, tmp_path):
    """encode_wav (speech_tokenizer.rs:680-721) and what its two callers do with the result (main.rs:440-472,
    api/openai.rs:322-345): a 16 kHz stereo 16-bit WAV -> codes -> voice JSON -> read back by the voice loader."""
    import wave
    e = enc_for(pkg, synth, "ENC_SMALL")
    rng = np.random.default_rng(3)
    left = (0.25 * synth.make_audio(16000 * 2, 5)).astype(np.float32)
    right = (0.5 * left + 0.01 * rng.standard_normal(left.size)).astype(np.float32)
    pcm = np.stack([left, right], axis=1)
    i16 = np.clip(np.round(pcm * 32767), -32768, 32767).astype(np.int16)
    p = str(tmp_path / "speaker.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000); w.writeframes(i16.tobytes())
    codes = e.encode_wav(p)
    f = i16.astype(np.float32) / np.float32(32768)
    mono = ((f[:, 0] + f[:, 1]) / np.float32(2)).astype(np.float32)
    want = e.encode(eo.resample(mono, 16000, 24000))
    assert codes.shape == want.shape == (25, 16) and np.array_equal(codes, want)        # 2 s -> 48000 samples -> 25 frames
    out = str(tmp_path / "speaker.json")
    pkg.voice.save_voice_json(out, codes, "two seconds of a synthetic voice")
    back, text = pkg.voice.load_voice_file(out)
    assert np.array_equal(back, codes) and text.startswith("two seconds")

"""hipvoc-worker (csrc/worker_main.cpp): the reference's `qwen3-tts worker -r vocoder` as a standalone process,
driven over TCP by the mirror of src/worker_client.rs (qwen3-tts_amd/wire.py), checked against the oracle."""
import os
import struct
import subprocess
import tempfile

import msgpack
import numpy as np
import pytest

from conftest import ROOT, oracle_for, weights_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def worker(synth):
    cfg = synth.SMALL
    path = os.path.join(tempfile.gettempdir(), "hipvoc_test_SMALL_0.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, weights_for(synth, "SMALL", 0), cfg.metadata())
    exe = os.path.join(ROOT, "qwen3-tts_amd", "hipvoc-worker")
    assert os.path.exists(exe), "hipvoc-worker not built (make -C qwen3-tts_amd/csrc)"
    p = subprocess.Popen([exe, "-r", "vocoder", "-m", path, "-b", "127.0.0.1:0", "--unit-frames", "4"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    line = p.stdout.readline()
    assert "listening on" in line and "2 device handle(s)" in line, line + p.stdout.read()      # default: two contexts on the GPU
    port = int(line.split("listening on ")[1].split()[0].rsplit(":", 1)[1])
    yield port
    p.kill()
    p.wait()


def test_worker_ping_init_vocode_matches_oracle(pkg, synth, vo, worker):
    wire = pkg.wire
    c = wire.WorkerClient("127.0.0.1", worker)
    assert c.ping()
    assert c.call(wire.request_init("vocoder", "/nonexistent")) == "Init"
    o = oracle_for(vo, synth, "SMALL", 0)
    for n in (1, 4, 10, 9):
        codes = synth.make_codes(n, 70 + n, synth.SMALL)
        pcm = c.vocode(codes)
        ref = o.synthesize(codes, 4)
        assert pcm.size == ref.size
        ref_pcm = vo.pcm16(ref)
        # the float waveform is within 1e-3 of the oracle's, so PCM is within 33 LSB; in practice 0-1
        assert np.abs(pcm.astype(np.int32) - ref_pcm.astype(np.int32)).max() <= 33
    # same connection keeps serving after an error (mod.rs:246-258): a talker request, then bad codes
    with pytest.raises(wire.WorkerError):
        c.call(msgpack.packb({"TalkerStep": [None, 0.9, 1.05, 0.0]}))
    bad = synth.make_codes(4, 1, synth.SMALL)
    bad[2, 3] = synth.SMALL.codebook_size
    with pytest.raises(wire.WorkerError):
        c.vocode(bad)
    with pytest.raises(wire.WorkerError):          # fewer code bytes than n_tokens*16
        c.call(wire.request_vocode(synth.make_codes(2, 1, synth.SMALL), n_tokens=3))
    assert c.ping()
    c.close()


def test_worker_batch_equals_single_requests_and_two_clients(pkg, synth, worker):
    wire = pkg.wire
    a = wire.WorkerClient("127.0.0.1", worker)
    b = wire.WorkerClient("127.0.0.1", worker)
    seqs = [synth.make_codes(n, 90 + n, synth.SMALL) for n in (3, 8, 1, 5)]
    single = [(a if i % 2 else b).vocode(s) for i, s in enumerate(seqs)]
    batch = a.vocode_batch(seqs)
    for x, y in zip(single, batch):
        assert np.array_equal(x, y)
    assert a.vocode_batch([]) == []
    a.close()
    b.close()


def test_worker_rejects_oversize_frame_then_closes(pkg, worker):
    wire = pkg.wire
    c = wire.WorkerClient("127.0.0.1", worker)
    c.sock.sendall(struct.pack(">I", wire.MAX_MESSAGE + 1))
    status, data, err = c.read_response()
    assert status == "error" and "too large" in err
    c.close()
    d = wire.WorkerClient("127.0.0.1", worker)     # the worker itself keeps running
    assert d.ping()
    d.close()


def test_worker_batches_across_connections_on_two_handles(pkg, synth, worker):
    """requests of concurrent connections are queued per device handle and decoded as one batched
    pass; results equal the one-at-a-time results bit for bit (`--devices 0,0` = two handles on this box's GPU)."""
    import signal
    import threading
    wire = pkg.wire
    cfg = synth.SMALL
    path = os.path.join(tempfile.gettempdir(), "hipvoc_test_SMALL_0.safetensors")
    exe = os.path.join(ROOT, "qwen3-tts_amd", "hipvoc-worker")
    p = subprocess.Popen([exe, "-m", path, "-b", "127.0.0.1:0", "--unit-frames", "4", "--devices", "0,0", "--contexts", "1"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        line = p.stdout.readline()
        assert "2 device handle(s)" in line, line
        port = int(line.split("listening on ")[1].split()[0].rsplit(":", 1)[1])
        n_clients, n_req = 8, 6
        seqs = [[synth.make_codes(1 + (3 * c + r) % 11, 500 + 10 * c + r, cfg) for r in range(n_req)] for c in range(n_clients)]
        ref_client = wire.WorkerClient("127.0.0.1", worker)          # the single-handle worker of this module
        expect = [[ref_client.vocode(s) for s in row] for row in seqs]
        ref_client.close()
        got = [[None] * n_req for _ in range(n_clients)]
        errs = []
        gate = threading.Barrier(n_clients)

        def run(c):
            try:
                cl = wire.WorkerClient("127.0.0.1", port)
                for r in range(n_req):
                    gate.wait(timeout=60)
                    got[c][r] = cl.vocode(seqs[c][r])
                cl.close()
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))

        ths = [threading.Thread(target=run, args=(c,)) for c in range(n_clients)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=120)
        assert not errs, errs
        for c in range(n_clients):
            for r in range(n_req):
                assert np.array_equal(got[c][r], expect[c][r]), (c, r)
        p.send_signal(signal.SIGTERM)
        out = p.communicate(timeout=30)[0]
        stats = [ln for ln in out.splitlines() if "stats:" in ln]
        assert len(stats) == 2, out
        jobs = sum(int(ln.split("jobs=")[1].split()[0]) for ln in stats)
        batches = sum(int(ln.split("batches=")[1].split()[0]) for ln in stats)
        assert jobs == n_clients * n_req
        assert batches < jobs, f"no request was ever batched with another: {stats}"
        assert all(int(ln.split("jobs=")[1].split()[0]) > 0 for ln in stats), f"one handle stayed idle: {stats}"
    finally:
        if p.poll() is None:
            p.kill()
            p.wait()


def test_worker_decodes_a_voice_profile_file(pkg, synth, vo, worker, tmp_path):
    """A voice JSON ([N,16] codec tokens + ref_text, src/voice_loader.rs:10-17) is the one realistic code source a
    deployment has on disk: load it with the mirror loader, send it as one Vocode request, compare with the oracle."""
    cfg = synth.SMALL
    # reference-speaker-like statistics: codes dwell (repeat across frames) instead of being i.i.d. uniform
    rng = np.random.default_rng(7)
    base = synth.make_codes(23, 3, cfg)
    hold = rng.random(base.shape) < 0.6
    for t in range(1, base.shape[0]):
        base[t] = np.where(hold[t], base[t - 1], base[t])
    path = str(tmp_path / "speaker.json")
    pkg.voice.save_voice_json(path, base, "reference sentence")
    codes, text = pkg.voice.load_voice_file(path)
    assert text == "reference sentence" and np.array_equal(codes, base)
    c = pkg.wire.WorkerClient("127.0.0.1", worker)
    pcm = c.vocode(codes)
    ref = oracle_for(vo, synth, "SMALL", 0).synthesize(codes, 4)
    assert pcm.size == ref.size
    assert np.abs(pcm.astype(np.int32) - vo.pcm16(ref).astype(np.int32)).max() <= 33


def test_worker_serves_the_orchestrators_pipelined_requests_in_order(pkg, synth, worker):
    """src/pipeline.rs:586-673 keeps several Vocode requests in flight on ONE connection and reads the responses in
    submission order.  The worker decodes them concurrently (they share a batched pass) and must answer in order, with
    exactly the bytes of one-at-a-time calls; an error in the middle stays in its slot."""
    wire = pkg.wire
    cfg = synth.SMALL
    codes = synth.make_codes(23, 321, cfg)                  # chunk 4 -> 5 full requests + a 3-token tail, all in flight
    c = wire.WorkerClient("127.0.0.1", worker)
    got = c.vocode_streamed(codes, chunk_size=4)
    c2 = wire.WorkerClient("127.0.0.1", worker)
    want = np.concatenate([c2.vocode(codes[s: s + 4]) for s in range(0, 23, 4)])
    assert got.size == 5 * cfg.out_len(4) + 3 * 1920 and np.array_equal(got, want)      # full units lose the 555-sample trim
    # 12 requests at once (more than the worker keeps in flight per connection), the 6th is invalid
    reqs = [synth.make_codes(2 + i % 3, 400 + i, cfg) for i in range(12)]
    bad = reqs[5].copy()
    bad[0, 0] = cfg.codebook_size + 7
    for i, r in enumerate(reqs):
        c.send_raw(wire.request_vocode(bad if i == 5 else r))
    for i, r in enumerate(reqs):
        status, data, err = c.read_response()
        if i == 5:
            assert status == "error" and "code" in err
            continue
        assert status == "ok"
        pcm = np.frombuffer(bytes(data["Vocode"][0]), dtype="<i2")
        assert np.array_equal(pcm, c2.vocode(r)), i
    assert c.ping()


def _start(args):
    exe = os.path.join(ROOT, "qwen3-tts_amd", "hipvoc-worker")
    p = subprocess.Popen([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    line = p.stdout.readline()
    while line and "listening on" not in line:          # (a models directory prints the discovered file first)
        line = p.stdout.readline()
    assert "listening on" in line, line + p.stdout.read()
    return p, int(line.split("listening on ")[1].split()[0].rsplit(":", 1)[1])


def test_worker_fans_a_batch_out_over_handles_and_finds_its_model_in_a_directory(pkg, synth, worker, tmp_path):
    """SURVEY 8f-2 "fans out over devices": the items of ONE VocodeBatch are spread over the device queues (`--devices 0,0`
    = two handles on this box's GPU) and come back in request order, bit-equal to single requests.  The worker is given a
    models DIRECTORY, as the reference's `worker -m <models>` is, and applies find_vocoder_model's candidate list to it
    (mod.rs:792-813, here <dir>/vocoder/vocoder.safetensors)."""
    import shutil
    import signal
    wire = pkg.wire
    cfg = synth.SMALL
    models = tmp_path / "models"
    (models / "vocoder").mkdir(parents=True)
    shutil.copy(os.path.join(tempfile.gettempdir(), "hipvoc_test_SMALL_0.safetensors"), models / "vocoder" / "vocoder.safetensors")
    p, port = _start(["-r", "vocoder", "-m", str(models), "-b", "127.0.0.1:0", "--unit-frames", "4", "--devices", "0,0", "--contexts", "1"])
    try:
        seqs = [synth.make_codes(n, 800 + n, cfg) for n in (5, 1, 9, 4, 13, 2, 7, 8, 3)]
        ref = wire.WorkerClient("127.0.0.1", worker)
        want = [ref.vocode(s) for s in seqs]
        ref.close()
        c = wire.WorkerClient("127.0.0.1", port)
        got = c.vocode_batch(seqs)
        assert len(got) == 9 and all(np.array_equal(a, b) for a, b in zip(got, want))
        c.close()
        p.send_signal(signal.SIGTERM)
        out = p.communicate(timeout=30)[0]
        stats = [ln for ln in out.splitlines() if "stats:" in ln]
        assert len(stats) == 2 and all(int(ln.split("jobs=")[1].split()[0]) > 0 for ln in stats), out
        assert sum(int(ln.split("jobs=")[1].split()[0]) for ln in stats) == 9
    finally:
        if p.poll() is None:
            p.kill()
            p.wait()


def test_worker_survives_hostile_requests(pkg, synth, worker):
    """allocation sizes must not be attacker-controlled and nothing a request carries may end the process.
    (1) a VocodeBatch whose array32 header claims 1e8 items in a 40-byte frame; (2) a frame that does not deserialize:
    error response, then the connection is closed as read_request's Err arm does (mod.rs:642-651); (3) more frames than
    --max-frames: error response, connection stays; the worker serves on after all of them."""
    wire = pkg.wire
    p, port = _start(["-m", os.path.join(tempfile.gettempdir(), "hipvoc_test_SMALL_0.safetensors"), "-b", "127.0.0.1:0",
                      "--unit-frames", "4", "--max-frames", "50", "--max-items", "8"])
    try:
        c = wire.WorkerClient("127.0.0.1", port)
        hostile = b"\x81" + msgpack.packb("VocodeBatch") + b"\x92" + b"\xdd" + struct.pack(">I", 100_000_000) + b"\x90" * 8
        c.send_raw(hostile)
        status, _, err = c.read_response()
        assert status == "error" and "Deserialize" in err
        with pytest.raises((ConnectionError, OSError)):        # closed after the malformed frame
            c.ping()
            c.ping()
        c.close()
        c = wire.WorkerClient("127.0.0.1", port)
        with pytest.raises(wire.WorkerError, match="max-frames"):
            c.vocode(synth.make_codes(51, 1, synth.SMALL))
        with pytest.raises(wire.WorkerError, match="max-frames"):
            c.vocode_batch([synth.make_codes(30, 1, synth.SMALL), synth.make_codes(30, 2, synth.SMALL)])
        with pytest.raises(wire.WorkerError, match="max-items"):
            c.vocode_batch([synth.make_codes(1, i, synth.SMALL) for i in range(9)])
        ok = c.vocode(synth.make_codes(50, 1, synth.SMALL))
        ref = wire.WorkerClient("127.0.0.1", worker)
        assert np.array_equal(ok, ref.vocode(synth.make_codes(50, 1, synth.SMALL)))
        ref.close()
        assert c.ping()
        c.close()
    finally:
        p.kill()
        p.wait()


def test_orchestrator_queue_coalesces_concurrent_utterances_into_vocode_batches(pkg, synth, worker):
    """SURVEY 8f-1, orchestrator side (mirror of integration/pipeline_vocode_batch.rs): three utterances are "generated"
    concurrently, each submitting its 4-token chunks as they appear and its tail last; the queue owns the one vocoder
    connection and sends whatever is waiting as ONE VocodeBatch.  Every utterance's audio, assembled in its own submission
    order, equals the reference's streamed single requests bit for bit."""
    import threading
    wire = pkg.wire
    cfg = synth.SMALL
    utts = [synth.make_codes(n, 600 + n, cfg) for n in (23, 9, 14)]
    ref = wire.WorkerClient("127.0.0.1", worker)
    want = [ref.vocode_streamed(u, chunk_size=4) for u in utts]
    ref.close()
    q = wire.VocoderQueue("127.0.0.1", worker)
    got = [None] * len(utts)
    gate = threading.Barrier(len(utts))

    def synthesize(i):
        gate.wait(timeout=30)
        u = utts[i]
        futs = [q.submit(u[s: s + 4]) for s in range(0, u.shape[0], 4)]        # full chunks, then the tail
        got[i] = np.concatenate([f.result(timeout=60) for f in futs])           # "collect responses in submission order"

    ths = [threading.Thread(target=synthesize, args=(i,)) for i in range(len(utts))]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    q.close()
    for g, w in zip(got, want):
        assert g is not None and np.array_equal(g, w)
    assert q.batches_sent >= 1, "nothing was ever coalesced"
    assert q.batches_sent + q.singles_sent < sum((u.shape[0] + 3) // 4 for u in utts)

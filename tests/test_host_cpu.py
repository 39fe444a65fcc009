"""CPU-side tests: the C ABI library loads and exports every declared symbol, host logic, sharding
(world_size-2 gloo), safetensors round trip.  No GPU compute here."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("header,exports", [("hipvoc.h", "hipvoc.EXPORTS"), ("hipvoc_wire.h", "wire.WIRE_EXPORTS"),
                                            ("hipvoc_encoder.h", "encoder.ENC_EXPORTS")])
def test_library_exports_every_declared_symbol(pkg, header, exports):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(hipvoc_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = pkg.load_library()
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in include/{header} but not exported"
    mod, name = exports.split(".")
    assert declared == set(getattr(getattr(pkg, mod), name))


def test_no_cpu_fallback(pkg, tmp_path):
    """Without a HIP device the product path must fail loudly, not compute on the CPU."""
    if pkg.load_library().hipvoc_device_count() > 0:
        pytest.skip("a GPU is present")
    W = pkg.synth.make_weights(pkg.synth.ODD, 0)
    path = str(tmp_path / "w.safetensors")
    pkg.synth.save_safetensors(path, W, pkg.synth.ODD.metadata())
    with pytest.raises(pkg.HipVocError):
        pkg.HipVocoder.load(path)
    EW = pkg.synth.make_encoder_weights(pkg.synth.ENC_ODD, 0)
    pkg.synth.save_safetensors(path, EW, pkg.synth.ENC_ODD.metadata())
    with pytest.raises(pkg.HipVocError):
        pkg.HipSpeechTokenizer.load(path)


def test_product_never_touches_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "qwen3-tts_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "voc_oracle" not in text and "enc_oracle" not in text and "oracle/" not in text, os.path.join(dirpath, f)


def test_safetensors_writer_matches_the_reference_package(pkg, tmp_path):
    from safetensors.numpy import load_file
    W = pkg.synth.make_weights(pkg.synth.ODD, 3)
    path = str(tmp_path / "w.safetensors")
    pkg.synth.save_safetensors(path, W, pkg.synth.ODD.metadata())
    back = load_file(path)
    assert set(back) == set(W)
    for k in W:
        assert np.array_equal(back[k], W[k])


def test_unit_spans_and_shard_ranges(pkg):
    sh = pkg.sharding
    assert sh.unit_spans(125, 64) == [(0, 64), (64, 125)]
    assert sh.unit_spans(128, 64) == [(0, 64), (64, 128)]
    assert sh.unit_spans(0, 64) == []
    assert len(sh.unit_spans(7500, 64)) == 118 and sh.unit_spans(7500, 64)[-1] == (7488, 7500)
    for n in (0, 1, 7, 118, 512):
        for world in (1, 2, 4, 8):
            got = [sh.shard_range(n, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(got[i][1] == got[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in got]
            assert max(sizes) - min(sizes) <= 1


def test_model_config_lengths(pkg):
    assert pkg.synth.FULL.out_len(64) == 122325
    assert pkg.synth.FULL.out_len(1) == 1920 - 555


WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch.distributed as dist
import _pkgload
pkg = _pkgload.load_pkg(); vo = _pkgload.load_oracle(); synth = pkg.synth
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
cfg = synth.SMALL
o = vo.Oracle(synth.make_weights(cfg, 0), {{k: float(v) for k, v in cfg.metadata().items()}})
codes = synth.make_codes(11, 77, cfg)          # 3 units of 4 frames: 4 + 4 + 3
unit = 4
local = pkg.sharding.decode_stream_shard(lambda c: o.synthesize(c, unit), codes, unit, rank, world)
full = pkg.sharding.gather_stream(local, rank, world, dist)
if rank == 0:
    ref = o.synthesize(codes, unit)
    assert full.shape == ref.shape and np.array_equal(full, ref), "sharded stream differs from single decode"
    print("OK", full.size)
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    """A port the kernel just handed out (bind to 0), as bench.py's launcher does: two suites on one box do not collide."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_longform_sharding_world2_gloo(tmp_path):
    """The N>1 path (contiguous unit ranges per rank + gather on rank 0) on CPU with gloo; the decode
    function is the oracle here, the partition/stitch logic is the product's."""
    script = tmp_path / "w.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK 20010" in r.stdout        # 2 full units of out_len(4)=7125 + a 3-frame tail of 3*1920


def test_longform_expected_sample_count(pkg):
    cfg = pkg.synth.SMALL
    assert 2 * cfg.out_len(4) + 3 * 1920 == 20010


ENC_WORKER = """
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch.distributed as dist
import _pkgload
pkg = _pkgload.load_pkg(); eo = _pkgload.load_enc_oracle(); synth = pkg.synth
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
cfg = synth.ENC_SMALL
o = eo.EncOracle(synth.make_encoder_weights(cfg, 0), {{k: float(v) for k, v in cfg.metadata().items()}})
clips = [synth.make_audio(n, 50 + i) for i, n in enumerate((1920 * 3, 1920 * 5 + 7, 1920 * 2, 1920 * 4 + 1000, 1920))]
local = pkg.sharding.encode_clips_shard(lambda cs: [o.encode(c) for c in cs], clips, rank, world)
full = pkg.sharding.gather_codes(local, rank, world, dist)
if rank == 0:
    assert len(full) == len(clips)
    for c, got in zip(clips, full):
        assert np.array_equal(got, o.encode(c)), "sharded batch differs from encoding each clip alone"
    print("OK", sum(len(c) for c in full))
dist.barrier()
dist.destroy_process_group()
"""


def test_encoder_clip_sharding_world2_gloo(tmp_path):
    """The encoder's N>1 path (contiguous clip ranges per rank, ragged gather on rank 0) on CPU with gloo; the encode
    function is the oracle here, the partition / gather logic is the product's."""
    script = tmp_path / "e.py"
    script.write_text(ENC_WORKER.format(root=ROOT))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK 15" in r.stdout          # 3 + 5 + 2 + 4 + 1 frames


def test_safetensors_reader_survives_fuzzing_under_asan_ubsan(pkg, tmp_path):
    """tests/native/fuzz_safetensors.cpp: truncations and 20k header / length mutations of a valid file through the
    product's reader, built with -fsanitize=address,undefined (host code only): load or fail with a message, never an
    out-of-bounds access -- weight files come from outside."""
    import shutil
    if not shutil.which("g++"):
        pytest.skip("no g++")
    rng = np.random.default_rng(0)
    good = str(tmp_path / "good.safetensors")
    pkg.synth.save_safetensors(good, {"a.weight": rng.standard_normal((5, 3, 2)).astype(np.float32),
                                      "b.bias": rng.standard_normal((7,)).astype(np.float32),
                                      "c": rng.standard_normal((2, 2)).astype(np.float32)}, {"k": "v"}, dtype="F16")
    exe = str(tmp_path / "fuzz_st")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                        os.path.join(ROOT, "tests", "native", "fuzz_safetensors.cpp"),
                        os.path.join(ROOT, "qwen3-tts_amd", "csrc", "safetensors.cpp")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, good, str(tmp_path / "scratch.safetensors")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fuzz_safetensors OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_chunk_loop_and_exception_barrier_under_asan_ubsan(tmp_path):
    """tests/native/fuzz_chunking.cpp: the host logic the decode entry points run before any device work (chunking.hpp:
    unit walk, extents, code validation on unaligned bytes) against a literal restatement of vocoder.rs:65-82 on 20k
    random batches, INT_MAX-frame extremes without overflow or allocation, and the C ABI's exception barrier
    (abi_guard.hpp: injected bad_alloc / length_error / runtime_error / non-std throws come back as HIPVOC_E_NOMEM /
    HIPVOC_E_INTERNAL), built with -fsanitize=address,undefined."""
    import shutil
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "fuzz_chunking")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                        os.path.join(ROOT, "tests", "native", "fuzz_chunking.cpp")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fuzz_chunking OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_every_extern_c_body_that_allocates_sits_behind_the_barrier():
    """include/hipvoc.h promises that nothing is thrown across the boundary: every extern "C" function of hipvoc.cpp /
    encoder.inc.cpp whose body builds a std::vector / std::string / plan goes through guarded() or a try block."""
    for name in ("hipvoc.cpp", "encoder.inc.cpp", "wire.cpp", "discovery.cpp"):
        src = open(os.path.join(ROOT, "qwen3-tts_amd", "csrc", name)).read()
        ext = src[src.index('extern "C"'):]
        # split into top-level function bodies: "\n}\n" ends one
        for body in re.split(r"\n}\n", ext):
            m = re.search(r"\n(?:static )?(?:[\w:\*]+ )+\**(hipvoc_\w+)\(", "\n" + body)
            if not m:
                continue
            allocates = re.search(r"std::vector|std::string|get_plan|enc_run|fmt\(|new ", body)
            if allocates:
                assert "guarded(" in body or "try {" in body or "_impl(" in body, f"{name}: {m.group(1)} allocates outside the exception barrier"


def test_find_vocoder_model_candidate_order(pkg, tmp_path):
    """find_vocoder_model (src/worker/mod.rs:792-813) for this backend: ordered candidates, the role sub-directory that
    download::ensure_models returns, and the reference's error wording when nothing is there."""
    hv = pkg.hipvoc
    d = tmp_path / "models"
    d.mkdir()
    with pytest.raises(pkg.HipVocError, match="No vocoder model found in"):
        hv.find_vocoder_model(str(d))
    (d / "vocoder.onnx").write_bytes(b"x")
    with pytest.raises(pkg.HipVocError, match="ONNX/RKNN"):
        hv.find_vocoder_model(str(d))
    (d / "speech_tokenizer").mkdir()
    (d / "speech_tokenizer" / "model.safetensors").write_bytes(b"x")
    assert hv.find_vocoder_model(str(d)) == str(d / "speech_tokenizer" / "model.safetensors")
    (d / "vocoder").mkdir()
    (d / "vocoder" / "vocoder_64.safetensors").write_bytes(b"x")
    assert hv.find_vocoder_model(str(d)) == str(d / "speech_tokenizer" / "model.safetensors")      # the directory itself first
    assert hv.find_vocoder_model(str(d / "vocoder")) == str(d / "vocoder" / "vocoder_64.safetensors")
    (d / "vocoder.safetensors").write_bytes(b"x")
    assert hv.find_vocoder_model(str(d)) == str(d / "vocoder.safetensors")
    (d / "vocoder.safetensors").unlink()
    (d / "vocoder.safetensors").mkdir()                     # a directory of that name is not a model file
    assert hv.find_vocoder_model(str(d)) == str(d / "speech_tokenizer" / "model.safetensors")


def test_load_error_text_is_per_thread(pkg):
    """hipvoc_last_error(NULL): the calling thread's own failed load, so concurrent loads (hipvoc-worker: contexts x
    devices) cannot clobber or race on each other's message."""
    import ctypes as C
    import threading
    L = pkg.load_library()
    seen = {}

    def run(i):
        for _ in range(200):
            h = L.hipvoc_load(None if i == 0 else f"/nonexistent/{i}.safetensors".encode(), 0, 7 if i == 1 else 0, 64)
            assert not h
            seen.setdefault(i, set()).add(L.hipvoc_last_error(None).decode())

    ths = [threading.Thread(target=run, args=(i,)) for i in range(3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert seen[0] == {"weights_path is NULL"}
    assert len(seen[1]) == 1 and "dtype" in next(iter(seen[1]))
    assert len(seen[2]) == 1


# ------------------------------------------------------------------------------------ the static-link recipe
STATIC_HOST = os.path.join(ROOT, "qwen3-tts_amd", "static_link_host")


def test_static_archive_links_a_plain_c_host_as_build_rs_says(tmp_path):
    """integration/build_rs_fragment.rs links `static=hipvoc`, `dylib=amdhip64`, `dylib=stdc++`.  csrc/Makefile builds
    libhipvoc.a and links tests/native/static_link_host.c (plain C) with exactly those; here: the link line still says so,
    the binary does not depend on libhipvoc.so, and it runs -- device count, find_vocoder_model on a models directory,
    and a failing hipvoc_load whose error text comes back through hipvoc_last_error(NULL)."""
    mk = open(os.path.join(ROOT, "qwen3-tts_amd", "csrc", "Makefile")).read()
    link = [l for l in mk.splitlines() if "static_link_host.o" in l and " -o $@" in l]
    assert link and "-Wl,-Bstatic -lhipvoc -Wl,-Bdynamic" in link[0] and "-lamdhip64" in link[0] and "-lstdc++" in link[0]
    assert link[0].lstrip().startswith("gcc "), "the host must be linked by a plain C driver, as rustc's cc would"
    frag = open(os.path.join(ROOT, "integration", "build_rs_fragment.rs")).read()
    for want in ("static=hipvoc", "dylib=amdhip64", "dylib=stdc++", "../libhipvoc.a"):
        assert want in frag
    assert os.path.exists(STATIC_HOST), "run `make -C qwen3-tts_amd/csrc` (or __graft_entry__.build())"
    assert os.path.exists(os.path.join(ROOT, "qwen3-tts_amd", "libhipvoc.a"))
    ldd = subprocess.run(["ldd", STATIC_HOST], capture_output=True, text=True).stdout
    assert "libhipvoc" not in ldd and "libamdhip64" in ldd and "libstdc++" in ldd, ldd
    models = tmp_path / "models"
    (models / "vocoder").mkdir(parents=True)
    (models / "vocoder" / "vocoder.safetensors").write_bytes(b"")
    r = subprocess.run([STATIC_HOST, "probe", str(models)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    out = dict(l.split(" ", 1) for l in r.stdout.strip().splitlines())
    assert int(out["device_count"]) >= 0
    assert out["find_vocoder_model"].startswith("rc=0 ") and out["find_vocoder_model"].endswith("vocoder/vocoder.safetensors")
    assert len(out["load_error"]) > 8 and out["load_error"] != "(empty)"

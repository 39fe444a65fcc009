#!/usr/bin/env python3
"""End-to-end throughput of hipvoc-worker over TCP: P client processes, each sending R Vocode requests of one 10 s
utterance (125 frames) over its own connection, responses fully read (not decoded: the clients must not be the limit)."""
import multiprocessing as mp
import os, socket, struct, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def client(port, R, seed, inflight, q):
    import _pkgload
    pkg = _pkgload.load_pkg()
    codes = pkg.synth.make_codes(125, seed, pkg.synth.FULL)
    payload = pkg.wire.request_vocode(codes)
    frame = struct.pack(">I", len(payload)) + payload
    s = socket.create_connection(("127.0.0.1", port))
    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)

    def read_resp():
        hdr = b""
        while len(hdr) < 4:
            hdr += s.recv(4 - len(hdr))
        n = struct.unpack(">I", hdr)[0]
        got = 0
        while got < n:
            got += len(s.recv(min(1 << 20, n - got)))
        return n
    s.sendall(frame); read_resp()          # warm-up
    q.put("ready")
    t0 = time.perf_counter()
    sent = done = 0
    while done < R:
        while sent < R and sent - done < inflight:
            s.sendall(frame); sent += 1
        read_resp(); done += 1
    q.put(time.perf_counter() - t0)


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    inflight = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    dtype = sys.argv[4] if len(sys.argv) > 4 else "f32"
    devices = sys.argv[5] if len(sys.argv) > 5 else "0"
    import _pkgload
    pkg = _pkgload.load_pkg(); synth = pkg.synth
    path = os.path.join(tempfile.gettempdir(), "hipvoc_prof_full.safetensors")
    if not os.path.exists(path):
        synth.save_safetensors(path, synth.make_weights(synth.FULL, 0), synth.FULL.metadata())
    exe = os.path.join(ROOT, "qwen3-tts_amd", "hipvoc-worker")
    w = subprocess.Popen([exe, "-r", "vocoder", "-m", path, "-b", "127.0.0.1:0", "--dtype", dtype, "--devices", devices], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True)
    line = w.stdout.readline()
    port = int(line.split("listening on ")[1].split()[0].rsplit(":", 1)[1])
    q = mp.Queue()
    procs = [mp.Process(target=client, args=(port, R, 100 + i, inflight, q)) for i in range(P)]
    for p in procs:
        p.start()
    for _ in procs:
        assert q.get() == "ready"
    t0 = time.perf_counter()
    times = [q.get() for _ in procs]
    wall = time.perf_counter() - t0
    for p in procs:
        p.join()
    w.terminate()
    out = w.stdout.read()
    print(f"{dtype} devices={devices}: {P} clients x {R} requests x 10 s, {inflight} in flight each: {P*R*10/wall:.0f} audio-s/s through the worker "
          f"(slowest client {max(times):.2f} s)")
    if P == 1 and inflight == 1:
        print(f"  one client, one request at a time: {1e3 * max(times) / R:.2f} ms per 10 s utterance, request sent -> response (PCM16) fully read")
    print(out.strip()[-400:])


if __name__ == "__main__":
    main()

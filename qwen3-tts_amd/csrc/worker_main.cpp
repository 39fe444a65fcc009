// hipvoc-worker -- a drop-in for `qwen3-tts worker -r vocoder -b HOST:PORT` (reference: src/worker/mod.rs:552-698).
//
// Speaks the reference's TCP protocol (u32 BE length + rmp-serde payload, include/hipvoc_wire.h) for the Vocoder
// role and nothing else: Ping, Init, Vocode (+ the VocodeBatch extension); requests of other roles get the
// reference's error response and the connection stays up.  The orchestrator (pipeline.rs, the HTTP API) needs
// no change: point `--vocoder-ip` at this process.
//
// Serving model.  The reference serialises every request behind one Arc<Mutex<Worker>>
// (mod.rs:567-577,628-631).  Here each GPU (`--devices 0,1,...`) has one libhipvoc handle, a job queue and one
// service thread; a client thread enqueues its request on the least-loaded GPU and waits.  A service thread
// takes EVERYTHING that is queued when it becomes free (up to --max-batch) and decodes it as one batched pass
// (hipvoc_decode_batch), so concurrent connections are batched across requests without any added latency
// when the worker is idle.  Results are bit-identical to one request at a time (units are independent).
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <signal.h>
#include <sys/socket.h>
#include <unistd.h>

#include <poll.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hipvoc.h"
#include "../../include/hipvoc_wire.h"

static const size_t MAX_MSG = 100000000;   // mod.rs:666
static inline int16_t pcm16(float s);

// One decode job = one utterance ([n,16] i64 codes).  A VocodeBatch request enqueues one job per item.
struct Job {
    const int64_t *codes = nullptr;
    int n_frames = 0;
    std::vector<int16_t> pcm;
    int rc = 0;
    std::string err;
    bool done = false;
};

struct Gpu {
    void *voc = nullptr;
    int device = 0;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<Job *> q;
    std::atomic<int> load{0};
    std::thread th;
    // stats
    long batches = 0, jobs = 0, max_batch = 0;
};
static std::vector<std::unique_ptr<Gpu>> g_gpus;
static std::atomic<bool> g_stop{false};
static int g_max_batch = 256;

static void gpu_service(Gpu *g) {
    std::vector<Job *> take;
    for (;;) {
        take.clear();
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv_work.wait(lk, [&] { return !g->q.empty() || g_stop.load(); });
            if (g->q.empty()) return;
            while (!g->q.empty() && (int)take.size() < g_max_batch) { take.push_back(g->q.front()); g->q.pop_front(); }
            g->batches++; g->jobs += (long)take.size();
            if ((long)take.size() > g->max_batch) g->max_batch = (long)take.size();
        }
        size_t B = take.size();
        // one batched pass for everything that was waiting
        std::vector<int64_t> all;
        std::vector<int32_t> nf(B);
        std::vector<size_t> off(B + 1, 0), n_out(B, 0);
        for (size_t j = 0; j < B; j++) {
            all.insert(all.end(), take[j]->codes, take[j]->codes + (size_t)take[j]->n_frames * 16);
            nf[j] = take[j]->n_frames;
            off[j + 1] = off[j] + (size_t)take[j]->n_frames * HIPVOC_SAMPLES_PER_TOKEN;
        }
        std::vector<float> wav(off[B] + 1);
        if (all.empty()) all.push_back(0);
        int rc = hipvoc_decode_batch(g->voc, all.data(), nf.data(), (int)B, wav.data(), off.data(), n_out.data());
        std::string err = rc ? hipvoc_last_error(g->voc) : "";
        if (rc && B > 1) {
            // a bad request must not fail its batch neighbours: redo them one by one
            for (size_t j = 0; j < B; j++) {
                size_t n1 = 0, o1 = 0;
                int32_t f1 = take[j]->n_frames;
                std::vector<float> w1((size_t)f1 * HIPVOC_SAMPLES_PER_TOKEN + 1);
                int64_t zero = 0;
                const int64_t *cp = f1 ? take[j]->codes : &zero;
                take[j]->rc = hipvoc_decode_batch(g->voc, cp, &f1, 1, w1.data(), &o1, &n1);
                if (take[j]->rc) take[j]->err = hipvoc_last_error(g->voc);
                else { take[j]->pcm.resize(n1); for (size_t k = 0; k < n1; k++) take[j]->pcm[k] = pcm16(w1[k]); }
            }
        } else {
            for (size_t j = 0; j < B; j++) {
                take[j]->rc = rc;
                take[j]->err = err;
                if (!rc) {
                    take[j]->pcm.resize(n_out[j]);
                    const float *w = wav.data() + off[j];
                    for (size_t k = 0; k < n_out[j]; k++) take[j]->pcm[k] = pcm16(w[k]);
                }
            }
        }
        {
            std::lock_guard<std::mutex> lk(g->mu);
            for (Job *j : take) j->done = true;
        }
        g->load -= (int)B;
        g->cv_done.notify_all();
    }
}

// enqueue on the least-loaded GPU and wait for all of them
static void run_jobs(std::vector<Job> &jobs) {
    if (jobs.empty()) return;
    static std::atomic<unsigned> rr{0};
    size_t best = rr++ % g_gpus.size();
    for (size_t i = 0; i < g_gpus.size(); i++)
        if (g_gpus[i]->load.load() < g_gpus[best]->load.load()) best = i;
    Gpu *g = g_gpus[best].get();
    g->load += (int)jobs.size();
    {
        std::lock_guard<std::mutex> lk(g->mu);
        for (Job &j : jobs) g->q.push_back(&j);
    }
    g->cv_work.notify_one();
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv_done.wait(lk, [&] { for (Job &j : jobs) if (!j.done) return false; return true; });
}

static bool read_all(int fd, uint8_t *p, size_t n) {
    while (n) {
        ssize_t r = recv(fd, p, n, 0);
        if (r <= 0) return false;
        p += r;
        n -= (size_t)r;
    }
    return true;
}
static bool write_all(int fd, const uint8_t *p, size_t n) {
    while (n) {
        ssize_t r = send(fd, p, n, MSG_NOSIGNAL);
        if (r <= 0) return false;
        p += r;
        n -= (size_t)r;
    }
    return true;
}
static bool send_frame(int fd, const std::vector<uint8_t> &payload) {
    uint32_t l = htonl((uint32_t)payload.size());
    return write_all(fd, (const uint8_t *)&l, 4) && write_all(fd, payload.data(), payload.size());
}
static std::vector<uint8_t> simple(int ok, const char *variant, const char *err) {
    std::vector<uint8_t> b(hipvoc_wire_encode_simple(ok, variant, err, nullptr, 0));
    hipvoc_wire_encode_simple(ok, variant, err, b.data(), b.size());
    return b;
}
// f32 -> i16 exactly as src/worker/mod.rs:521-524 (Rust `as i16`: truncation toward zero, saturating, NaN -> 0)
static inline int16_t pcm16(float s) {
    float v = s * 32767.0f;
    if (v != v) return 0;
    if (v < -32768.0f) v = -32768.0f;
    if (v > 32767.0f) v = 32767.0f;
    return (int16_t)v;
}

static std::vector<uint8_t> handle(const std::vector<uint8_t> &msg) {
    size_t clen = 0;
    uint64_t nt = 0;
    int kind = hipvoc_wire_parse_request(msg.data(), msg.size(), nullptr, 0, &clen, &nt);
    if (kind == HIPVOC_WIRE_PING) return simple(1, "Ping", nullptr);
    if (kind == HIPVOC_WIRE_INIT) return simple(1, "Init", nullptr);
    if (kind == HIPVOC_WIRE_OTHER) return simple(0, "Init", "This worker is a vocoder (hipvoc): request belongs to another role");
    if (kind == HIPVOC_WIRE_VOCODE) {
        std::vector<uint8_t> codes(clen ? clen : 1);
        hipvoc_wire_parse_request(msg.data(), msg.size(), codes.data(), codes.size(), &clen, &nt);
        if (nt > (1u << 24)) return simple(0, "Init", "n_tokens out of range");
        if (clen / 8 < nt * 16) return simple(0, "Init", "codes shorter than n_tokens*16");
        std::vector<int64_t> c64((size_t)nt * 16 + 1);
        memcpy(c64.data(), codes.data(), (size_t)nt * 16 * 8);          // i64 little-endian == host order (mod.rs:513-516)
        std::vector<Job> jobs(1);
        jobs[0].codes = c64.data();
        jobs[0].n_frames = (int)nt;
        static const bool trace = getenv("HIPVOC_WORKER_TIMING") != nullptr;     // per-request host timing to stderr
        auto t0 = std::chrono::steady_clock::now();
        run_jobs(jobs);
        auto t1 = std::chrono::steady_clock::now();
        if (jobs[0].rc) return simple(0, "Init", jobs[0].err.c_str());
        std::vector<int16_t> &pcm = jobs[0].pcm;
        int16_t dummy = 0;
        const int16_t *pp = pcm.empty() ? &dummy : pcm.data();
        std::vector<uint8_t> b(hipvoc_wire_encode_vocode(pp, pcm.size(), nullptr, 0));
        hipvoc_wire_encode_vocode(pp, pcm.size(), b.data(), b.size());
        if (trace) {
            auto t2 = std::chrono::steady_clock::now();
            fprintf(stderr, "vocode %d frames: queue+decode+pcm %.3f ms, encode %.3f ms (%zu bytes)\n", (int)nt,
                    std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(), b.size());
        }
        return b;
    }
    if (kind == HIPVOC_WIRE_VOCODE_BATCH) {
        size_t B = 0;
        if (hipvoc_wire_parse_vocode_batch(msg.data(), msg.size(), nullptr, 0, nullptr, nullptr, 0, &B) != HIPVOC_WIRE_VOCODE_BATCH)
            return simple(0, "Init", "Deserialize request failed: malformed VocodeBatch");
        std::vector<uint8_t> codes(msg.size() + 1);
        std::vector<size_t> off(B + 1);
        std::vector<uint64_t> nts(B + 1);
        if (hipvoc_wire_parse_vocode_batch(msg.data(), msg.size(), codes.data(), codes.size(), off.data(), nts.data(), B, &B) !=
            HIPVOC_WIRE_VOCODE_BATCH)
            return simple(0, "Init", "Deserialize request failed: malformed VocodeBatch");
        // gather [sum n, 16] i64 codes; every item must carry at least n_tokens*16 codes (the reference would
        // panic on the slice; here it is an error response)
        std::vector<int64_t> all;
        std::vector<int32_t> nf(B ? B : 1);
        std::vector<size_t> out_off(B + 1, 0), n_out(B ? B : 1, 0);
        for (size_t j = 0; j < B; j++) {
            if (nts[j] > (1u << 24) || (off[j + 1] - off[j]) / 8 < nts[j] * 16) return simple(0, "Init", "codes shorter than n_tokens*16");
            size_t base = all.size();
            all.resize(base + nts[j] * 16);
            memcpy(all.data() + base, codes.data() + off[j], nts[j] * 16 * 8);
            nf[j] = (int32_t)nts[j];
            out_off[j + 1] = out_off[j] + nts[j] * HIPVOC_SAMPLES_PER_TOKEN;
        }
        if (all.empty()) all.push_back(0);
        std::vector<Job> jobs(B);
        {
            size_t base = 0;
            for (size_t j = 0; j < B; j++) { jobs[j].codes = all.data() + base; jobs[j].n_frames = nf[j]; base += (size_t)nts[j] * 16; }
        }
        run_jobs(jobs);
        std::vector<int16_t> pcm;
        std::vector<size_t> so(B + 1, 0);
        for (size_t j = 0; j < B; j++) {
            if (jobs[j].rc) return simple(0, "Init", jobs[j].err.c_str());
            pcm.insert(pcm.end(), jobs[j].pcm.begin(), jobs[j].pcm.end());
            so[j + 1] = pcm.size();
        }
        if (pcm.empty()) pcm.push_back(0);
        std::vector<uint8_t> b(hipvoc_wire_encode_vocode_batch(pcm.data(), so.data(), B, nullptr, 0));
        hipvoc_wire_encode_vocode_batch(pcm.data(), so.data(), B, b.data(), b.size());
        return b;
    }
    return simple(0, "Init", "Read: Deserialize request failed: malformed message");
}

// One connection.  The reference's orchestrator PIPELINES vocoder requests on its connection (pipeline.rs:586-607 sends a
// Vocode per 64-token chunk while the talker is still generating, 651 queues the tail, 660-672 then reads the responses
// in submission order), so this thread only reads and dispatches frames; up to MAX_INFLIGHT requests of the connection
// are decoded concurrently -- they meet in the GPU queue and leave as one batched pass -- and a writer thread returns
// the responses in order.
static const size_t MAX_INFLIGHT = 8;

static void client(int fd) {
    int one = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::future<std::vector<uint8_t>>> inflight;
    bool eof = false, broken = false;
    std::thread writer([&] {
        for (;;) {
            std::future<std::vector<uint8_t>> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !inflight.empty() || eof; });
                if (inflight.empty()) return;
                f = std::move(inflight.front());
            }
            std::vector<uint8_t> resp = f.get();
            bool ok = send_frame(fd, resp);
            {
                std::lock_guard<std::mutex> lk(mu);
                inflight.pop_front();                       // (popped after the send: the reader's bound counts it)
                if (!ok) broken = true;
            }
            cv.notify_all();
            if (!ok) {                                      // peer gone: drain what is still running, send nothing
                for (;;) {
                    std::future<std::vector<uint8_t>> g;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return !inflight.empty() || eof; });
                        if (inflight.empty()) return;
                        g = std::move(inflight.front());
                        inflight.pop_front();
                    }
                    g.get();
                    cv.notify_all();
                }
            }
        }
    });
    for (;;) {
        uint8_t lb[4];
        if (!read_all(fd, lb, 4)) break;                       // client went away (mod.rs:662-664)
        size_t len = ((size_t)lb[0] << 24) | ((size_t)lb[1] << 16) | ((size_t)lb[2] << 8) | lb[3];
        std::future<std::vector<uint8_t>> fut;
        bool last = false;
        if (len > MAX_MSG) {                                   // mod.rs:666-668, 603-611: error response, then close
            std::promise<std::vector<uint8_t>> pr;
            pr.set_value(simple(0, "Init", "Read: Message too large"));
            fut = pr.get_future();
            last = true;
        } else {
            std::vector<uint8_t> msg(len);
            if (len && !read_all(fd, msg.data(), len)) break;
            fut = std::async(std::launch::async, [m = std::move(msg)] { return handle(m); });
        }
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return inflight.size() < MAX_INFLIGHT || broken; });
            inflight.push_back(std::move(fut));
            if (broken) last = true;
        }
        cv.notify_all();
        if (last) break;
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        eof = true;
    }
    cv.notify_all();
    writer.join();
    close(fd);
}

static void on_term(int) { g_stop.store(true); }

int main(int argc, char **argv) {
    std::string bind_addr = "0.0.0.0:9092", model, devices = "0";
    int dtype = HIPVOC_DTYPE_F32, unit = 64, contexts = 2;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "-b" || a == "--bind") bind_addr = next();
        else if (a == "-m" || a == "--model") model = next();
        else if (a == "--device" || a == "--devices") devices = next();
        else if (a == "--dtype") { std::string d = next(); dtype = d == "bf16" ? HIPVOC_DTYPE_BF16 : d == "bf16x3" ? HIPVOC_DTYPE_BF16X3 : HIPVOC_DTYPE_F32; }
        else if (a == "--unit-frames") unit = atoi(next());
        else if (a == "--max-batch") g_max_batch = atoi(next());
        else if (a == "--contexts") contexts = atoi(next());
        else if (a == "-r" || a == "--role") { std::string r = next(); if (r != "vocoder") { fprintf(stderr, "hipvoc-worker only serves role 'vocoder'\n"); return 2; } }
        else { fprintf(stderr, "usage: hipvoc-worker -m <decoder.safetensors> [-b host:port] [--devices 0,1,..] [--dtype f32|bf16|bf16x3] [--unit-frames 64] [--max-batch 256] [--contexts 2]\n"); return 2; }
    }
    if (model.empty()) { fprintf(stderr, "hipvoc-worker: -m <weights.safetensors> is required\n"); return 2; }
    if (g_max_batch < 1) g_max_batch = 1;
    size_t colon = bind_addr.rfind(':');
    if (colon == std::string::npos) { fprintf(stderr, "bad bind address\n"); return 2; }
    std::string host = bind_addr.substr(0, colon);
    int port = atoi(bind_addr.c_str() + colon + 1);

    // `contexts` handles + service threads per listed device (default 2): while one context's service thread converts
    // and hands back a finished batch (D2H, f32 -> i16, wake-ups) the other one's batch is on the GPU.  Measured on one
    // MI355X over TCP (tools/worker_load.py): fp32 1754 -> 1788 audio-s/s, bf16 6417 -> 8031, i.e. 93-94 % of the
    // device-resident bench line.  (A device may also be listed more than once.)
    if (contexts < 1) contexts = 1;
    for (size_t p = 0; p <= devices.size();) {
        size_t e = devices.find(',', p);
        if (e == std::string::npos) e = devices.size();
        for (int k = 0; e > p && k < contexts; k++) {
            std::unique_ptr<Gpu> g(new Gpu);
            g->device = atoi(devices.substr(p, e - p).c_str());
            g->voc = hipvoc_load(model.c_str(), g->device, dtype, unit);
            if (!g->voc) { fprintf(stderr, "hipvoc-worker: load on device %d failed: %s\n", g->device, hipvoc_last_error(nullptr)); return 1; }
            g_gpus.push_back(std::move(g));
        }
        p = e + 1;
    }
    if (g_gpus.empty()) { fprintf(stderr, "hipvoc-worker: no device given\n"); return 2; }
    for (auto &g : g_gpus) g->th = std::thread(gpu_service, g.get());

    int ls = socket(AF_INET, SOCK_STREAM, 0);
    int one = 1;
    setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    sockaddr_in sa{};
    sa.sin_family = AF_INET;
    sa.sin_port = htons((uint16_t)port);
    if (inet_pton(AF_INET, host.c_str(), &sa.sin_addr) != 1) { fprintf(stderr, "bad bind host %s\n", host.c_str()); return 2; }
    if (bind(ls, (sockaddr *)&sa, sizeof sa) != 0 || listen(ls, 64) != 0) { perror("hipvoc-worker: bind/listen"); return 1; }
    socklen_t sl = sizeof sa;
    getsockname(ls, (sockaddr *)&sa, &sl);
    printf("hipvoc-worker ready, listening on %s:%d (unit %d frames, %s, %zu device handle(s))\n", host.c_str(),
           (int)ntohs(sa.sin_port), unit, dtype == HIPVOC_DTYPE_BF16 ? "bf16" : dtype == HIPVOC_DTYPE_BF16X3 ? "bf16x3" : "f32", g_gpus.size());
    fflush(stdout);
    signal(SIGPIPE, SIG_IGN);
    signal(SIGTERM, on_term);
    signal(SIGINT, on_term);
    while (!g_stop.load()) {
        pollfd pf{ls, POLLIN, 0};
        if (poll(&pf, 1, 200) <= 0) continue;
        int fd = accept(ls, nullptr, nullptr);
        if (fd < 0) continue;
        std::thread(client, fd).detach();
    }
    // drain: service threads exit once their queues are empty
    for (auto &g : g_gpus) { g->cv_work.notify_all(); g->th.join(); }
    for (auto &g : g_gpus)
        printf("hipvoc-worker stats: device %d batches=%ld jobs=%ld max_batch=%ld\n", g->device, g->batches, g->jobs, g->max_batch);
    fflush(stdout);
    _exit(0);   // client threads may still sit in recv(); nothing left to flush
}

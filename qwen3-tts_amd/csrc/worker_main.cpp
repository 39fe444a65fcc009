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
#include <sys/stat.h>
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

// Request limits, checked BEFORE anything is allocated for a request (the reference never allocates more than the message
// it has already read, mod.rs:666-671; a batched, GPU-backed worker must not let one frame header reserve gigabytes):
static long g_max_frames = 131072;         // --max-frames: code frames per request, summed over a VocodeBatch (2.9 h of audio)
static long g_max_items = 4096;            // --max-items : sequences per VocodeBatch
static int g_max_conns = 256;              // --max-connections: further clients are refused (closed) until one leaves
static std::atomic<int> g_conns{0};

// Completion of one request's jobs, wherever they were queued.
struct Latch {
    std::mutex mu;
    std::condition_variable cv;
    size_t remaining = 0;
};

// One decode job = one utterance ([n, n_q] i64-LE codes).  A VocodeBatch request enqueues one job per item.
struct Job {
    const uint8_t *codes_le = nullptr;
    int n_frames = 0;
    std::shared_ptr<std::vector<int16_t>> buf;      // the PCM of the pass this job was decoded in ...
    size_t off = 0, n = 0;                          // ... and this job's part of it
    int rc = 0;
    std::string err;
    Latch *latch = nullptr;
    const int16_t *pcm() const { return buf ? buf->data() + off : nullptr; }
};

struct Gpu {
    void *voc = nullptr;
    int device = 0;
    std::mutex mu;
    std::condition_variable cv_work;
    std::deque<Job *> q;
    std::atomic<long> load{0};                      // frames queued or in flight
    std::thread th;
    // stats
    long batches = 0, jobs = 0, max_batch = 0;
};
static std::vector<std::unique_ptr<Gpu>> g_gpus;
static std::atomic<bool> g_stop{false};
static int g_max_batch = 256;
static int g_nq = 16;                               // hipvoc_n_codebooks of the loaded decoder

// decodes jobs[0..B) as one batched pass; PCM straight from the GPU (hipvoc_vocode_pcm16_batch: the f32 -> i16 of
// mod.rs:521-524 runs on the device and halves the bytes over PCIe)
static int decode_jobs(Gpu *g, Job *const *jobs, size_t B, std::string &err) {
    std::vector<uint8_t> all;
    std::vector<int32_t> nf(B);
    std::vector<size_t> off(B + 1, 0), n_out(B, 0);
    size_t bytes = 0;
    for (size_t j = 0; j < B; j++) bytes += (size_t)jobs[j]->n_frames * g_nq * 8;
    all.reserve(bytes + 8);
    for (size_t j = 0; j < B; j++) {
        all.insert(all.end(), jobs[j]->codes_le, jobs[j]->codes_le + (size_t)jobs[j]->n_frames * g_nq * 8);
        nf[j] = jobs[j]->n_frames;
        off[j + 1] = off[j] + (size_t)jobs[j]->n_frames * HIPVOC_SAMPLES_PER_TOKEN;
    }
    if (all.empty()) all.push_back(0);
    auto pcm = std::make_shared<std::vector<int16_t>>(off[B] + 1);
    int rc = hipvoc_vocode_pcm16_batch(g->voc, all.data(), nf.data(), (int)B, pcm->data(), off.data(), n_out.data());
    if (rc) {
        err = hipvoc_last_error(g->voc);
        return rc;
    }
    for (size_t j = 0; j < B; j++) {
        jobs[j]->buf = pcm;
        jobs[j]->off = off[j];
        jobs[j]->n = n_out[j];
        jobs[j]->rc = 0;
    }
    return 0;
}

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
        const size_t B = take.size();
        // nothing a request carries may take the service thread (and with it the worker) down: allocation failures and
        // anything else thrown become error responses (mod.rs:246-258)
        try {
            std::string err;
            int rc = decode_jobs(g, take.data(), B, err);      // one batched pass for everything that was waiting
            if (rc && B > 1) {
                // a bad request must not fail its batch neighbours: redo them one by one
                for (size_t j = 0; j < B; j++) {
                    std::string e1;
                    Job *one = take[j];
                    one->rc = decode_jobs(g, &one, 1, e1);
                    one->err = e1;
                }
            } else if (rc) {
                take[0]->rc = rc;
                take[0]->err = err;
            }
        } catch (const std::exception &e) {
            for (Job *j : take) { j->rc = -8; try { j->err = std::string("worker: ") + e.what(); } catch (...) {} j->buf.reset(); }
        } catch (...) {
            for (Job *j : take) { j->rc = -9; j->buf.reset(); }
        }
        long frames = 0;
        for (Job *j : take) frames += j->n_frames > 0 ? j->n_frames : 1;      // the weight run_jobs charged (an empty item counts 1)
        g->load -= frames;
        for (Job *j : take) {
            Latch *l = j->latch;
            std::lock_guard<std::mutex> lk(l->mu);
            if (--l->remaining == 0) l->cv.notify_all();
        }
    }
}

// Queues the jobs of one request and waits for all of them.  Every job goes to the queue with the fewest frames pending
// at that moment, so a VocodeBatch fans out over the GPUs (and contexts) instead of occupying one of them while the
// others idle; single requests spread the same way.  Results come back in request order whatever ran where.
static void run_jobs(std::vector<Job> &jobs) {
    if (jobs.empty()) return;
    Latch latch;
    latch.remaining = jobs.size();
    static std::atomic<unsigned> rr{0};
    const size_t n_q = g_gpus.size();
    std::vector<std::vector<Job *>> per(n_q);
    size_t start = rr++ % n_q;
    for (Job &j : jobs) {
        j.latch = &latch;
        size_t best = start;
        for (size_t i = 0; i < n_q; i++) {
            size_t k = (start + i) % n_q;
            if (g_gpus[k]->load.load() < g_gpus[best]->load.load()) best = k;
        }
        g_gpus[best]->load += j.n_frames > 0 ? j.n_frames : 1;
        per[best].push_back(&j);
    }
    for (size_t i = 0; i < n_q; i++) {
        if (per[i].empty()) continue;
        {
            std::lock_guard<std::mutex> lk(g_gpus[i]->mu);
            for (Job *j : per[i]) g_gpus[i]->q.push_back(j);
        }
        g_gpus[i]->cv_work.notify_one();
    }
    std::unique_lock<std::mutex> lk(latch.mu);
    latch.cv.wait(lk, [&] { return latch.remaining == 0; });
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
static std::vector<uint8_t> handle_impl(const std::vector<uint8_t> &msg) {
    size_t clen = 0;
    uint64_t nt = 0;
    int kind = hipvoc_wire_parse_request(msg.data(), msg.size(), nullptr, 0, &clen, &nt);
    if (kind == HIPVOC_WIRE_PING) return simple(1, "Ping", nullptr);
    if (kind == HIPVOC_WIRE_INIT) return simple(1, "Init", nullptr);
    if (kind == HIPVOC_WIRE_OTHER) return simple(0, "Init", "This worker is a vocoder (hipvoc): request belongs to another role");
    const size_t frame_bytes = (size_t)g_nq * 8;
    if (kind == HIPVOC_WIRE_VOCODE) {
        if (nt > (uint64_t)g_max_frames) return simple(0, "Init", "n_tokens beyond this worker's --max-frames");
        if (clen / frame_bytes < nt) return simple(0, "Init", "codes shorter than n_tokens*16");      // the reference panics on the slice
        std::vector<uint8_t> codes(clen ? clen : 1);                 // (bounded by the message that was read)
        if (hipvoc_wire_parse_request(msg.data(), msg.size(), codes.data(), codes.size(), &clen, &nt) != HIPVOC_WIRE_VOCODE)
            return simple(0, "Init", "Read: Deserialize request failed: malformed message");
        std::vector<Job> jobs(1);
        jobs[0].codes_le = codes.data();                             // i64 little-endian, read in place (mod.rs:513-516)
        jobs[0].n_frames = (int)nt;
        static const bool trace = getenv("HIPVOC_WORKER_TIMING") != nullptr;     // per-request host timing to stderr
        auto t0 = std::chrono::steady_clock::now();
        run_jobs(jobs);
        auto t1 = std::chrono::steady_clock::now();
        if (jobs[0].rc) return simple(0, "Init", jobs[0].err.c_str());
        int16_t dummy = 0;
        const int16_t *pp = jobs[0].n ? jobs[0].pcm() : &dummy;
        std::vector<uint8_t> b(hipvoc_wire_encode_vocode(pp, jobs[0].n, nullptr, 0));
        hipvoc_wire_encode_vocode(pp, jobs[0].n, b.data(), b.size());
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
        if (B > (size_t)g_max_items) return simple(0, "Init", "VocodeBatch holds more items than this worker's --max-items");
        std::vector<uint8_t> codes(msg.size() + 1);                  // (bounded by the message that was read)
        std::vector<size_t> off(B + 1);
        std::vector<uint64_t> nts(B + 1);
        if (hipvoc_wire_parse_vocode_batch(msg.data(), msg.size(), codes.data(), codes.size(), off.data(), nts.data(), B, &B) !=
            HIPVOC_WIRE_VOCODE_BATCH)
            return simple(0, "Init", "Deserialize request failed: malformed VocodeBatch");
        // every item must carry at least n_tokens*16 codes (the reference would panic on the slice; here it is an error
        // response), and the request as a whole stays under --max-frames
        uint64_t total = 0;
        for (size_t j = 0; j < B; j++) {
            if (nts[j] > (uint64_t)g_max_frames || (off[j + 1] - off[j]) / frame_bytes < nts[j]) return simple(0, "Init", "codes shorter than n_tokens*16");
            total += nts[j];
        }
        if (total > (uint64_t)g_max_frames) return simple(0, "Init", "VocodeBatch holds more frames than this worker's --max-frames");
        std::vector<Job> jobs(B);
        for (size_t j = 0; j < B; j++) { jobs[j].codes_le = codes.data() + off[j]; jobs[j].n_frames = (int)nts[j]; }
        run_jobs(jobs);
        std::vector<int16_t> pcm;
        std::vector<size_t> so(B + 1, 0);
        for (size_t j = 0; j < B; j++) {
            if (jobs[j].rc) return simple(0, "Init", jobs[j].err.c_str());
            so[j + 1] = so[j] + jobs[j].n;
        }
        pcm.resize(so[B] + 1);
        for (size_t j = 0; j < B; j++)
            if (jobs[j].n) memcpy(pcm.data() + so[j], jobs[j].pcm(), jobs[j].n * sizeof(int16_t));
        std::vector<uint8_t> b(hipvoc_wire_encode_vocode_batch(pcm.data(), so.data(), B, nullptr, 0));
        hipvoc_wire_encode_vocode_batch(pcm.data(), so.data(), B, b.data(), b.size());
        return b;
    }
    return simple(0, "Init", "Read: Deserialize request failed: malformed message");
}

// The barrier of the request path: handle() runs under std::async and its result is fetched by the connection's writer
// thread; an exception escaping here would be rethrown there and end the whole process (std::terminate).  Whatever a
// request provokes -- bad_alloc included -- is an error response and the worker keeps serving (mod.rs:246-258).
static std::vector<uint8_t> handle(const std::vector<uint8_t> &msg) noexcept {
    try {
        return handle_impl(msg);
    } catch (const std::bad_alloc &) {
    } catch (...) {
    }
    try {
        return simple(0, "Init", "worker: request could not be served (out of memory or internal error)");
    } catch (...) {
        return std::vector<uint8_t>();       // (an empty frame: the client's decoder reports it; nothing more can be done)
    }
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
            std::vector<uint8_t> msg;
            try {
                msg.resize(len);
            } catch (...) {                                    // cannot buffer it: same answer as an oversized frame
                std::promise<std::vector<uint8_t>> pr;
                pr.set_value(simple(0, "Init", "Read: out of memory"));
                fut = pr.get_future();
                last = true;
            }
            if (!last && len && !read_all(fd, msg.data(), len)) break;
            if (!last) {
                // a frame that does not deserialize ends the connection after its error response, as read_request's Err arm
                // does (mod.rs:642-651); requests that deserialize but cannot be served get an error and the connection stays
                size_t cl = 0, nb = 0;
                uint64_t nt = 0;
                int kind = hipvoc_wire_parse_request(msg.data(), msg.size(), nullptr, 0, &cl, &nt);
                if (kind == HIPVOC_WIRE_VOCODE_BATCH &&
                    hipvoc_wire_parse_vocode_batch(msg.data(), msg.size(), nullptr, 0, nullptr, nullptr, 0, &nb) != HIPVOC_WIRE_VOCODE_BATCH)
                    kind = HIPVOC_WIRE_E_MALFORMED;
                if (kind == HIPVOC_WIRE_E_MALFORMED) {
                    std::promise<std::vector<uint8_t>> pr;
                    pr.set_value(simple(0, "Init", "Read: Deserialize request failed: malformed message"));
                    fut = pr.get_future();
                    last = true;
                }
            }
            if (!last)
            try {
                fut = std::async(std::launch::async, [m = std::move(msg)] { return handle(m); });
            } catch (...) {                                    // no thread to be had: refuse this request, keep the worker
                std::promise<std::vector<uint8_t>> pr;
                pr.set_value(simple(0, "Init", "worker: too busy to serve this request"));
                fut = pr.get_future();
            }
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
        else if (a == "--max-frames") g_max_frames = atol(next());
        else if (a == "--max-items") g_max_items = atol(next());
        else if (a == "--max-connections") g_max_conns = atoi(next());
        else if (a == "--models") model = next();
        else if (a == "-r" || a == "--role") { std::string r = next(); if (r != "vocoder") { fprintf(stderr, "hipvoc-worker only serves role 'vocoder'\n"); return 2; } }
        else { fprintf(stderr, "usage: hipvoc-worker -m <decoder.safetensors | models dir> [-b host:port] [--devices 0,1,..] [--dtype f32|bf16|bf16x3] [--unit-frames 64] [--max-batch 256] [--contexts 2] [--max-frames 131072] [--max-items 4096] [--max-connections 256]\n"); return 2; }
    }
    if (model.empty()) { fprintf(stderr, "hipvoc-worker: -m <weights.safetensors | models dir> is required\n"); return 2; }
    {   // a models directory, as the reference's `worker -m <models>` takes (cli.rs:63-65): its discovery (mod.rs:792-813)
        struct stat st;
        if (stat(model.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) {
            char found[4096];
            if (hipvoc_find_vocoder_model(model.c_str(), found, sizeof found) != HIPVOC_OK) { fprintf(stderr, "hipvoc-worker: %s\n", found); return 1; }
            printf("hipvoc-worker: Loading vocoder from %s\n", found);
            model = found;
        }
    }
    if (g_max_frames < 1) g_max_frames = 1;
    if (g_max_items < 1) g_max_items = 1;
    if (g_max_conns < 1) g_max_conns = 1;
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
    // two contexts on a device already are two passes side by side: the handles' own second lane (hipvoc.h, HIPVOC_LANES) would
    // only add a workspace each, so it stays off here unless the operator asks for it
    if (contexts > 1) setenv("HIPVOC_LANES", "1", 0);
    for (size_t p = 0; p <= devices.size();) {
        size_t e = devices.find(',', p);
        if (e == std::string::npos) e = devices.size();
        for (int k = 0; e > p && k < contexts; k++) {
            std::unique_ptr<Gpu> g(new Gpu);
            g->device = atoi(devices.substr(p, e - p).c_str());
            g->voc = hipvoc_load(model.c_str(), g->device, dtype, unit);
            if (!g->voc) { fprintf(stderr, "hipvoc-worker: load on device %d failed: %s\n", g->device, hipvoc_last_error(nullptr)); return 1; }
            g_nq = hipvoc_n_codebooks(g->voc);
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
        if (g_conns.load() >= g_max_conns) { close(fd); continue; }      // bounded: every connection owns threads and buffers
        g_conns++;
        try {
            std::thread([fd] { client(fd); g_conns--; }).detach();
        } catch (...) {                                                  // no thread to be had: refuse this client, keep serving
            g_conns--;
            close(fd);
        }
    }
    // drain: service threads exit once their queues are empty
    for (auto &g : g_gpus) { g->cv_work.notify_all(); g->th.join(); }
    for (auto &g : g_gpus)
        printf("hipvoc-worker stats: device %d batches=%ld jobs=%ld max_batch=%ld\n", g->device, g->batches, g->jobs, g->max_batch);
    fflush(stdout);
    _exit(0);   // client threads may still sit in recv(); nothing left to flush
}

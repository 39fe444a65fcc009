// hipvoc-worker -- a drop-in for `qwen3-tts worker -r vocoder -b HOST:PORT` (reference: src/worker/mod.rs:552-698).
//
// Speaks the reference's TCP protocol (u32 BE length + rmp-serde payload, include/hipvoc_wire.h) for the Vocoder
// role and nothing else: Ping, Init, Vocode (+ the VocodeBatch extension); requests of other roles get the
// reference's error response and the connection stays up.  One thread per client, requests serialised behind
// one mutex exactly like the reference's Arc<Mutex<Worker>> (mod.rs:567-577,628-631); the decode itself is
// libhipvoc (include/hipvoc.h).  The orchestrator (pipeline.rs, the HTTP API) needs no change: point
// `--vocoder-ip` at this process.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <signal.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hipvoc.h"
#include "../../include/hipvoc_wire.h"

static void *g_voc = nullptr;
static std::mutex g_mu;
static const size_t MAX_MSG = 100000000;   // mod.rs:666

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
        std::vector<int16_t> pcm((size_t)nt * HIPVOC_SAMPLES_PER_TOKEN + 1);
        size_t n_samples = 0;
        int rc;
        std::string err;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            rc = hipvoc_vocode_pcm16(g_voc, codes.data(), (int)nt, pcm.data(), pcm.size(), &n_samples);
            if (rc) err = hipvoc_last_error(g_voc);
        }
        if (rc) return simple(0, "Init", err.c_str());
        std::vector<uint8_t> b(hipvoc_wire_encode_vocode(pcm.data(), n_samples, nullptr, 0));
        hipvoc_wire_encode_vocode(pcm.data(), n_samples, b.data(), b.size());
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
        std::vector<float> wav(out_off[B] + 1);
        int rc;
        std::string err;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            rc = hipvoc_decode_batch(g_voc, all.data(), nf.data(), (int)B, wav.data(), out_off.data(), n_out.data());
            if (rc) err = hipvoc_last_error(g_voc);
        }
        if (rc) return simple(0, "Init", err.c_str());
        std::vector<int16_t> pcm;
        std::vector<size_t> so(B + 1, 0);
        for (size_t j = 0; j < B; j++) {
            for (size_t k = 0; k < n_out[j]; k++) pcm.push_back(pcm16(wav[out_off[j] + k]));
            so[j + 1] = pcm.size();
        }
        if (pcm.empty()) pcm.push_back(0);
        std::vector<uint8_t> b(hipvoc_wire_encode_vocode_batch(pcm.data(), so.data(), B, nullptr, 0));
        hipvoc_wire_encode_vocode_batch(pcm.data(), so.data(), B, b.data(), b.size());
        return b;
    }
    return simple(0, "Init", "Read: Deserialize request failed: malformed message");
}

static void client(int fd) {
    int one = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    std::vector<uint8_t> msg;
    for (;;) {
        uint8_t lb[4];
        if (!read_all(fd, lb, 4)) break;                       // client went away (mod.rs:662-664)
        size_t len = ((size_t)lb[0] << 24) | ((size_t)lb[1] << 16) | ((size_t)lb[2] << 8) | lb[3];
        if (len > MAX_MSG) {                                   // mod.rs:666-668, 603-611: error response, then close
            send_frame(fd, simple(0, "Init", "Read: Message too large"));
            break;
        }
        msg.resize(len);
        if (len && !read_all(fd, msg.data(), len)) break;
        if (!send_frame(fd, handle(msg))) break;
    }
    close(fd);
}

int main(int argc, char **argv) {
    std::string bind_addr = "0.0.0.0:9092", model;
    int device = 0, dtype = HIPVOC_DTYPE_F32, unit = 64;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "-b" || a == "--bind") bind_addr = next();
        else if (a == "-m" || a == "--model") model = next();
        else if (a == "--device") device = atoi(next());
        else if (a == "--dtype") { std::string d = next(); dtype = d == "bf16" ? HIPVOC_DTYPE_BF16 : HIPVOC_DTYPE_F32; }
        else if (a == "--unit-frames") unit = atoi(next());
        else if (a == "-r" || a == "--role") { std::string r = next(); if (r != "vocoder") { fprintf(stderr, "hipvoc-worker only serves role 'vocoder'\n"); return 2; } }
        else { fprintf(stderr, "usage: hipvoc-worker -m <decoder.safetensors> [-b host:port] [--device N] [--dtype f32|bf16] [--unit-frames 64]\n"); return 2; }
    }
    if (model.empty()) { fprintf(stderr, "hipvoc-worker: -m <weights.safetensors> is required\n"); return 2; }
    size_t colon = bind_addr.rfind(':');
    if (colon == std::string::npos) { fprintf(stderr, "bad bind address\n"); return 2; }
    std::string host = bind_addr.substr(0, colon);
    int port = atoi(bind_addr.c_str() + colon + 1);

    g_voc = hipvoc_load(model.c_str(), device, dtype, unit);
    if (!g_voc) { fprintf(stderr, "hipvoc-worker: load failed: %s\n", hipvoc_last_error(nullptr)); return 1; }

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
    printf("hipvoc-worker ready, listening on %s:%d (unit %d frames, %s)\n", host.c_str(), (int)ntohs(sa.sin_port), unit,
           dtype == HIPVOC_DTYPE_BF16 ? "bf16" : "f32");
    fflush(stdout);
    signal(SIGPIPE, SIG_IGN);
    for (;;) {
        int fd = accept(ls, nullptr, nullptr);
        if (fd < 0) continue;
        std::thread(client, fd).detach();
    }
}

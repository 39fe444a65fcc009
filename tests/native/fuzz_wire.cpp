// Host-only robustness check of the wire codec (qwen3-tts_amd/csrc/wire.cpp) under AddressSanitizer + UBSan:
// well-formed requests, every truncation of them, and a few hundred thousand random mutations must never read or
// write out of bounds -- the worker feeds these parsers bytes straight from a TCP socket.  Built and run by
// tests/test_wire_cpu.py (g++ -fsanitize=address,undefined; no GPU code involved).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/hipvoc_wire.h"

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

static void put_str(std::vector<uint8_t> &v, const char *s) {
    size_t n = strlen(s);
    v.push_back((uint8_t)(0xa0 | n));
    v.insert(v.end(), s, s + n);
}
static void put_u8_array(std::vector<uint8_t> &v, size_t n) {   // Vec<u8> as rmp-serde writes it: array of ints
    v.push_back(0xdc); v.push_back((uint8_t)(n >> 8)); v.push_back((uint8_t)n);
    for (size_t i = 0; i < n; i++) {
        uint8_t b = (uint8_t)rnd();
        if (b >= 0x80) v.push_back(0xcc);
        v.push_back(b);
    }
}

static std::vector<uint8_t> make_vocode(size_t n_tokens) {
    std::vector<uint8_t> v;
    v.push_back(0x81); put_str(v, "Vocode");
    v.push_back(0x92);
    put_u8_array(v, n_tokens * 128);
    v.push_back((uint8_t)n_tokens);
    return v;
}
static std::vector<uint8_t> make_batch(size_t items) {
    std::vector<uint8_t> v;
    v.push_back(0x81); put_str(v, "VocodeBatch");
    v.push_back(0x92);
    v.push_back((uint8_t)(0x90 | items));
    for (size_t i = 0; i < items; i++) put_u8_array(v, (i + 1) * 128);
    v.push_back((uint8_t)(0x90 | items));
    for (size_t i = 0; i < items; i++) v.push_back((uint8_t)(i + 1));
    return v;
}

static long calls = 0;
static void feed(const uint8_t *p, size_t n, size_t cap) {
    // exact-size heap copies, so that a one-byte overrun of the message is an ASan report
    uint8_t *msg = (uint8_t *)malloc(n ? n : 1);
    if (n) memcpy(msg, p, n);
    uint8_t *codes = (uint8_t *)malloc(cap ? cap : 1);
    size_t len = 0, items = 0, offs[9];
    uint64_t nt = 0, nts[8];
    int k = hipvoc_wire_parse_request(msg, n, codes, cap, &len, &nt);
    if (k == HIPVOC_WIRE_VOCODE && len > cap) { fprintf(stderr, "codes_len beyond capacity\n"); abort(); }
    k = hipvoc_wire_parse_vocode_batch(msg, n, codes, cap, offs, nts, 8, &items);
    if (k >= 0 && (items > 8 || offs[items] > cap)) { fprintf(stderr, "batch beyond capacity\n"); abort(); }
    free(msg); free(codes);
    calls++;
}

int main() {
    std::vector<std::vector<uint8_t>> seeds;
    seeds.push_back(make_vocode(1));
    seeds.push_back(make_vocode(5));
    seeds.push_back(make_batch(3));
    seeds.push_back({0xa4, 'P', 'i', 'n', 'g'});
    {
        std::vector<uint8_t> v; v.push_back(0x81); put_str(v, "Init"); v.push_back(0x92); put_str(v, "vocoder"); put_str(v, "/models");
        seeds.push_back(v);
    }
    for (auto &s : seeds) {
        for (size_t cut = 0; cut <= s.size(); cut++) feed(s.data(), cut, 4096);      // every truncation
        feed(s.data(), s.size(), 0);
        feed(s.data(), s.size(), 100);                                                // capacity too small
    }
    for (int it = 0; it < 200000; it++) {
        std::vector<uint8_t> m = seeds[rnd() % seeds.size()];
        int nm = 1 + rnd() % 4;
        for (int j = 0; j < nm && !m.empty(); j++) {
            size_t pos = rnd() % m.size();
            switch (rnd() % 4) {
                case 0: m[pos] = (uint8_t)rnd(); break;
                case 1: m[pos] ^= (uint8_t)(1u << (rnd() % 8)); break;
                case 2: m.erase(m.begin() + pos); break;
                default: m.insert(m.begin() + pos, (uint8_t)rnd()); break;
            }
        }
        feed(m.data(), m.size(), rnd() % 2 ? 4096 : rnd() % 700);
    }
    for (int it = 0; it < 20000; it++) {       // pure noise, with length prefixes that promise gigabytes
        std::vector<uint8_t> m(rnd() % 64);
        for (auto &b : m) b = (uint8_t)rnd();
        feed(m.data(), m.size(), 256);
    }
    // encoders: exact-capacity and too-small buffers
    std::vector<int16_t> pcm(5000, 1234);
    for (size_t cap : {size_t(0), size_t(10), size_t(20000), size_t(40000)}) {
        uint8_t *out = (uint8_t *)malloc(cap ? cap : 1);
        // (the encoders return the size they need; `out` is an exact-size heap block, so a write past cap is an ASan report)
        size_t need = hipvoc_wire_encode_vocode(pcm.data(), pcm.size(), out, cap);
        size_t offs[3] = {0, 2000, 5000};
        need += hipvoc_wire_encode_vocode_batch(pcm.data(), offs, 2, out, cap);
        need += hipvoc_wire_encode_simple(0, "Ping", "some error text", out, cap);
        if (!need) abort();
        free(out);
    }
    printf("fuzz_wire OK %ld calls\n", calls);
    return 0;
}

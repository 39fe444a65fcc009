// Host-only check of the decode entry points' host logic (qwen3-tts_amd/csrc/chunking.hpp: the reference's chunk loop,
// src/worker/vocoder.rs:60-85, as the arithmetic hipvoc_decode / decode_batch / vocode_pcm16* run before any device
// work) under AddressSanitizer + UBSan, against a literal restatement of that loop.  Also the exception barrier of the
// C ABI (abi_guard.hpp): injected std::bad_alloc / other exceptions must come back as error codes, never propagate.
// Built and run by tests/test_host_cpu.py (g++ -fsanitize=address,undefined; no GPU code involved).
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../qwen3-tts_amd/csrc/abi_guard.hpp"
#include "../../qwen3-tts_amd/csrc/chunking.hpp"

using namespace hipvoc;

static uint64_t rng_state = 0x243f6a8885a308d3ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #c); abort(); } } while (0)

// vocoder.rs:65-82, literally
struct RefUnit { int seq; int64_t start; int len; size_t keep; size_t dst; };
static std::vector<RefUnit> reference_units(const std::vector<int32_t> &nf, int T, int64_t Lu, const std::vector<size_t> &off) {
    std::vector<RefUnit> r;
    for (size_t b = 0; b < nf.size(); b++) {
        int64_t pos = 0;
        size_t written = 0;
        while (pos < nf[b]) {
            int chunk_len = (int)std::min<int64_t>(T, nf[b] - pos);
            size_t expected = (size_t)chunk_len * 1920;
            size_t actual = std::min<size_t>(expected, (size_t)Lu);
            r.push_back({(int)b, pos, chunk_len, actual, off[b] + written});
            written += actual;
            pos += chunk_len;
        }
    }
    return r;
}

int main() {
    long cases = 0;
    for (int it = 0; it < 20000; it++) {
        const int T = 1 + (int)(rnd() % 70);
        // both trim policies: unit_samples below T*1920 (the .min() of vocoder.rs:80 is live) and exactly T*1920
        const int64_t Lu = (rnd() & 1) ? (int64_t)T * 1920 : std::max<int64_t>(1, (int64_t)T * 1920 - (int64_t)(rnd() % 2000));
        const int B = (int)(rnd() % 6);
        std::vector<int32_t> nf((size_t)B);
        std::vector<size_t> off((size_t)B);
        size_t o = 0;
        for (int b = 0; b < B; b++) {
            nf[(size_t)b] = (rnd() % 5 == 0) ? 0 : (int32_t)(rnd() % (5 * (unsigned)T + 3));
            off[(size_t)b] = o;
            o += (size_t)nf[(size_t)b] * 1920;
        }
        std::vector<RefUnit> ref = reference_units(nf, T, Lu, off);
        UnitWalker w(nf.data(), B, T, Lu, off.data());
        UnitRef u;
        size_t i = 0;
        std::vector<size_t> total((size_t)B, 0);
        while (w.next(u)) {
            CHECK(i < ref.size());
            CHECK(u.seq == ref[i].seq && u.start == ref[i].start && u.len == ref[i].len && u.keep == ref[i].keep && u.dst == ref[i].dst);
            CHECK(u.dst + u.keep <= off[(size_t)u.seq] + (size_t)nf[(size_t)u.seq] * 1920);      // never beyond the sequence's capacity
            total[(size_t)u.seq] += u.keep;
            i++;
        }
        CHECK(i == ref.size());
        CHECK(!w.next(u));
        for (int b = 0; b < B; b++) {
            size_t smp = 0, nu = 0;
            CHECK(sequence_extent(nf[(size_t)b], T, Lu, &smp, &nu));
            CHECK(smp == total[(size_t)b]);
        }
        cases++;
    }
    {   // extremes: no overflow, no allocation
        size_t smp = 0, nu = 0;
        CHECK(sequence_extent(INT_MAX, 64, 122325, &smp, &nu));
        CHECK(nu == (size_t)INT_MAX / 64 + 1 && smp == (size_t)(INT_MAX / 64) * 122325 + (size_t)(INT_MAX % 64) * 1920);
        CHECK(sequence_extent(INT_MAX, 1, 1920, &smp, &nu) && smp == (size_t)INT_MAX * 1920);
        CHECK(!sequence_extent(-1, 64, 122325, &smp, &nu));
        CHECK(!sequence_extent(5, 0, 122325, &smp, &nu));
        CHECK(sequence_extent(0, 64, 122325, &smp, &nu) && smp == 0 && nu == 0);
    }
    {   // code validation on exact-size heap buffers
        for (int it = 0; it < 2000; it++) {
            size_t n = rnd() % 200;
            // (odd byte offset: the wire's code bytes have no alignment; exact-size buffer: an over-read is an ASan report)
            unsigned char *raw = (unsigned char *)malloc(n * sizeof(int64_t) + 1);
            struct Unaligned {
                unsigned char *p;
                struct Ref {
                    unsigned char *q;
                    void operator=(int64_t v) { memcpy(q, &v, 8); }
                };
                Ref operator[](size_t k) { return Ref{p + 8 * k}; }
            } c{raw + 1};
            for (size_t k = 0; k < n; k++) c[k] = (int64_t)(rnd() % 2048);
            CHECK(first_bad_code(c.p, n, 2048) == n);
            if (n) {
                size_t at = rnd() % n;
                c[at] = (rnd() & 1) ? 2048 + (int64_t)(rnd() % 5) : -1 - (int64_t)(rnd() % 5);
                int64_t seen = 0;
                CHECK(first_bad_code(c.p, n, 2048, &seen) == at && (seen >= 2048 || seen < 0));
                c[at] = INT64_MIN;
                CHECK(first_bad_code(c.p, n, 2048) == at);
            }
            free(raw);
        }
    }
    {   // the exception barrier
        std::string text;
        auto note = [&](const char *w) noexcept { try { text = w; } catch (...) {} };
        CHECK(abi_guard([]() -> int { return 7; }, note) == 7 && text.empty());
        CHECK(abi_guard([]() -> int { throw std::bad_alloc(); }, note) == HIPVOC_E_NOMEM && text == "out of host memory");
        CHECK(abi_guard([]() -> int { std::vector<char> v; v.reserve((size_t)-1); return 0; }, note) == HIPVOC_E_INTERNAL);      // std::length_error
        CHECK(abi_guard([]() -> int { throw std::runtime_error("boom"); }, note) == HIPVOC_E_INTERNAL && text == "boom");
        CHECK(abi_guard([]() -> int { throw 42; }, note) == HIPVOC_E_INTERNAL && text == "unknown C++ exception");
    }
    printf("fuzz_chunking OK (%ld random batches)\n", cases);
    return 0;
}

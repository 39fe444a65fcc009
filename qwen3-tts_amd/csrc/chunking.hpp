// chunking.hpp -- the reference's chunk loop (src/worker/vocoder.rs:60-85) as pure host arithmetic: no HIP, no allocation.
// Shared by the C ABI's host entry points (hipvoc.cpp) and fuzzed natively under ASan/UBSan (tests/native/fuzz_chunking.cpp).
//
//   while pos < n_tokens { chunk_len = min(max_tokens, n_tokens - pos); pad with code 0; run;
//                          keep min(chunk_len * 1920, unit_samples); pos += chunk_len }
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace hipvoc {

constexpr int64_t kSamplesPerToken = 1920;      // src/worker/vocoder.rs:6

// One decode unit of one sequence.
struct UnitRef {
    int seq;          // sequence index in the batch
    int64_t start;    // first frame of the unit within its sequence
    int len;          // real frames (1..T); the rest of the unit is code 0
    size_t keep;      // samples kept: min(len * 1920, unit_samples)            (vocoder.rs:79-80)
    size_t dst;       // element offset in the caller's output: out_off[seq] + samples kept by the sequence's earlier units
};

// Samples a sequence of n frames yields, and its unit count.  false on overflow / bad arguments.
inline bool sequence_extent(int64_t n, int T, int64_t unit_samples, size_t *samples, size_t *units) {
    if (n < 0 || T < 1 || unit_samples < 0) return false;
    const uint64_t full = (uint64_t)(n / T), rem = (uint64_t)(n % T);
    const uint64_t keep_full = (uint64_t)((int64_t)T * kSamplesPerToken < unit_samples ? (int64_t)T * kSamplesPerToken : unit_samples);
    const uint64_t keep_rem = rem * (uint64_t)kSamplesPerToken < (uint64_t)unit_samples ? rem * (uint64_t)kSamplesPerToken : (uint64_t)unit_samples;
    uint64_t s = 0;
    if (__builtin_mul_overflow(full, keep_full, &s) || __builtin_add_overflow(s, rem ? keep_rem : 0, &s)) return false;
    if (s > (uint64_t)SIZE_MAX / 8) return false;
    *samples = (size_t)s;
    *units = (size_t)(full + (rem ? 1 : 0));
    return true;
}

// Walks the units of B sequences in request order without materialising the list.
class UnitWalker {
  public:
    UnitWalker(const int32_t *n_frames, int B, int T, int64_t unit_samples, const size_t *out_off)
        : nf_(n_frames), off_(out_off), B_(B), T_(T), Lu_(unit_samples) {}
    bool next(UnitRef &u) {
        while (b_ < B_ && start_ >= (int64_t)nf_[b_]) { b_++; start_ = 0; pos_ = 0; }
        if (b_ >= B_) return false;
        const int64_t left = (int64_t)nf_[b_] - start_;
        const int len = (int)(left < T_ ? left : T_);
        const int64_t want = (int64_t)len * kSamplesPerToken;
        u.seq = b_;
        u.start = start_;
        u.len = len;
        u.keep = (size_t)(want < Lu_ ? want : Lu_);
        u.dst = (off_ ? off_[b_] : 0) + pos_;
        pos_ += u.keep;
        start_ += len;
        return true;
    }

  private:
    const int32_t *nf_;
    const size_t *off_;
    int B_, T_;
    int64_t Lu_;
    int b_ = 0;
    int64_t start_ = 0;
    size_t pos_ = 0;
};

// Index of the first code outside [0, cb_size), or n if all are valid.  `codes` are n i64 values in host byte order at
// ANY alignment (the wire hands them over as bytes, src/worker/mod.rs:513-516): read with memcpy.
inline size_t first_bad_code(const void *codes, size_t n, int64_t cb_size, int64_t *value = nullptr) {
    const unsigned char *p = static_cast<const unsigned char *>(codes);
    for (size_t i = 0; i < n; i++) {
        int64_t v;
        __builtin_memcpy(&v, p + i * sizeof(int64_t), sizeof v);
        if ((uint64_t)v >= (uint64_t)cb_size) {
            if (value) *value = v;
            return i;
        }
    }
    return n;
}

}  // namespace hipvoc

// Host-only robustness check of the safetensors reader (qwen3-tts_amd/csrc/safetensors.cpp) under ASan + UBSan:
// a valid file, every header truncation, and mutated headers / lengths / offsets must either load or fail with a
// message -- never read outside the mapping.  argv[1] = a valid .safetensors file, argv[2] = scratch path.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../qwen3-tts_amd/csrc/safetensors.hpp"

static uint64_t rng_state = 0x2545f4914f6cdd1dull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

static long loads = 0, oks = 0;
static void try_file(const std::vector<uint8_t> &bytes, const char *scratch) {
    {
        std::ofstream f(scratch, std::ios::binary | std::ios::trunc);
        f.write((const char *)bytes.data(), (std::streamsize)bytes.size());
    }
    hipvoc::SafeTensors st;
    std::string err;
    loads++;
    if (!st.open(scratch, err)) {
        if (err.empty()) { fprintf(stderr, "failure without a message\n"); abort(); }
        return;
    }
    oks++;
    // touch every byte of every tensor the reader accepted, and convert them
    for (const char *name : {"a.weight", "b.bias", "c", "nope"}) {
        const hipvoc::StTensor *t = st.find(name);
        if (!t) continue;
        volatile uint8_t sink = 0;
        for (size_t i = 0; i < t->nbytes; i++) sink ^= t->data[i];
        std::vector<float> v;
        std::vector<int64_t> sh;
        std::string e2;
        (void)st.to_f32(name, v, sh, e2);
    }
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> good((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (good.size() < 16) return 2;
    try_file(good, argv[2]);
    if (oks != 1) { fprintf(stderr, "the valid file did not load\n"); return 1; }
    uint64_t hlen = 0;
    memcpy(&hlen, good.data(), 8);
    for (size_t cut = 0; cut < good.size(); cut += (cut < 8 + hlen ? 1 : 97)) {       // truncations
        std::vector<uint8_t> m(good.begin(), good.begin() + cut);
        try_file(m, argv[2]);
    }
    for (int it = 0; it < 20000; it++) {
        std::vector<uint8_t> m = good;
        int nm = 1 + rnd() % 3;
        for (int j = 0; j < nm; j++) {
            size_t span = rnd() % 4 ? 8 + (size_t)hlen : 8;        // mostly the header, sometimes the length field
            size_t pos = rnd() % span;
            switch (rnd() % 5) {
                case 0: m[pos] = (uint8_t)rnd(); break;
                case 1: m[pos] ^= (uint8_t)(1u << (rnd() % 8)); break;
                case 2: if (pos >= 8) m[pos] = "0123456789"[rnd() % 10]; break;      // digits: offsets and shapes
                case 3: if (pos >= 8) m.erase(m.begin() + pos); break;
                default: if (pos >= 8) m.insert(m.begin() + pos, (uint8_t)"9[]{},:\"-"[rnd() % 9]); break;
            }
        }
        try_file(m, argv[2]);
    }
    // hand-made hostile headers: products of dims that wrap around 2^64, integers beyond int64, dims beyond
    // int, an unknown dtype with an absurd shape, unbounded nesting -- refused with a message, nothing sized from them
    auto with_header = [](const std::string &h, size_t data) {
        std::vector<uint8_t> m(8 + h.size() + data, 0);
        uint64_t n = h.size();
        memcpy(m.data(), &n, 8);
        memcpy(m.data() + 8, h.data(), h.size());
        return m;
    };
    const long before = oks;
    const char *hostile[] = {
        "{\"a.weight\":{\"dtype\":\"F32\",\"shape\":[4611686018427387904,4],\"data_offsets\":[0,0]}}",
        "{\"a.weight\":{\"dtype\":\"F32\",\"shape\":[4294967296,4294967296,1],\"data_offsets\":[0,0]}}",
        "{\"a.weight\":{\"dtype\":\"F32\",\"shape\":[99999999999999999999999999,1],\"data_offsets\":[0,4]}}",
        "{\"a.weight\":{\"dtype\":\"F32\",\"shape\":[3000000000],\"data_offsets\":[0,4]}}",
        "{\"a.weight\":{\"dtype\":\"I64\",\"shape\":[2147483647,2147483647,2147483647],\"data_offsets\":[0,8]}}",
        "{\"a.weight\":{\"dtype\":\"F32\",\"shape\":[1],\"data_offsets\":[0,18446744073709551620]}}",
    };
    for (const char *h : hostile) try_file(with_header(h, 16), argv[2]);
    if (oks != before) { fprintf(stderr, "a hostile header was accepted\n"); return 1; }
    {
        std::string deep = "{\"__metadata__\":{\"k\":";
        deep += std::string(200000, '[');
        deep += "}}";
        try_file(with_header(deep, 0), argv[2]);
        if (oks != before) { fprintf(stderr, "unbounded nesting was accepted\n"); return 1; }
    }
    printf("fuzz_safetensors OK %ld files, %ld accepted\n", loads, oks);
    return 0;
}

// wire.cpp -- minimal MessagePack codec for the reference's worker protocol (include/hipvoc_wire.h).
// No device code; linked into libhipvoc and the hipvoc-worker binary.
#include "../../include/hipvoc_wire.h"

#include <cstring>
#include <string>
#include <vector>

namespace {

struct Rd {
    const uint8_t *p;
    size_t n, i = 0;
    bool bad = false;
    bool need(size_t k) { if (n - i < k) { bad = true; return false; } return true; }
    uint8_t peek() { if (!need(1)) return 0xc1; return p[i]; }
    uint64_t be(int k) {
        if (!need((size_t)k)) return 0;
        uint64_t v = 0;
        for (int j = 0; j < k; j++) v = (v << 8) | p[i + j];
        i += (size_t)k;
        return v;
    }
};

// header readers: return true and the element count when the next object is of that family
static bool rd_array(Rd &r, size_t &cnt) {
    uint8_t b = r.peek();
    if ((b & 0xf0) == 0x90) { r.i++; cnt = b & 0x0f; return true; }
    if (b == 0xdc) { r.i++; cnt = (size_t)r.be(2); return !r.bad; }
    if (b == 0xdd) { r.i++; cnt = (size_t)r.be(4); return !r.bad; }
    return false;
}
static bool rd_map(Rd &r, size_t &cnt) {
    uint8_t b = r.peek();
    if ((b & 0xf0) == 0x80) { r.i++; cnt = b & 0x0f; return true; }
    if (b == 0xde) { r.i++; cnt = (size_t)r.be(2); return !r.bad; }
    if (b == 0xdf) { r.i++; cnt = (size_t)r.be(4); return !r.bad; }
    return false;
}
static bool rd_str(Rd &r, std::string &s) {
    uint8_t b = r.peek();
    size_t len;
    if ((b & 0xe0) == 0xa0) { r.i++; len = b & 0x1f; }
    else if (b == 0xd9) { r.i++; len = (size_t)r.be(1); }
    else if (b == 0xda) { r.i++; len = (size_t)r.be(2); }
    else if (b == 0xdb) { r.i++; len = (size_t)r.be(4); }
    else return false;
    if (r.bad || !r.need(len)) return false;
    s.assign((const char *)r.p + r.i, len);
    r.i += len;
    return true;
}
static bool rd_uint(Rd &r, uint64_t &v) {
    uint8_t b = r.peek();
    if (b < 0x80) { r.i++; v = b; return true; }
    if (b == 0xcc) { r.i++; v = r.be(1); return !r.bad; }
    if (b == 0xcd) { r.i++; v = r.be(2); return !r.bad; }
    if (b == 0xce) { r.i++; v = r.be(4); return !r.bad; }
    if (b == 0xcf) { r.i++; v = r.be(8); return !r.bad; }
    // non-negative values in signed encodings
    if (b == 0xd0) { r.i++; int8_t x = (int8_t)r.be(1); if (x < 0) return false; v = (uint64_t)x; return !r.bad; }
    if (b == 0xd1) { r.i++; int16_t x = (int16_t)r.be(2); if (x < 0) return false; v = (uint64_t)x; return !r.bad; }
    if (b == 0xd2) { r.i++; int32_t x = (int32_t)r.be(4); if (x < 0) return false; v = (uint64_t)x; return !r.bad; }
    if (b == 0xd3) { r.i++; int64_t x = (int64_t)r.be(8); if (x < 0) return false; v = (uint64_t)x; return !r.bad; }
    return false;
}
// a Vec<u8>: array of ints 0..255 (serde default) or bin (serde_bytes)
static bool rd_bytes(Rd &r, std::vector<uint8_t> &out) {
    uint8_t b = r.peek();
    size_t len;
    if (b == 0xc4 || b == 0xc5 || b == 0xc6) {
        r.i++;
        len = (size_t)r.be(b == 0xc4 ? 1 : b == 0xc5 ? 2 : 4);
        if (r.bad || !r.need(len)) return false;
        out.insert(out.end(), r.p + r.i, r.p + r.i + len);
        r.i += len;
        return true;
    }
    if (!rd_array(r, len)) return false;
    if (len > r.n - r.i) return false;          // every element takes at least one byte
    out.reserve(out.size() + len);
    for (size_t k = 0; k < len; k++) {
        if (r.i >= r.n) return false;
        uint8_t e = r.p[r.i];
        if (e < 0x80) { out.push_back(e); r.i++; continue; }       // the common case
        uint64_t v;
        if (!rd_uint(r, v) || v > 255) return false;
        out.push_back((uint8_t)v);
    }
    return true;
}
// the same into a caller's buffer, without any allocation: out == null counts only; bytes beyond cap are counted, not
// written (overflow = true)
static bool rd_bytes_to(Rd &r, uint8_t *out, size_t cap, size_t &n_out, bool &overflow) {
    uint8_t b = r.peek();
    size_t len;
    n_out = 0;
    if (b == 0xc4 || b == 0xc5 || b == 0xc6) {
        r.i++;
        len = (size_t)r.be(b == 0xc4 ? 1 : b == 0xc5 ? 2 : 4);
        if (r.bad || !r.need(len)) return false;
        if (out) {
            if (len > cap) overflow = true;
            else if (len) memcpy(out, r.p + r.i, len);
        }
        r.i += len;
        n_out = len;
        return true;
    }
    if (!rd_array(r, len)) return false;
    if (len > r.n - r.i) return false;          // every element takes at least one byte
    const bool wr = out && len <= cap;
    if (out && len > cap) overflow = true;
    for (size_t k = 0; k < len; k++) {
        if (r.i >= r.n) return false;
        uint8_t e = r.p[r.i];
        if (e < 0x80) { if (wr) out[k] = e; r.i++; continue; }       // the common case
        uint64_t v;
        if (!rd_uint(r, v) || v > 255) return false;
        if (wr) out[k] = (uint8_t)v;
    }
    n_out = len;
    return true;
}
// skip any object (bounded depth)
static bool skip(Rd &r, int depth = 0) {
    if (depth > 32) return false;
    uint8_t b = r.peek();
    if (r.bad) return false;
    size_t cnt;
    std::string s;
    if (b < 0x80 || b >= 0xe0) { r.i++; return true; }
    if ((b & 0xf0) == 0x90 || b == 0xdc || b == 0xdd) {
        if (!rd_array(r, cnt)) return false;
        for (size_t k = 0; k < cnt; k++) if (!skip(r, depth + 1)) return false;
        return true;
    }
    if ((b & 0xf0) == 0x80 || b == 0xde || b == 0xdf) {
        if (!rd_map(r, cnt)) return false;
        for (size_t k = 0; k < 2 * cnt; k++) if (!skip(r, depth + 1)) return false;
        return true;
    }
    if ((b & 0xe0) == 0xa0 || b == 0xd9 || b == 0xda || b == 0xdb) return rd_str(r, s);
    switch (b) {
        case 0xc0: case 0xc2: case 0xc3: r.i++; return true;
        case 0xc4: case 0xc5: case 0xc6: { std::vector<uint8_t> t; return rd_bytes(r, t); }
        case 0xca: r.i++; r.be(4); return !r.bad;
        case 0xcb: r.i++; r.be(8); return !r.bad;
        case 0xcc: case 0xd0: r.i++; r.be(1); return !r.bad;
        case 0xcd: case 0xd1: r.i++; r.be(2); return !r.bad;
        case 0xce: case 0xd2: r.i++; r.be(4); return !r.bad;
        case 0xcf: case 0xd3: r.i++; r.be(8); return !r.bad;
        case 0xd4: r.i++; r.need(2); r.i += 2; return !r.bad;
        case 0xd5: r.i++; r.need(3); r.i += 3; return !r.bad;
        case 0xd6: r.i++; r.need(5); r.i += 5; return !r.bad;
        case 0xd7: r.i++; r.need(9); r.i += 9; return !r.bad;
        case 0xd8: r.i++; r.need(17); r.i += 17; return !r.bad;
        case 0xc7: { r.i++; size_t l = (size_t)r.be(1); if (!r.need(l + 1)) return false; r.i += l + 1; return true; }
        case 0xc8: { r.i++; size_t l = (size_t)r.be(2); if (!r.need(l + 1)) return false; r.i += l + 1; return true; }
        case 0xc9: { r.i++; size_t l = (size_t)r.be(4); if (!r.need(l + 1)) return false; r.i += l + 1; return true; }
        default: return false;
    }
}

static bool known_variant(const std::string &v) {
    static const char *names[] = {"Init", "TokenizeAndEmbed", "TalkerPrefill", "TalkerStep", "CodePredict", "Vocode",
                                  "LoadVoice", "Ping", "VocodeBatch"};
    for (const char *n : names) if (v == n) return true;
    return false;
}

// reads the variant name; on return the reader sits at the payload (has_payload) or at the end
static bool rd_variant(Rd &r, std::string &name, bool &has_payload) {
    size_t cnt;
    has_payload = false;
    if (rd_str(r, name)) return true;
    if (r.bad) return false;
    if (rd_map(r, cnt)) {
        if (cnt != 1 || !rd_str(r, name)) return false;
        has_payload = true;
        return true;
    }
    return false;
}

// struct-variant payload: positional array, or map keyed by field name
template <typename F>
static bool rd_fields(Rd &r, const char *const *fields, size_t nf, F &&on_field) {
    size_t cnt;
    if (rd_array(r, cnt)) {
        if (cnt != nf) return false;
        for (size_t k = 0; k < nf; k++) if (!on_field(k)) return false;
        return true;
    }
    if (r.bad) return false;
    if (rd_map(r, cnt)) {
        std::vector<bool> seen(nf, false);
        for (size_t k = 0; k < cnt; k++) {
            std::string key;
            if (!rd_str(r, key)) return false;
            size_t f = 0;
            while (f < nf && key != fields[f]) f++;
            if (f == nf) { if (!skip(r)) return false; continue; }
            if (seen[f] || !on_field(f)) return false;
            seen[f] = true;
        }
        for (size_t f = 0; f < nf; f++) if (!seen[f]) return false;
        return true;
    }
    return false;
}

struct Wr {
    uint8_t *out;
    size_t cap, n = 0;
    void put(uint8_t b) { if (n < cap) out[n] = b; n++; }
    void be(uint64_t v, int k) { for (int j = k - 1; j >= 0; j--) put((uint8_t)(v >> (8 * j))); }
    void array(size_t c) { if (c < 16) put(0x90 | (uint8_t)c); else if (c < 65536) { put(0xdc); be(c, 2); } else { put(0xdd); be(c, 4); } }
    void map1() { put(0x81); }
    void str(const char *s) {
        size_t l = strlen(s);
        if (l < 32) put(0xa0 | (uint8_t)l); else if (l < 256) { put(0xd9); be(l, 1); } else if (l < 65536) { put(0xda); be(l, 2); } else { put(0xdb); be(l, 4); }
        for (size_t k = 0; k < l; k++) put((uint8_t)s[k]);
    }
    void uint(uint64_t v) {
        if (v < 128) put((uint8_t)v);
        else if (v < 256) { put(0xcc); be(v, 1); }
        else if (v < 65536) { put(0xcd); be(v, 2); }
        else if (v <= 0xffffffffull) { put(0xce); be(v, 4); }
        else { put(0xcf); be(v, 8); }
    }
    void nil() { put(0xc0); }
    // Vec<u8> as rmp-serde writes it: array of minimal-width unsigned ints
    void bytes_as_ints(const uint8_t *b, size_t l) {
        array(l);
        size_t hi_n = 0;                            // one byte each plus a 0xcc marker for values >= 128
        for (size_t k = 0; k < l; k++) hi_n += b[k] >> 7;
        if (!out || n + l + hi_n > cap) {           // sizing pass, or no room: count only (put() semantics)
            n += l + hi_n;
            return;
        }
        // branch-free (the high bytes of PCM samples are coin flips for a branch predictor); an element writes only its
        // own one or two bytes, so the exact-size buffer of the usual size-then-encode call pair is enough
        uint8_t *o = out + n;
        for (size_t k = 0; k < l; k++) { const uint8_t v = b[k]; const unsigned hi = v >> 7; o[0] = 0xcc; o[hi] = v; o += 1 + hi; }
        n = (size_t)(o - out);
    }
};

}  // namespace

extern "C" {

// The parsers allocate nothing proportional to the message (a hostile length prefix cannot make them reserve memory:
// code bytes go straight into the caller's buffer) and let no C++ exception out (variant names are short std::strings).
static int parse_request_impl(const uint8_t *msg, size_t len, uint8_t *codes_out, size_t codes_cap, size_t *codes_len,
                              uint64_t *n_tokens) {
    if (!msg) return HIPVOC_WIRE_E_MALFORMED;
    Rd r{msg, len};
    std::string name;
    bool payload;
    if (!rd_variant(r, name, payload)) return HIPVOC_WIRE_E_MALFORMED;
    if (!known_variant(name)) return HIPVOC_WIRE_E_MALFORMED;
    if (name == "Ping") {
        if (payload && !skip(r)) return HIPVOC_WIRE_E_MALFORMED;
        return r.i == len ? HIPVOC_WIRE_PING : HIPVOC_WIRE_E_MALFORMED;
    }
    if (!payload) return HIPVOC_WIRE_E_MALFORMED;
    if (name == "Vocode") {
        static const char *const f[] = {"codes", "n_tokens"};
        size_t clen = 0;
        uint64_t nt = 0;
        bool overflow = false;
        bool ok = rd_fields(r, f, 2, [&](size_t k) { return k == 0 ? rd_bytes_to(r, codes_out, codes_cap, clen, overflow) : rd_uint(r, nt); });
        if (!ok || r.bad || r.i != len) return HIPVOC_WIRE_E_MALFORMED;
        if (codes_len) *codes_len = clen;
        if (n_tokens) *n_tokens = nt;
        if (overflow) return HIPVOC_WIRE_E_CAP;
        return HIPVOC_WIRE_VOCODE;
    }
    if (!skip(r) || r.i != len) return HIPVOC_WIRE_E_MALFORMED;
    if (name == "Init") return HIPVOC_WIRE_INIT;
    if (name == "VocodeBatch") return HIPVOC_WIRE_VOCODE_BATCH;
    return HIPVOC_WIRE_OTHER;
}

int hipvoc_wire_parse_request(const uint8_t *msg, size_t len, uint8_t *codes_out, size_t codes_cap, size_t *codes_len,
                              uint64_t *n_tokens) {
    try {
        return parse_request_impl(msg, len, codes_out, codes_cap, codes_len, n_tokens);
    } catch (...) {
        return HIPVOC_WIRE_E_MALFORMED;
    }
}

static int parse_vocode_batch_impl(const uint8_t *msg, size_t len, uint8_t *codes_out, size_t codes_cap,
                                   size_t *offsets, uint64_t *n_tokens, size_t max_items, size_t *n_items) {
    if (!msg || !n_items) return HIPVOC_WIRE_E_MALFORMED;
    Rd r{msg, len};
    std::string name;
    bool payload;
    if (!rd_variant(r, name, payload) || !payload || name != "VocodeBatch") return HIPVOC_WIRE_E_MALFORMED;
    static const char *const f[] = {"codes", "n_tokens"};
    const bool query = !offsets && !codes_out && !n_tokens;         // size query: count the items only
    size_t n_codes = 0, n_nts = 0, off = 0;
    bool overflow = false, too_many = false;
    bool ok = rd_fields(r, f, 2, [&](size_t k) {
        size_t cnt;
        if (!rd_array(r, cnt) || cnt > r.n - r.i) return false;
        if (!query && cnt > max_items) too_many = true;             // keep parsing (validation), write nothing more
        if (k == 0) {
            n_codes = cnt;
            for (size_t j = 0; j < cnt; j++) {
                const bool wr = !query && !too_many;
                if (wr && offsets) offsets[j] = off;
                size_t got = 0;
                uint8_t *dst = wr && codes_out && !overflow ? codes_out + off : nullptr;
                if (!rd_bytes_to(r, dst, dst ? codes_cap - off : 0, got, overflow)) return false;
                if (wr && codes_out && !dst) overflow = true;
                off += got;
            }
            if (!query && !too_many && offsets) offsets[cnt] = off;
        } else {
            n_nts = cnt;
            for (size_t j = 0; j < cnt; j++) {
                uint64_t v;
                if (!rd_uint(r, v)) return false;
                if (!query && !too_many && n_tokens) n_tokens[j] = v;
            }
        }
        return true;
    });
    if (!ok || r.bad || r.i != len || n_codes != n_nts) return HIPVOC_WIRE_E_MALFORMED;
    *n_items = n_codes;
    if (query) return HIPVOC_WIRE_VOCODE_BATCH;
    if (too_many || overflow) return HIPVOC_WIRE_E_CAP;
    return HIPVOC_WIRE_VOCODE_BATCH;
}

int hipvoc_wire_parse_vocode_batch(const uint8_t *msg, size_t len, uint8_t *codes_out, size_t codes_cap,
                                   size_t *offsets, uint64_t *n_tokens, size_t max_items, size_t *n_items) {
    try {
        return parse_vocode_batch_impl(msg, len, codes_out, codes_cap, offsets, n_tokens, max_items, n_items);
    } catch (...) {
        return HIPVOC_WIRE_E_MALFORMED;
    }
}

size_t hipvoc_wire_encode_vocode(const int16_t *pcm, size_t n_samples, uint8_t *out, size_t cap) {
    Wr w{out, out ? cap : 0};
    w.array(3);
    w.str("ok");
    w.map1();
    w.str("Vocode");
    w.array(2);
    w.bytes_as_ints((const uint8_t *)pcm, n_samples * 2);     // i16 little-endian == in-memory bytes on this host
    w.uint(n_samples);
    w.nil();
    return w.n;
}

size_t hipvoc_wire_encode_vocode_batch(const int16_t *pcm, const size_t *so, size_t n_items, uint8_t *out, size_t cap) {
    Wr w{out, out ? cap : 0};
    w.array(3);
    w.str("ok");
    w.map1();
    w.str("VocodeBatch");
    w.array(2);
    w.array(n_items);
    for (size_t j = 0; j < n_items; j++) w.bytes_as_ints((const uint8_t *)(pcm + so[j]), (so[j + 1] - so[j]) * 2);
    w.array(n_items);
    for (size_t j = 0; j < n_items; j++) w.uint(so[j + 1] - so[j]);
    w.nil();
    return w.n;
}

size_t hipvoc_wire_encode_simple(int ok, const char *variant, const char *error, uint8_t *out, size_t cap) {
    Wr w{out, out ? cap : 0};
    w.array(3);
    w.str(ok ? "ok" : "error");
    w.str(variant ? variant : "Init");
    if (error) w.str(error); else w.nil();
    return w.n;
}

}  // extern "C"

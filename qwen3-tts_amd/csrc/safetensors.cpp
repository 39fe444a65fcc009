#include "safetensors.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstring>

namespace hipvoc {

namespace {

// Just enough JSON for a safetensors header: objects, arrays, strings, integers.
struct JParser {
    const char *p, *end;
    std::string err;
    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++;
    }
    bool lit(char c) {
        ws();
        if (p < end && *p == c) {
            p++;
            return true;
        }
        return false;
    }
    bool str(std::string &out) {
        ws();
        if (p >= end || *p != '"') return fail("expected string");
        p++;
        out.clear();
        while (p < end && *p != '"') {
            if (*p == '\\') {
                p++;
                if (p >= end) return fail("bad escape");
                char c = *p++;
                switch (c) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if (end - p < 4) return fail("bad \\u");
                        unsigned v = 0;
                        for (int i = 0; i < 4; i++) {
                            char h = p[i];
                            v = v * 16 + (h >= '0' && h <= '9' ? h - '0' : (h | 32) >= 'a' && (h | 32) <= 'f' ? (h | 32) - 'a' + 10 : 0);
                        }
                        p += 4;
                        out += v < 128 ? (char)v : '?';
                        break;
                    }
                    default: out += c;
                }
            } else {
                out += *p++;
            }
        }
        if (p >= end) return fail("unterminated string");
        p++;
        return true;
    }
    bool integer(int64_t &v) {
        ws();
        bool neg = false;
        if (p < end && *p == '-') {
            neg = true;
            p++;
        }
        if (p >= end || *p < '0' || *p > '9') return fail("expected integer");
        v = 0;
        while (p < end && *p >= '0' && *p <= '9') {
            if (v > (INT64_MAX - 9) / 10) return fail("integer out of range");
            v = v * 10 + (*p++ - '0');
        }
        if (neg) v = -v;
        return true;
    }
    bool int_array(std::vector<int64_t> &out) {
        out.clear();
        if (!lit('[')) return fail("expected [");
        if (lit(']')) return true;
        for (;;) {
            int64_t v;
            if (!integer(v)) return false;
            out.push_back(v);
            if (lit(',')) continue;
            if (lit(']')) return true;
            return fail("expected , or ]");
        }
    }
    bool skip_value(int depth = 0) {   // strings, numbers, objects, arrays, literals
        ws();
        if (depth > 32) return fail("nesting too deep");
        if (p >= end) return fail("eof");
        if (*p == '"') {
            std::string s;
            return str(s);
        }
        if (*p == '{' || *p == '[') {
            char open = *p, close = open == '{' ? '}' : ']';
            p++;
            if (lit(close)) return true;
            for (;;) {
                if (open == '{') {
                    std::string k;
                    if (!str(k) || !lit(':')) return fail("bad object");
                }
                if (!skip_value(depth + 1)) return false;
                if (lit(',')) continue;
                if (lit(close)) return true;
                return fail("bad container");
            }
        }
        while (p < end && *p != ',' && *p != '}' && *p != ']') p++;
        return true;
    }
    bool fail(const char *m) {
        if (err.empty()) err = m;
        return false;
    }
};

float half_to_float(uint16_t h) {
    uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023, u;
    if (e == 0) {
        if (m == 0) u = s << 31;
        else {
            int sh = 0;
            while (!(m & 1024)) { m <<= 1; sh++; }
            m &= 1023;
            u = (s << 31) | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31) u = (s << 31) | 0x7f800000u | (m << 13);
    else u = (s << 31) | ((e - 15 + 127) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

}  // namespace

SafeTensors::~SafeTensors() {
    if (map_) munmap(map_, map_len_);
    if (fd_ >= 0) close(fd_);
}

bool SafeTensors::open(const std::string &path, std::string &err) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) {
        err = "cannot open " + path;
        return false;
    }
    struct stat st;
    if (fstat(fd_, &st) != 0 || st.st_size < 8) {
        err = "cannot stat / too small: " + path;
        return false;
    }
    map_len_ = (size_t)st.st_size;
    map_ = mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (map_ == MAP_FAILED) {
        map_ = nullptr;
        err = "mmap failed: " + path;
        return false;
    }
    const uint8_t *base = (const uint8_t *)map_;
    uint64_t hlen = 0;
    for (int i = 0; i < 8; i++) hlen |= (uint64_t)base[i] << (8 * i);
    if (hlen > map_len_ - 8 || hlen > (1ull << 30)) {
        err = "bad safetensors header length";
        return false;
    }
    const uint8_t *data0 = base + 8 + hlen;
    size_t data_len = map_len_ - 8 - (size_t)hlen;
    JParser j{(const char *)base + 8, (const char *)base + 8 + hlen, {}};
    if (!j.lit('{')) {
        err = "safetensors header: expected {";
        return false;
    }
    if (!j.lit('}')) {
        for (;;) {
            std::string key;
            if (!j.str(key) || !j.lit(':')) break;
            if (key == "__metadata__") {
                if (!j.lit('{')) { j.fail("metadata"); break; }
                if (!j.lit('}')) {
                    for (;;) {
                        std::string k, v;
                        if (!j.str(k) || !j.lit(':')) break;
                        j.ws();
                        if (j.p < j.end && *j.p == '"') {
                            if (!j.str(v)) break;
                            metadata[k] = v;
                        } else if (!j.skip_value()) break;
                        if (j.lit(',')) continue;
                        if (j.lit('}')) break;
                        j.fail("metadata object");
                        break;
                    }
                }
            } else {
                StTensor t;
                std::vector<int64_t> offs;
                if (!j.lit('{')) { j.fail("tensor entry"); break; }
                for (;;) {
                    std::string k;
                    if (!j.str(k) || !j.lit(':')) break;
                    if (k == "dtype") { if (!j.str(t.dtype)) break; }
                    else if (k == "shape") { if (!j.int_array(t.shape)) break; }
                    else if (k == "data_offsets") { if (!j.int_array(offs)) break; }
                    else if (!j.skip_value()) break;
                    if (j.lit(',')) continue;
                    if (j.lit('}')) break;
                    j.fail("tensor object");
                    break;
                }
                if (!j.err.empty()) break;
                if (offs.size() != 2 || offs[0] < 0 || offs[1] < offs[0] || (size_t)offs[1] > data_len) {
                    err = "bad data_offsets for " + key;
                    return false;
                }
                t.data = data0 + offs[0];
                t.nbytes = (size_t)(offs[1] - offs[0]);
                size_t esz = t.dtype == "F32" ? 4 : (t.dtype == "F16" || t.dtype == "BF16") ? 2 : 0;
                // every dim fits an int and the element count cannot overflow, whatever the dtype (the loader casts
                // dims to int and sizes host vectors from numel())
                uint64_t numel = 1;
                for (auto d : t.shape) {
                    if (d < 0 || d > 0x7fffffffLL) { err = "dim out of range in " + key; return false; }
                    if (__builtin_mul_overflow(numel, (uint64_t)d, &numel) || numel > (1ull << 40)) { err = "tensor too large: " + key; return false; }
                }
                if (t.shape.size() > 8) { err = "too many dims in " + key; return false; }
                if (esz && numel * esz != t.nbytes) {
                    err = "size mismatch for " + key;
                    return false;
                }
                tensors_[key] = t;
            }
            if (!j.err.empty()) break;
            if (j.lit(',')) continue;
            if (j.lit('}')) break;
            j.fail("expected , or }");
            break;
        }
    }
    if (!j.err.empty()) {
        err = "safetensors header: " + j.err;
        return false;
    }
    return true;
}

const StTensor *SafeTensors::find(const std::string &name) const {
    auto it = tensors_.find(name);
    return it == tensors_.end() ? nullptr : &it->second;
}

bool SafeTensors::to_f32(const std::string &name, std::vector<float> &out, std::vector<int64_t> &shape,
                         std::string &err) const {
    const StTensor *t = find(name);
    if (!t) {
        err = "missing tensor " + name;
        return false;
    }
    if (t->dtype != "F32" && t->dtype != "BF16" && t->dtype != "F16") {      // before anything is sized from the header
        err = "unsupported dtype " + t->dtype + " for " + name;
        return false;
    }
    shape = t->shape;
    size_t n = (size_t)t->numel();
    out.resize(n);
    if (t->dtype == "F32") {
        memcpy(out.data(), t->data, n * 4);
    } else if (t->dtype == "BF16") {
        for (size_t i = 0; i < n; i++) {
            uint32_t u = ((uint32_t)t->data[2 * i] | ((uint32_t)t->data[2 * i + 1] << 8)) << 16;
            memcpy(&out[i], &u, 4);
        }
    } else if (t->dtype == "F16") {
        for (size_t i = 0; i < n; i++) out[i] = half_to_float((uint16_t)(t->data[2 * i] | (t->data[2 * i + 1] << 8)));
    } else {
        err = "unsupported dtype " + t->dtype + " for " + name;
        return false;
    }
    return true;
}

}  // namespace hipvoc

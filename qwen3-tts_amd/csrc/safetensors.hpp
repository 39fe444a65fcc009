// Minimal read-only safetensors container (8-byte LE header length, JSON header, raw LE data),
// memory-mapped.  Mirrors what the reference's VarBuilder::from_mmaped_safetensors call does for
// the encoder side (src/speech_tokenizer.rs:617-640); the decoder's tensors use the same container.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace hipvoc {

struct StTensor {
    std::string dtype;              // "F32", "F16", "BF16"
    std::vector<int64_t> shape;
    const uint8_t *data = nullptr;  // into the mapping
    size_t nbytes = 0;
    int64_t numel() const {
        int64_t n = 1;
        for (auto d : shape) n *= d;
        return n;
    }
};

class SafeTensors {
   public:
    SafeTensors() = default;
    ~SafeTensors();
    SafeTensors(const SafeTensors &) = delete;
    SafeTensors &operator=(const SafeTensors &) = delete;
    // returns false and fills err on failure
    bool open(const std::string &path, std::string &err);
    const StTensor *find(const std::string &name) const;
    bool has(const std::string &name) const { return find(name) != nullptr; }
    // tensor converted to fp32 (copy); false if missing / unsupported dtype
    bool to_f32(const std::string &name, std::vector<float> &out, std::vector<int64_t> &shape, std::string &err) const;
    std::map<std::string, std::string> metadata;

   private:
    std::map<std::string, StTensor> tensors_;
    void *map_ = nullptr;
    size_t map_len_ = 0;
    int fd_ = -1;
};

}  // namespace hipvoc

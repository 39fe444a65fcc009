// discovery.cpp -- which weights file the vocoder role loads from a models directory.
// Mirrors find_vocoder_model (/root/reference/src/worker/mod.rs:792-813): an ordered candidate list, first existing
// file wins, an error naming the directory otherwise.  The reference's candidates are vocoder.onnx / vocoder_64.onnx /
// vocoder_256.onnx (+ .rknn under its feature); the hip-vocoder backend reads safetensors (the upstream speech-tokenizer
// checkpoint's `decoder.*` tensors), so its list is the same names with that extension plus the checkpoint's own layout.
// `models_dir` may be the role directory (what download::ensure_models returns, download.rs:43-94: "<models>/vocoder") or
// its parent (what `worker -m <models>` is given, cli.rs:63-65).  Pure host code, no HIP.
#include "../../include/hipvoc.h"

#include <sys/stat.h>

#include <cstdio>
#include <cstring>
#include <string>

static bool is_file(const std::string &p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

extern "C" int hipvoc_find_vocoder_model(const char *models_dir, char *path_out, size_t cap) {
    try {
        if (!models_dir || !path_out || cap == 0) return HIPVOC_E_ARG;
        path_out[0] = 0;
        static const char *const names[] = {
            "vocoder.safetensors",    "vocoder_64.safetensors",        "vocoder_256.safetensors",
            "model.safetensors",      "speech_tokenizer/model.safetensors",
        };
        const std::string root(models_dir);
        const std::string dirs[] = {root, root + "/vocoder"};
        for (const std::string &d : dirs)
            for (const char *n : names) {
                std::string p = d + "/" + n;
                if (is_file(p)) {
                    if (p.size() + 1 > cap) return HIPVOC_E_CAP;
                    memcpy(path_out, p.c_str(), p.size() + 1);
                    return HIPVOC_OK;
                }
            }
        // nothing usable: say what IS there when it is the other backends' file
        std::string msg = "No vocoder model found in " + root + " (hip-vocoder wants vocoder.safetensors or speech_tokenizer/model.safetensors)";
        for (const std::string &d : dirs)
            for (const char *n : {"vocoder.onnx", "vocoder_64.onnx", "vocoder_256.onnx", "vocoder.rknn"})
                if (is_file(d + "/" + n)) msg += std::string("; ") + d + "/" + n + " is the ONNX/RKNN backends' file, not read by this backend";
        snprintf(path_out, cap, "%s", msg.c_str());      // the reason, in place of a path
        return HIPVOC_E_MODEL;
    } catch (...) {
        return HIPVOC_E_NOMEM;
    }
}

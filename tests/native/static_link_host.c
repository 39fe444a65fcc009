/* A plain-C host linked against libhipvoc.a exactly as integration/build_rs_fragment.rs tells cargo to link the reference's
 * worker: -lhipvoc (static), -lamdhip64, -lstdc++.  It proves two things the dlopen'ed .so cannot: that the archive's
 * members resolve from a C program with no C++ runtime of its own, and that the HIP fat binaries of every kernel object
 * register when they arrive through `ar` (each object carries its own module constructor; the linker keeps an archive
 * member only if something references it).
 *
 *   static_link_host probe <dir>                         no GPU needed: device count, model discovery, a failing load
 *   static_link_host decode <weights> <codes.bin> <n_frames> <unit_frames> <out.bin>
 *                                                        one hipvoc_decode (codes: n_frames x 16 little-endian i64)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hipvoc.h"

static int probe(const char *dir) {
    int n = hipvoc_device_count();
    printf("device_count %d\n", n);
    char path[4096];
    int rc = hipvoc_find_vocoder_model(dir, path, sizeof path);
    printf("find_vocoder_model rc=%d path=%s\n", rc, rc == 0 ? path : "-");
    void *h = hipvoc_load("/nonexistent/vocoder.safetensors", 0, HIPVOC_DTYPE_F32, 64);
    if (h) {
        printf("load of a missing file succeeded?\n");
        hipvoc_free(h);
        return 1;
    }
    const char *err = hipvoc_last_error(NULL);
    printf("load_error %s\n", err && err[0] ? err : "(empty)");
    return err && err[0] ? 0 : 1;
}

static int decode(const char *weights, const char *codes_path, int n_frames, int unit_frames, const char *out_path) {
    void *h = hipvoc_load(weights, 0, HIPVOC_DTYPE_F32, unit_frames);
    if (!h) {
        fprintf(stderr, "hipvoc_load: %s\n", hipvoc_last_error(NULL));
        return 2;
    }
    int nq = hipvoc_n_codebooks(h);
    size_t n_codes = (size_t)n_frames * (size_t)nq;
    int64_t *codes = (int64_t *)malloc(n_codes * sizeof(int64_t));
    FILE *f = fopen(codes_path, "rb");
    if (!codes || !f || fread(codes, sizeof(int64_t), n_codes, f) != n_codes) {
        fprintf(stderr, "cannot read %zu codes from %s\n", n_codes, codes_path);
        return 2;
    }
    fclose(f);
    size_t cap = (size_t)n_frames * 1920, n_out = 0;
    float *out = (float *)malloc(cap * sizeof(float));
    int rc = hipvoc_decode(h, codes, n_frames, out, cap, &n_out);
    if (rc) {
        fprintf(stderr, "hipvoc_decode rc=%d: %s\n", rc, hipvoc_last_error(h));
        return 3;
    }
    f = fopen(out_path, "wb");
    if (!f || fwrite(out, sizeof(float), n_out, f) != n_out) return 2;
    fclose(f);
    printf("decoded %zu samples\n", n_out);
    hipvoc_free(h);
    free(codes);
    free(out);
    return 0;
}

int main(int argc, char **argv) {
    if (argc == 3 && !strcmp(argv[1], "probe")) return probe(argv[2]);
    if (argc == 7 && !strcmp(argv[1], "decode")) return decode(argv[2], argv[3], atoi(argv[4]), atoi(argv[5]), argv[6]);
    fprintf(stderr, "usage: %s probe <dir> | decode <weights> <codes.bin> <n_frames> <unit_frames> <out.bin>\n", argv[0]);
    return 64;
}

// Accuracy of sin(t)^2 by v_fract_f32 + v_sin_f32 (hardware, input in revolutions) against the Cody-Waite polynomial of
// device_common.h and a float64 reference, over the argument range SnakeBeta sees (t = x e^alpha).
//   hipcc --offload-arch=gfx950 -O2 -I../../qwen3-tts_amd/csrc vsin_err.hip -o vsin_err && ./vsin_err
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "device_common.h"
using namespace hipvoc;
__global__ void k(const float *t, float *fast, float *poly, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float rev = t[i] * 0.15915494309189535f;
    float s = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rev));
    fast[i] = s * s;
    poly[i] = sin_sq(t[i]);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> t(n), f(n), p(n);
    unsigned s = 12345;
    for (int i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        float u = (s >> 8) * (1.0f / 16777216.0f);
        float range = i < n / 4 ? 1.0f : (i < n / 2 ? 10.0f : (i < 3 * n / 4 ? 100.0f : 1000.0f));
        t[i] = (2.0f * u - 1.0f) * range;
    }
    float *dt, *df, *dp;
    hipMalloc(&dt, n * 4); hipMalloc(&df, n * 4); hipMalloc(&dp, n * 4);
    hipMemcpy(dt, t.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dt, df, dp, n);
    hipMemcpy(f.data(), df, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(p.data(), dp, n * 4, hipMemcpyDeviceToHost);
    const char *names[4] = {"|t|<1", "|t|<10", "|t|<100", "|t|<1000"};
    for (int q = 0; q < 4; q++) {
        double ef = 0, ep = 0;
        for (int i = q * (n / 4); i < (q + 1) * (n / 4); i++) {
            double r = std::sin((double)t[i]); r *= r;
            ef = std::fmax(ef, std::fabs(f[i] - r));
            ep = std::fmax(ep, std::fabs(p[i] - r));
        }
        printf("%-9s max abs err of sin^2: v_sin_f32 %.3e   polynomial %.3e\n", names[q], ef, ep);
    }
    return 0;
}

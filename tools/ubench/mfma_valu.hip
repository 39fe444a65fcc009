// Microbenchmark: how many fp32 VALU instructions can ride in the shadow of v_mfma_f32_32x32x2_f32 with W waves per SIMD?
// Each wave: 8 independent accumulator tiles; per MFMA, NV VALU fmas on other registers.  No memory traffic in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, bool PK>
__global__ __launch_bounds__(512, 1) void k(float *out, int iters, float seed) {
    f32x16 acc[8];
    for (int q = 0; q < 8; q++)
        for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
    float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x;
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = seed * i + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[q], acc[q], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; j++) {
                if (PK) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 x = {v[(q + j) & 7], v[(q + j + 1) & 7]};
                    f2 y = {b, a};
                    x = __builtin_elementwise_fma(x, y, x);
                    v[(q + j) & 7] = x.x; v[(q + j + 1) & 7] = x.y;
                } else
                    v[(q + j + 1) & 7] = fmaf(v[(q + j + 1) & 7], b, v[(q + j + 2) & 7]);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
        }
    }
    float s = 0;
    for (int q = 0; q < 8; q++)
        for (int r = 0; r < 16; r++) s += acc[q][r];
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, bool PK>
void run(int threads, const char *name) {
    float *out;
    int blocks = 256 * 4;
    hipMalloc(&out, sizeof(float) * blocks * 512);
    int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, PK>), dim3(blocks), dim3(threads), 0, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, PK>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * (threads / 64) * iters * 8 * 4096.0;
    printf("%s NV=%d threads=%d: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", name, NV, threads, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
    hipFree(out);
}
int main() {
    for (int t : {256, 512, 1024}) {
        run<0, false>(t, "fma"); run<2, false>(t, "fma"); run<4, false>(t, "fma"); run<6, false>(t, "fma"); run<8, false>(t, "fma"); run<12, false>(t, "fma");
        run<2, true>(t, "pk "); run<4, true>(t, "pk ");
    }
    return 0;
}

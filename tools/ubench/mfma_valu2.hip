// Does a VALU instruction cost matrix-core time on gfx950 fp32?  Per MFMA (v_mfma_f32_32x32x2_f32), NV independent VALU ops of
// kind K on registers the MFMAs do not touch; NQ independent accumulator tiles; W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// K: 0 fma f32, 1 add u32, 2 cndmask, 3 16x16x4 mfma with fma f32
template <int NV, int K, int NQ, int NTH>
__global__ __launch_bounds__(NTH, 1) void k(float *out, int iters, float seed) {
    f32x16 acc[NQ];
    for (int q = 0; q < NQ; q++)
        for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
    float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x;
    float v[12];
    unsigned u[12];
    for (int i = 0; i < 12; i++) { v[i] = seed * i + threadIdx.x; u[i] = threadIdx.x * i; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; j++) {
                const int i = (q * NV + j) % 12;                 // 12 independent chains
                if (K == 0) v[i] = fmaf(v[i], b, a);
                if (K == 1) u[i] = u[i] + (unsigned)it;
                if (K == 2) v[i] = (u[i] >> (it & 31)) & 1 ? v[i] : a;
                if (K == 3) v[i] = __builtin_amdgcn_sinf(v[i]);
                if (K == 4) v[i] = __builtin_amdgcn_rcpf(v[i]);
                if (K == 5) v[i] = __builtin_amdgcn_exp2f(v[i]);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, K == 2 ? 3 * NV : NV, 0);
        }
    }
    float s = 0;
    for (int q = 0; q < NQ; q++)
        for (int r = 0; r < 16; r++) s += acc[q][r];
    for (int i = 0; i < 12; i++) s += v[i] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int K, int NQ, int NTH>
void run(const char *name) {
    float *out;
    int blocks = 256 * 4;
    hipMalloc(&out, sizeof(float) * blocks * NTH);
    int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, K, NQ, NTH>), dim3(blocks), dim3(NTH), 0, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, K, NQ, NTH>), dim3(blocks), dim3(NTH), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipError_t e = hipGetLastError();
    double nm = (double)blocks * (NTH / 64) * iters * NQ;          // MFMAs
    double cyc = ms * 1e-3 * 2.4e9 * 256 * 4;                        // SIMD-cycles available at 2.4 GHz
    printf("%s NV=%d NQ=%d waves/SIMD=%d: %.3f ms  %.1f%% of fp32 MFMA peak, %.1f SIMD-cycles per MFMA+%dVALU  (%s)\n", name, NV, NQ, NTH / 256, ms,
           nm * 4096 / ms / 1e9 / 157.3 * 100, cyc / nm, NV, hipGetErrorString(e));
    hipFree(out);
}
int main() {
    run<0, 0, 8, 256>("fma"); run<4, 0, 8, 256>("fma"); run<8, 0, 8, 256>("fma");
    run<0, 0, 8, 512>("fma"); run<2, 0, 8, 512>("fma"); run<4, 0, 8, 512>("fma"); run<8, 0, 8, 512>("fma");
    run<4, 1, 8, 512>("addu32"); run<8, 1, 8, 512>("addu32");
    run<4, 2, 8, 512>("shift+and+cndmask"); 
    run<2, 3, 8, 512>("v_sin"); run<4, 3, 8, 512>("v_sin"); run<8, 3, 8, 512>("v_sin"); run<4, 4, 8, 512>("v_rcp"); run<4, 5, 8, 512>("v_exp");
    run<0, 0, 4, 1024>("fma"); run<4, 0, 4, 1024>("fma"); run<8, 0, 4, 1024>("fma");
    return 0;
}

// Micro-benchmark: do MFMAs of one wave overlap with VALU / transcendental work of ANOTHER wave on the same SIMD?
// Workgroup = 512 threads (8 waves, two per SIMD), one workgroup per CU.  Waves 0-3 run role A, waves 4-7 role B.
// role 0 = idle, 1 = MFMA chain (32x32x16 bf16), 2 = v_fma chain, 3 = v_exp chain, 4 = interleaved 1 MFMA : NV valu (same wave)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int NV, int NE>
__device__ __forceinline__ void mixed(int iters, f32x16& acc, bf16x8 a, bf16x8 b, float* v) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < NV; ++e) v[e % 8] = fmaf(v[e % 8], 1.0001f, 0.5f);
#pragma unroll
            for (int e = 0; e < NE; ++e) v[(e + 4) % 8] = __builtin_amdgcn_exp2f(v[(e + 4) % 8]);
        }
    }
}

__global__ __launch_bounds__(512, 2) void k(int roleA, int roleB, int iters, float* out) {
    const int wid = threadIdx.x >> 6;
    const int role = __builtin_amdgcn_readfirstlane(wid < 4 ? roleA : roleB);
    f32x16 acc = f32x16{0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    if (role == 1) {
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    } else if (role == 2) {
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 48; ++u) v[u % 8] = fmaf(v[u % 8], 1.0001f, 0.5f);
    } else if (role == 3) {
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u % 8] = __builtin_amdgcn_exp2f(v[u % 8]);
    } else if (role == 4) mixed<4, 0>(iters, acc, a, b, v);
    else if (role == 5) mixed<6, 0>(iters, acc, a, b, v);
    else if (role == 6) mixed<4, 2>(iters, acc, a, b, v);
    else if (role == 7) mixed<2, 1>(iters, acc, a, b, v);
    else if (role == 8) mixed<0, 2>(iters, acc, a, b, v);
    else if (role == 9) mixed<8, 0>(iters, acc, a, b, v);
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    const int combos[][2] = {{1, 0}, {0, 2}, {1, 2}, {0, 3}, {1, 3}, {1, 1}, {2, 2}, {3, 3}, {4, 0}, {5, 0}, {9, 0}, {6, 0}, {7, 0}, {8, 0},
                             {4, 4}, {6, 6}, {7, 7}, {2, 3}};
    for (auto& c : combos) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, c[0], c[1], 10, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, c[0], c[1], iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // cycles per inner "unit" (8 MFMA / 48 fma / 16 exp) at an assumed 2.0 GHz
        printf("roleA=%d roleB=%d  %.3f ms  -> %.1f ns per iteration (8 MFMA | 48 fma | 16 exp | mixed 8x)\n", c[0], c[1], ms,
               ms * 1e6 / iters);
    }
    return 0;
}

// Micro-benchmark: matrix-pipe time per v_mfma_f32_32x32x16_bf16 when the SAME wave interleaves softmax-like VALU work,
// one workgroup of 256 threads (one wave per SIMD) x 4 workgroups per CU, all waves the same role.
//   per MFMA:  F scalar fma, A scalar add, E exp2, PF packed fma (2 lanes of work each), PA packed add
// Prints ns per 8 MFMAs; 8 x 32 cycles = 256 cycles is the pure-MFMA figure.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int F, int A, int E, int PF, int PA>
__global__ __launch_bounds__(256, 4) void k(int iters, float* out) {
    f32x16 acc = f32x16{0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    float v[8];
    f32x2 p[4];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 4; ++i) p[i] = f32x2{v[i], v[i + 4]};
    float col = 0.f;
    f32x2 pcol = f32x2{0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < F; ++e) v[(2 * u + e) % 8] = fmaf(v[(2 * u + e) % 8], 0.127f, -1.5f);
#pragma unroll
            for (int e = 0; e < E; ++e) v[(2 * u + e) % 8] = __builtin_amdgcn_exp2f(v[(2 * u + e) % 8]);
#pragma unroll
            for (int e = 0; e < A; ++e) col += v[(2 * u + e) % 8];
#pragma unroll
            for (int e = 0; e < PF; ++e) p[(u + e) % 4] = __builtin_elementwise_fma(p[(u + e) % 4], f32x2{0.127f, 0.127f}, f32x2{-1.5f, -1.5f});
#pragma unroll
            for (int e = 0; e < PA; ++e) pcol += p[(u + e) % 4];
        }
    }
    float s = col + pcol[0] + pcol[1];
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int F, int A, int E, int PF, int PA>
void run(const char* name, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    hipLaunchKernelGGL((k<F, A, E, PF, PA>), dim3(1024), dim3(256), 0, 0, 10, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<F, A, E, PF, PA>), dim3(1024), dim3(256), 0, 0, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // 1024 workgroups x 4 waves = 16 waves per CU = 4 per SIMD: each SIMD runs 4 x iters x 8 MFMAs
    printf("%-34s %8.3f ms   %.1f ns per 8 MFMAs per SIMD-slot (pure = 256 cycles)\n", name, ms, ms * 1e6 / iters / 4);
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    run<0, 0, 0, 0, 0>("mfma only", out);
    run<2, 2, 2, 0, 0>("2 fma + 2 add + 2 exp (pass 2)", out);
    run<0, 2, 2, 0, 0>("2 add + 2 exp", out);
    run<2, 0, 2, 0, 0>("2 fma + 2 exp", out);
    run<2, 2, 0, 0, 0>("2 fma + 2 add", out);
    run<0, 0, 2, 0, 0>("2 exp", out);
    run<0, 0, 2, 1, 1>("1 pk_fma + 1 pk_add + 2 exp", out);
    run<0, 2, 2, 1, 0>("1 pk_fma + 2 add + 2 exp", out);
    run<2, 0, 2, 0, 1>("2 fma + 1 pk_add + 2 exp", out);
    run<4, 0, 0, 0, 0>("4 fma", out);
    run<0, 0, 0, 2, 0>("2 pk_fma", out);
    return 0;
}

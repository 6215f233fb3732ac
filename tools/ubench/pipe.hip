// Micro-benchmark of the score kernels' inner structure on one CU-filling grid: what does one
// v_mfma_f32_32x32x16_bf16 cost per SIMD when each wave runs
//     [8 x NB MFMAs of a 32 x (32 NB) logit block, K = 128]  ->  [softmax-like VALU on the 16 NB results per lane]
// with W waves per SIMD, V VALU instructions per logit (0: none, 1: add, 2: exp + add, 3: fma + exp + add),
// the A fragments either constant registers (LDS = 0) or read from LDS with ds_read_b128 (LDS = 1: one read feeds NB
// MFMAs), and optionally a one-block software pipeline inside the wave (PIPE = 1: the VALU of block b is issued after
// the MFMAs of block b + 1, so it can overlap them in program order).
// Prints shader cycles (s_memtime) per MFMA per SIMD; 32 = the matrix pipe's own rate.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 pipe.hip -o pipe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = uint32_t __attribute__((ext_vector_type(4)));

template <int V>
__device__ __forceinline__ void softmax_block(float& col, const f32x16& acc, const float* ls, float c2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (V == 3) col += __builtin_amdgcn_exp2f(fmaf(acc[r], c2, -ls[r]));
        else if (V == 4) col += __builtin_amdgcn_exp2f(acc[r] * c2);                 // v_mul (normaliser folded into the accumulator)
        else if (V == 5) col += __builtin_amdgcn_exp2f(acc[r] - ls[r]);              // v_sub
        else if (V == 6) col += __builtin_amdgcn_exp2f(fmaf(acc[r], c2, ls[0]));     // fma with ONE addend register (pass 1 form)
        else if (V == 7) col += __builtin_amdgcn_exp2f(fmaf(acc[r], 0.127f, -1.5f)); // fma with literals
        else if (V == 8) col = fmaf(__builtin_amdgcn_exp2f(acc[r]), ls[r], col);     // exp, then fma into the sum (weights)
        else if (V == 2) col += __builtin_amdgcn_exp2f(acc[r]);
        else if (V == 1) col += acc[r];
    }
    if (V == 0) asm volatile("" ::"v"(acc));
}

template <int W, int NB, int V, int LDS, int PIPE>
__global__ __launch_bounds__(256, W) void k(int iters, float* out, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) char smem[16384 + 256];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384 / 4; i += 256) ((float*)smem)[i] = 0.001f * (i & 255);
    for (int i = tid; i < 64; i += 256) ((float*)(smem + 16384))[i] = 1.0f + 0.01f * i;
    __syncthreads();
    u32x4 kf[NB][8];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 8; ++r) kf[nb][r] = u32x4{0x3f803f80u + tid, 0x3f003f80u, 0x3e803f80u + r, 0x3f803e80u + nb};
    int frag_off[8];
    {
        const int row = lane & 31, hf = lane >> 5;
#pragma unroll
        for (int r = 0; r < 8; ++r) frag_off[r] = row * 256 + (((2 * r + hf) ^ (row & 15)) * 16);
    }
    u32x4 areg[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) areg[r] = u32x4{0x3f803f80u, 0x3f003f80u + lane, 0x3e803f80u, 0x3f803e80u + r};
    float col[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) col[nb] = 0.f;
    const float c2 = 0.12751743f;
    f32x16 pend[PIPE ? NB : 1];
    float pls[16];
    if (PIPE) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) pend[nb] = f32x16{0};
#pragma unroll
        for (int r = 0; r < 16; ++r) pls[r] = 1.f;
    }
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            u32x4 a[8];
            float ls[16];
            if (LDS) {
#pragma unroll
                for (int r = 0; r < 8; ++r) a[r] = *(const u32x4*)(smem + blk * 8192 + frag_off[r]);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) *(float4*)(ls + 4 * r4) = *(const float4*)(smem + 16384 + (blk * 32 + 8 * r4 + 4 * (lane >> 5)) * 4);
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) { a[r] = areg[r]; asm volatile("" : "+v"(a[r])); }
#pragma unroll
                for (int r = 0; r < 16; ++r) ls[r] = 1.0f + 0.01f * r;
            }
            f32x16 acc[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x16{0};
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[r]), __builtin_bit_cast(bf16x8, kf[nb][r]), acc[nb], 0, 0, 0);
            if constexpr (PIPE) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) softmax_block<V>(col[nb], pend[nb], pls, c2);
                if constexpr (PIPE == 2) {
                    // forced interleave: after each MFMA, the softmax instructions of 16 / 8 = 2 logits of the pending block
                    if (LDS) __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
                    for (int i = 0; i < 8 * NB; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) pend[nb] = acc[nb];
#pragma unroll
                for (int r = 0; r < 16; ++r) pls[r] = ls[r];
            } else {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) softmax_block<V>(col[nb], acc[nb], ls, c2);
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        s += col[nb];
        if (PIPE)
            for (int i = 0; i < 16; ++i) s += pend[nb][i];
    }
    if (s == 12345.678f) out[tid] = s;
    if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int W, int NB, int V, int LDS, int PIPE>
void run(float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * W;
    hipLaunchKernelGGL((k<W, NB, V, LDS, PIPE>), dim3(grid), dim3(256), 0, 0, 10, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<W, NB, V, LDS, PIPE>), dim3(grid), dim3(256), 0, 0, iters, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mfma_per_simd = (double)iters * 2 * 8 * NB * W;   // W waves per SIMD, each 16 NB MFMAs per iteration
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void*)k<W, NB, V, LDS, PIPE>);
    // the timed wave runs the whole kernel when the grid is one resident round: its cycles / wall = the shader clock
    printf("W=%d NB=%d V=%d LDS=%d PIPE=%d : %7.3f ns/MFMA/SIMD = %6.2f cycles at the measured %.2f GHz   (%d VGPRs, %zu B scratch)\n",
           W, NB, V, LDS, PIPE, ms * 1e6 / mfma_per_simd, (double)c / (mfma_per_simd / W) / W, (double)c / (ms * 1e6), fa.numRegs,
           (size_t)fa.localSizeBytes);
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 4096);
    hipMalloc(&cyc, 8);
    if (getenv("PIPE2")) {
        printf("-- forced interleave (PIPE=2) vs compiler (PIPE=1) vs unpipelined\n");
        run<3, 1, 3, 1, 0>(out, cyc);
        run<3, 1, 3, 1, 1>(out, cyc);
        run<3, 1, 3, 1, 2>(out, cyc);
        run<2, 1, 3, 1, 2>(out, cyc);
        run<1, 1, 3, 1, 2>(out, cyc);
        run<2, 2, 3, 1, 0>(out, cyc);
        run<2, 2, 3, 1, 1>(out, cyc);
        run<2, 2, 3, 1, 2>(out, cyc);
        run<1, 2, 3, 1, 2>(out, cyc);
        run<1, 2, 3, 0, 2>(out, cyc);
        run<1, 4, 3, 1, 2>(out, cyc);
        return 0;
    }
    printf("-- W=4 NB=1, LDS fragments, softmax forms\n");
    run<4, 1, 0, 0, 0>(out, cyc);
    run<4, 1, 1, 1, 0>(out, cyc);
    run<4, 1, 2, 1, 0>(out, cyc);
    run<4, 1, 3, 1, 0>(out, cyc);
    run<4, 1, 4, 1, 0>(out, cyc);
    run<4, 1, 5, 1, 0>(out, cyc);
    run<4, 1, 6, 1, 0>(out, cyc);
    run<4, 1, 7, 1, 0>(out, cyc);
    run<4, 1, 8, 1, 0>(out, cyc);
    printf("-- W=3 NB=1\n");
    run<3, 1, 2, 1, 0>(out, cyc);
    run<3, 1, 3, 1, 0>(out, cyc);
    run<3, 1, 4, 1, 0>(out, cyc);
    run<3, 1, 5, 1, 0>(out, cyc);
    run<3, 1, 6, 1, 0>(out, cyc);
    run<3, 1, 8, 1, 0>(out, cyc);
    printf("-- W=2 NB=2\n");
    run<2, 2, 2, 1, 0>(out, cyc);
    run<2, 2, 3, 1, 0>(out, cyc);
    run<2, 2, 4, 1, 0>(out, cyc);
    run<2, 2, 5, 1, 0>(out, cyc);
    run<2, 2, 6, 1, 0>(out, cyc);
    run<2, 2, 8, 1, 0>(out, cyc);
    printf("-- W=1 NB=4\n");
    run<1, 4, 2, 1, 0>(out, cyc);
    run<1, 4, 3, 1, 0>(out, cyc);
    run<1, 4, 4, 1, 0>(out, cyc);
    run<1, 4, 6, 1, 0>(out, cyc);
    return 0;
}

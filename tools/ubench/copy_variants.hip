// copy_variants.hip - which shape of a plain device copy reaches the most HBM bandwidth on MI355X?
// (calibration of bench.py's `hbm_achievable`: VERDICT r4 weak #5 - the library's copy kernel reported 4.98 TB/s while the
//  product's own gather_rows16_kernel moves 5.83 TB/s.)   hipcc --offload-arch=gfx950 -O3 -o copy_variants.bin copy_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// A: grid-strided, 4 vectors in flight per thread, each a whole grid apart (the round-4 rtk_profile_copy)
template <bool NT>
__global__ __launch_bounds__(256) void copy_gridstride(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        u32x4 a, b, c, e;
        if (NT) { a = __builtin_nontemporal_load(s + i); b = __builtin_nontemporal_load(s + i + stride); c = __builtin_nontemporal_load(s + i + 2 * stride); e = __builtin_nontemporal_load(s + i + 3 * stride); }
        else { a = s[i]; b = s[i + stride]; c = s[i + 2 * stride]; e = s[i + 3 * stride]; }
        if (NT) { __builtin_nontemporal_store(a, d + i); __builtin_nontemporal_store(b, d + i + stride); __builtin_nontemporal_store(c, d + i + 2 * stride); __builtin_nontemporal_store(e, d + i + 3 * stride); }
        else { d[i] = a; d[i + stride] = b; d[i + 2 * stride] = c; d[i + 3 * stride] = e; }
    }
    for (; i < n; i += stride) d[i] = s[i];
}

// B: wave-contiguous chunks: a wave copies U x 64 consecutive vectors (U KiB) per iteration, chunks grid-strided
template <int U, int NTL, int NTS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void copy_wavechunk(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n) {
    const int lane = threadIdx.x & 63;
    const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t chunk = (size_t)U * 64;
    for (size_t c = (((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * chunk; c < n; c += nw * chunk) {
        u32x4 buf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t v = c + (size_t)u * 64 + lane;
            if (v < n) buf[u] = NTL ? __builtin_nontemporal_load(s + v) : s[v];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t v = c + (size_t)u * 64 + lane;
            if (v < n) { if (NTS) __builtin_nontemporal_store(buf[u], d + v); else d[v] = buf[u]; }
        }
    }
}

// C: block-contiguous: a block copies U x BLOCK consecutive vectors per iteration
template <int U, int NTL, int NTS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void copy_blockchunk(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n) {
    const size_t chunk = (size_t)U * BLOCK;
    for (size_t c = (size_t)blockIdx.x * chunk; c < n; c += (size_t)gridDim.x * chunk) {
        u32x4 buf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t v = c + (size_t)u * BLOCK + threadIdx.x;
            if (v < n) buf[u] = NTL ? __builtin_nontemporal_load(s + v) : s[v];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t v = c + (size_t)u * BLOCK + threadIdx.x;
            if (v < n) { if (NTS) __builtin_nontemporal_store(buf[u], d + v); else d[v] = buf[u]; }
        }
    }
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : 2048) << 20;
    const size_t n = bytes / 16;
    u32x4 *s, *d;
    CK(hipMalloc(&s, bytes));
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(s, 1, bytes));
    CK(hipMemset(d, 2, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        const int reps = 5;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-58s %8.1f GB/s\n", name, 2.0 * bytes * reps / (ms * 1e-3) / 1e9);
    };
    for (unsigned grid : {2048u, 4096u, 8192u, 16384u}) {
        char nm[128];
        snprintf(nm, sizeof nm, "gridstride NT grid=%u", grid);
        timeit(nm, [&] { hipLaunchKernelGGL(copy_gridstride<true>, dim3(grid), dim3(256), 0, 0, s, d, n); });
        snprintf(nm, sizeof nm, "gridstride plain grid=%u", grid);
        timeit(nm, [&] { hipLaunchKernelGGL(copy_gridstride<false>, dim3(grid), dim3(256), 0, 0, s, d, n); });
    }
#define WV(U, NTL, NTS, BLOCK, GRID) do { char nm[128]; snprintf(nm, sizeof nm, "wavechunk U=%d ntl=%d nts=%d block=%d grid=%u", U, NTL, NTS, BLOCK, (unsigned)(GRID)); \
    timeit(nm, [&] { hipLaunchKernelGGL((copy_wavechunk<U, NTL, NTS, BLOCK>), dim3(GRID), dim3(BLOCK), 0, 0, s, d, n); }); } while (0)
#define BK(U, NTL, NTS, BLOCK, GRID) do { char nm[128]; snprintf(nm, sizeof nm, "blockchunk U=%d ntl=%d nts=%d block=%d grid=%u", U, NTL, NTS, BLOCK, (unsigned)(GRID)); \
    timeit(nm, [&] { hipLaunchKernelGGL((copy_blockchunk<U, NTL, NTS, BLOCK>), dim3(GRID), dim3(BLOCK), 0, 0, s, d, n); }); } while (0)
    for (unsigned grid : {2048u, 8192u, 32768u}) {
        WV(4, 0, 0, 256, grid); WV(4, 1, 1, 256, grid); WV(4, 0, 1, 256, grid); WV(4, 1, 0, 256, grid);
        WV(8, 0, 0, 256, grid); WV(8, 1, 1, 256, grid); WV(2, 0, 0, 256, grid); WV(1, 0, 0, 256, grid);
        BK(4, 0, 0, 256, grid); BK(4, 1, 1, 256, grid); BK(8, 0, 0, 256, grid); BK(4, 0, 0, 512, grid); BK(4, 0, 0, 1024, grid);
        BK(2, 0, 0, 256, grid); BK(1, 0, 0, 256, grid);
    }
    // one chunk per block (no loop): grid = n / chunk
    { const unsigned g = (unsigned)((n + 4 * 256 - 1) / (4 * 256)); BK(4, 0, 0, 256, g); BK(4, 1, 1, 256, g); }
    { const unsigned g = (unsigned)((n + 8 * 256 - 1) / (8 * 256)); BK(8, 0, 0, 256, g); }
    { const unsigned g = (unsigned)((n + 1 * 256 - 1) / (1 * 256)); BK(1, 0, 0, 256, g); }
    float ms;
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-58s %8.1f GB/s\n", "hipMemcpyAsync D2D", 2.0 * bytes * 5 / (ms * 1e-3) / 1e9);
    return 0;
}

/* A plain-C host on the C ABI (include/retake_hip.h): no Python, no torch.
 *
 * What a compiled host of the reference would do: device buffers from the HIP runtime, the hot path through the rtk_*
 * entry points, results copied back - checked here against the CPU oracle's C functions (oracle/retake_oracle.c, test
 * infrastructure; linked by this TEST program only).
 *   DPSelect   visual_compression.py:100-175   rtk_dpselect_dis -> rtk_dpselect_select -> rtk_gather_frames
 *   PivotKV    longvideo_cache.py:260-277      rtk_pivotkv_score (fp32, no reforge) -> rtk_pivotkv_select
 * Built and run by tests/test_c_host_gpu.py:
 *   gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c_host/hotpath_host.c -o hotpath_host \
 *       -Lvideo-retake_amd/retake/_lib -lretake_hip -Loracle/_build -lretake_oracle -L/opt/rocm/lib -lamdhip64 -lm
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "retake_hip.h"

/* the oracle's C entry points (oracle/retake_oracle.c) */
int orc_dpselect_dis_f32(const float* x, int T, int N, int C, float* dis);
int orc_dpselect_select(const float* dis, int T, int N, int tgt, int window, int sync, int64_t* idx, uint8_t* mask, float* keys_out);
int orc_gather_frames(const void* x, int esize, int T, int N, int C, const int64_t* idx, int t, int sync, void* out);
int orc_pivotkv_score(const float* q, const float* k, int Hq, int Hkv, int L, int D, float* score);
int orc_pivotkv_select(float* score, const uint8_t* mask, int L, int keep, int64_t* idx);

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define RTKCK(x) do { int r_ = (x); if (r_ != RTK_OK) { printf("rtk error %d (%s) at line %d\n", r_, rtk_last_error(), __LINE__); return 3; } } while (0)
#define ORCK(x) do { int r_ = (x); if (r_ != 0) { printf("oracle error %d at line %d\n", r_, __LINE__); return 4; } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float frand(void) {   /* uniform in (-1, 1), xorshift */
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (float)((double)(rng_state >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}
static float gauss(void) { float s = 0.f; for (int i = 0; i < 12; ++i) s += frand(); return s * 0.5f; }

static int dpselect_case(int sync) {
    const int T = 96, N = 12, C = 320, tgt = 40;
    const size_t nx = (size_t)T * N * C;
    float* x = (float*)malloc(nx * 4);
    /* video-like: AR(1) drift per patch position, so that the distance has real local maxima */
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) x[((size_t)0 * N + n) * C + c] = gauss();
    for (int t = 1; t < T; ++t)
        for (int n = 0; n < N; ++n) {
            const float rho = 0.55f + 0.43f * (0.5f * (frand() + 1.f));
            for (int c = 0; c < C; ++c)
                x[((size_t)t * N + n) * C + c] = rho * x[((size_t)(t - 1) * N + n) * C + c] + sqrtf(1.f - rho * rho) * gauss();
        }
    void *dx, *dout; float *ddis, *dkeys; int64_t* didx; uint8_t* dmask;
    const size_t nidx = sync ? (size_t)tgt : (size_t)tgt * N;
    HIPCK(hipMalloc(&dx, nx * 4));
    HIPCK(hipMalloc((void**)&ddis, (size_t)T * N * 4));
    HIPCK(hipMalloc((void**)&dkeys, (size_t)(sync ? 2 : N) * T * 4));
    HIPCK(hipMalloc((void**)&didx, nidx * 8));
    HIPCK(hipMalloc((void**)&dmask, (size_t)tgt * N));
    HIPCK(hipMalloc(&dout, (size_t)tgt * N * C * 4));
    HIPCK(hipMemcpy(dx, x, nx * 4, hipMemcpyHostToDevice));
    RTKCK(rtk_dpselect_dis(dx, T, N, C, RTK_F32, ddis, NULL));
    RTKCK(rtk_dpselect_select(ddis, T, N, tgt, 3, sync, didx, dmask, dkeys, NULL));
    RTKCK(rtk_gather_frames(dx, T, N, C, RTK_F32, didx, tgt, sync, dout, NULL));
    HIPCK(hipDeviceSynchronize());
    float* dis = (float*)malloc((size_t)T * N * 4); int64_t* idx = (int64_t*)malloc(nidx * 8);
    uint8_t* mask = (uint8_t*)malloc((size_t)tgt * N); float* out = (float*)malloc((size_t)tgt * N * C * 4);
    HIPCK(hipMemcpy(dis, ddis, (size_t)T * N * 4, hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(idx, didx, nidx * 8, hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(mask, dmask, (size_t)tgt * N, hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(out, dout, (size_t)tgt * N * C * 4, hipMemcpyDeviceToHost));
    /* oracle, from the DEVICE's distances for the selection (index parity is a statement about equal decision inputs;
       the distances themselves are compared to fp32 noise) */
    float* odis = (float*)malloc((size_t)T * N * 4); int64_t* oidx = (int64_t*)malloc(nidx * 8);
    uint8_t* omask = (uint8_t*)malloc((size_t)tgt * N); float* okeys = (float*)malloc((size_t)(sync ? 2 : N) * T * 4);
    float* oout = (float*)malloc((size_t)tgt * N * C * 4);
    ORCK(orc_dpselect_dis_f32(x, T, N, C, odis));
    double dmax = 0;
    for (size_t i = 0; i < (size_t)T * N; ++i) dmax = fmax(dmax, fabs((double)dis[i] - odis[i]));
    ORCK(orc_dpselect_select(dis, T, N, tgt, 3, sync, oidx, omask, okeys));
    ORCK(orc_gather_frames(x, 4, T, N, C, oidx, tgt, sync, oout));
    const int ok = dmax < 2e-6 && !memcmp(idx, oidx, nidx * 8) && !memcmp(mask, omask, (size_t)tgt * N) &&
                   !memcmp(out, oout, (size_t)tgt * N * C * 4);
    printf("DPSelect %s: T=%d N=%d C=%d tgt=%d: max |dis - oracle| %.2e, indices %s, mask %s, gathered frames %s\n", sync ? "sync" : "async",
           T, N, C, tgt, dmax, memcmp(idx, oidx, nidx * 8) ? "DIFFER" : "bit-exact", memcmp(mask, omask, (size_t)tgt * N) ? "DIFFER" : "bit-exact",
           memcmp(out, oout, (size_t)tgt * N * C * 4) ? "DIFFER" : "byte-identical");
    hipFree(dx); hipFree(ddis); hipFree(dkeys); hipFree(didx); hipFree(dmask); hipFree(dout);
    free(x); free(dis); free(idx); free(mask); free(out); free(odis); free(oidx); free(omask); free(okeys); free(oout);
    return ok ? 0 : 1;
}

static int pivotkv_case(void) {
    const int Hq = 28, Hkv = 4, L = 640, D = 128, keep = 160;
    const size_t nq = (size_t)Hq * L * D, nk = (size_t)Hkv * L * D;
    float* q = (float*)malloc(nq * 4); float* k = (float*)malloc(nk * 4); uint8_t* mask = (uint8_t*)malloc(L);
    for (size_t i = 0; i < nq; ++i) q[i] = 1.7f * gauss();
    for (size_t i = 0; i < nk; ++i) k[i] = 1.7f * gauss();
    for (int j = 0; j < L; ++j) mask[j] = frand() < -0.4f;   /* ~30 % key patches */
    void *dq, *dk, *dws, *dsel; float* dscore; uint8_t* dmask; int64_t* didx; int32_t* drank;
    const size_t wsb = rtk_pivotkv_score_workspace_bytes(Hq, Hkv, L, D, RTK_F32), selb = rtk_pivotkv_select_workspace_bytes(L);
    HIPCK(hipMalloc(&dq, nq * 4)); HIPCK(hipMalloc(&dk, nk * 4)); HIPCK(hipMalloc(&dws, wsb + 256)); HIPCK(hipMalloc(&dsel, selb));
    HIPCK(hipMalloc((void**)&dscore, (size_t)L * 4)); HIPCK(hipMalloc((void**)&dmask, L)); HIPCK(hipMalloc((void**)&didx, (size_t)keep * 8));
    HIPCK(hipMalloc((void**)&drank, (size_t)L * 4));
    HIPCK(hipMemcpy(dq, q, nq * 4, hipMemcpyHostToDevice)); HIPCK(hipMemcpy(dk, k, nk * 4, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(dmask, mask, L, hipMemcpyHostToDevice));
    void* ws = (void*)(((uintptr_t)dws + 255) & ~(uintptr_t)255);
    RTKCK(rtk_pivotkv_score(dq, (int64_t)L * D, D, dk, (int64_t)L * D, D, Hq, Hkv, L, D, RTK_F32, NULL, NULL, 1.0f, dscore, NULL, ws, wsb, NULL));
    float* score = (float*)malloc((size_t)L * 4);
    HIPCK(hipMemcpy(score, dscore, (size_t)L * 4, hipMemcpyDeviceToHost));
    RTKCK(rtk_pivotkv_select(dscore, dmask, L, keep, NULL, 0, 0, didx, drank, NULL, keep, dsel, selb, NULL));
    HIPCK(hipDeviceSynchronize());
    int64_t* idx = (int64_t*)malloc((size_t)keep * 8);
    HIPCK(hipMemcpy(idx, didx, (size_t)keep * 8, hipMemcpyDeviceToHost));
    float* oscore = (float*)malloc((size_t)L * 4); int64_t* oidx = (int64_t*)malloc((size_t)keep * 8);
    ORCK(orc_pivotkv_score(q, k, Hq, Hkv, L, D, oscore));
    double smax = 0;
    for (int j = 0; j < L; ++j) smax = fmax(smax, fabs((double)score[j] - oscore[j]));
    /* margin of the oracle's own k-th boundary after the mask override */
    float* tmp = (float*)malloc((size_t)L * 4); memcpy(tmp, oscore, (size_t)L * 4);
    ORCK(orc_pivotkv_select(tmp, mask, L, keep, oidx));
    float kth = 1e30f, next = -1e30f; char* in = (char*)calloc(L, 1);
    for (int r = 0; r < keep; ++r) { in[oidx[r]] = 1; if (tmp[oidx[r]] < kth) kth = tmp[oidx[r]]; }
    for (int j = 0; j < L; ++j) if (!in[j] && tmp[j] > next) next = tmp[j];
    const int same = !memcmp(idx, oidx, (size_t)keep * 8);
    const int ok = smax < 5e-6 && (same || kth - next < 2e-5);
    printf("PivotKV fp32: Hq=%d Hkv=%d L=%d D=%d keep=%d: max |score - oracle| %.2e, kept indices %s (oracle margin %.2e)\n", Hq, Hkv, L, D,
           keep, smax, same ? "bit-exact" : "differ inside fp32 noise of the boundary", (double)(kth - next));
    hipFree(dq); hipFree(dk); hipFree(dws); hipFree(dsel); hipFree(dscore); hipFree(dmask); hipFree(didx); hipFree(drank);
    free(q); free(k); free(mask); free(score); free(idx); free(oscore); free(oidx); free(tmp); free(in);
    return ok ? 0 : 1;
}

int main(void) {
    printf("libretake_hip ABI %d for %s\n", rtk_version(), rtk_arch());
    int rc = 0;
    rc |= dpselect_case(0);
    rc |= dpselect_case(1);
    rc |= pivotkv_case();
    /* error behaviour across the ABI: a status code and a message, no exception */
    if (rtk_dpselect_dis(NULL, 4, 4, 4, RTK_F32, NULL, NULL) != RTK_EINVAL || !strlen(rtk_last_error())) rc |= 1;
    printf(rc ? "C_HOST_FAILED\n" : "C_HOST_OK\n");
    return rc;
}

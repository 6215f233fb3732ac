// api.hip — ABI bookkeeping: version, architecture, thread-local error string.
#include <algorithm>
#include <cstdarg>
#include <cstdio>

#include "common.cuh"

namespace rtk {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
    set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
    return RTK_EHIP;
}
}  // namespace rtk

// ---- per-kernel event timing ------------------------------------------------------------------------
#include <atomic>
#include <mutex>
#include <vector>
namespace rtk {
static const char* const kKernelNames[KID_COUNT] = {"dpselect_dis", "dpselect_select", "gather_frames", "rope_table",
                                                    "unrotate_pack", "score_pass1", "score_pass2", "score_finalize",
                                                    "pivotkv_select", "evict_scan", "copy_rows", "append",
                                                    "evict_batched", "commit_batched", "position_shift", "pivotkv_emit",
                                                    "prologue", "compact_units"};
struct ProfRec { int kid; hipEvent_t a, b; };
static std::mutex g_pm;
static std::atomic<unsigned> g_prof{0};  // bit k set = time kernel id k
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static double g_total_ms[KID_COUNT];
static long long g_count[KID_COUNT];
static thread_local hipEvent_t g_open = nullptr;

bool profile_on(int kid) { return (g_prof.load(std::memory_order_relaxed) >> kid) & 1u; }
static hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void profile_begin(int, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_pm);
    g_open = get_event();
    (void)hipEventRecord(g_open, st);
}
void profile_end(int kid, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_pm);
    hipEvent_t b = get_event();
    (void)hipEventRecord(b, st);
    g_recs.push_back(ProfRec{kid, g_open, b});
    g_open = nullptr;
}
}  // namespace rtk

extern "C" int rtk_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(rtk::g_pm);
    rtk::g_prof = on ? 0xffffffffu : 0u;
    return RTK_OK;
}
// Time only the kernels whose id bit is set (ids = index into rtk_profile_kernel_name); 0 disables.
extern "C" int rtk_profile_enable_mask(unsigned mask) {
    std::lock_guard<std::mutex> lk(rtk::g_pm);
    rtk::g_prof = mask;
    return RTK_OK;
}
// Waits for the recorded events, folds them into per-kernel totals and recycles them.
extern "C" int rtk_profile_collect(void) {
    std::lock_guard<std::mutex> lk(rtk::g_pm);
    for (auto& r : rtk::g_recs) {
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) return rtk::hip_fail(e, "rtk_profile_collect");
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, r.a, r.b);
        if (e != hipSuccess) return rtk::hip_fail(e, "rtk_profile_collect");
        rtk::g_total_ms[r.kid] += ms;
        rtk::g_count[r.kid] += 1;
        rtk::g_pool.push_back(r.a);
        rtk::g_pool.push_back(r.b);
    }
    rtk::g_recs.clear();
    return RTK_OK;
}
extern "C" int rtk_profile_reset(void) {
    int rc = rtk_profile_collect();
    std::lock_guard<std::mutex> lk(rtk::g_pm);
    for (int i = 0; i < rtk::KID_COUNT; ++i) { rtk::g_total_ms[i] = 0; rtk::g_count[i] = 0; }
    return rc;
}
extern "C" int rtk_profile_num_kernels(void) { return rtk::KID_COUNT; }
extern "C" const char* rtk_profile_kernel_name(int kid) {
    return (kid >= 0 && kid < rtk::KID_COUNT) ? rtk::kKernelNames[kid] : "";
}
extern "C" int rtk_profile_read(int kid, long long* count, double* total_ms) {
    if (kid < 0 || kid >= rtk::KID_COUNT || !count || !total_ms) return RTK_EINVAL;
    std::lock_guard<std::mutex> lk(rtk::g_pm);
    *count = rtk::g_count[kid];
    *total_ms = rtk::g_total_ms[kid];
    return RTK_OK;
}

// ---- calibration copy for the HBM rooflines ------------------------------------------------------------
namespace rtk {
// dst[i] = src[i] over 16-byte vectors, non-temporal on both sides (no L2 / MALL residue between repetitions).  One
// workgroup = ONE contiguous 16 KiB piece (4 vectors per thread, a wave's four loads 1 KiB apart), no loop: measured
// 6.26 TB/s on a 2 GiB buffer (tools/ubench/copy_variants.hip, profiles/r14_copy_variants.txt) - the guide's 6.29 TB/s
// "float4 copy".  The round-4 form (grid-strided, a thread's four vectors a whole grid = 32 MiB apart) reached 4.9-5.2.
__global__ __launch_bounds__(256) void copy_nt_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
    const size_t c = (size_t)blockIdx.x * (4 * 256) + threadIdx.x;
    u32x4 buf[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (c + (size_t)u * 256 < n) buf[u] = __builtin_nontemporal_load(src + c + (size_t)u * 256);
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (c + (size_t)u * 256 < n) __builtin_nontemporal_store(buf[u], dst + c + (size_t)u * 256);
}
}  // namespace rtk

extern "C" int rtk_profile_copy(void* dst, const void* src, size_t bytes, rtk_stream_t stream) {
    RTK_CHECK_ARG(dst && src && bytes % 16 == 0 && (((uintptr_t)dst | (uintptr_t)src) & 15) == 0,
                  "rtk_profile_copy: 16-byte aligned buffers of a multiple of 16 bytes");
    if (bytes == 0) return RTK_OK;
    const size_t n = bytes / 16;
    RTK_CHECK_ARG((n + 4 * 256 - 1) / (4 * 256) <= 0x7fffffffu, "rtk_profile_copy: buffer too large for one launch");
    const unsigned grid = (unsigned)((n + 4 * 256 - 1) / (4 * 256));
    hipLaunchKernelGGL(rtk::copy_nt_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const rtk::u32x4*)src,
                       (rtk::u32x4*)dst, n);
    RTK_LAUNCH_CHECK("copy_nt_kernel");
    return RTK_OK;
}

extern "C" int rtk_version(void) { return 17; }
extern "C" const char* rtk_last_error(void) { return rtk::g_err; }
extern "C" const char* rtk_arch(void) { return "gfx950"; }

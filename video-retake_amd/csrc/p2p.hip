// p2p.hip — direct peer-to-peer all-gather over xGMI (SURVEY §5 "Distributed backend", §8(b) rtk_allgather_*).
// Every rank maps its peers' receive buffers (hipIpc) and PUSHES its block into all of them with plain 16-byte
// stores: one hop on every peer link at once instead of the world-1 serial steps of a ring, which is what the
// 200 KB - 26 MB payloads of this path (distance rows, id offsets, a chunk's kept rows) are bound by.  Arrival is
// signalled by a per-sender epoch word in the receiver's uncached flag array; waiting is a separate launch so that
// compute can sit between the push and the wait.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.cuh"

namespace rtk {

// Segment s (seg_bytes, a multiple of 16) of the source goes to byte dst_off + s * dst_stride of EVERY peer's buffer
// (blockIdx.y = peer, own rank included).  Data only: every wave fences its stores to system scope before it ends; the
// arrival flags are published by p2p_flag_kernel, a second launch behind this one on the same stream.
__global__ __launch_bounds__(256) void p2p_push_kernel(const char* __restrict__ src, size_t seg_bytes, int nseg,
                                                       size_t src_stride, rtk_p2p_peers peers, size_t dst_off,
                                                       size_t dst_stride) {
    char* dst = (char*)peers.buf[blockIdx.y] + dst_off;
    const size_t vec_per_seg = seg_bytes / 16, total = vec_per_seg * (size_t)nseg;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t s = i / vec_per_seg, v = i - s * vec_per_seg;
        *(uint4*)(dst + s * dst_stride + v * 16) = *(const uint4*)(src + s * src_stride + v * 16);
    }
    __threadfence_system();   // this wave's rows are visible system-wide before the wave ends
}

// The arrival flags of a push, as their own launch behind the data launch: a kernel boundary orders them behind EVERY
// workgroup of the copy (the first form counted finished workgroups on a device word inside the copy kernel and let the
// last one publish - one premature "last" and the receivers read rows that had not landed).
__global__ void p2p_flag_kernel(rtk_p2p_peers peers, int rank, int world, uint32_t epoch) {
    const int q = threadIdx.x;
    if (q >= world) return;
    __threadfence_system();
    __hip_atomic_store(peers.flag[q] + rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Thread r waits until sender r has published an epoch >= `epoch` (wrap-safe compare).  Bounded: after max_spins
// polls the thread gives up and records 1 + r in *status, so a lost peer shows up as an error instead of a hung GPU.
__global__ void p2p_wait_kernel(const uint32_t* flags, int world, uint32_t epoch, unsigned max_spins, uint32_t* status) {
    const int r = threadIdx.x;
    if (r >= world) return;
    unsigned spins = 0;
    while ((int32_t)(__hip_atomic_load(flags + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
        __builtin_amdgcn_s_sleep(64);
        if (++spins > max_spins) {
            atomicExch(status, 1u + (unsigned)r);
            return;
        }
    }
}

}  // namespace rtk

using namespace rtk;

extern "C" int rtk_p2p_alloc(size_t bytes, int uncached, void** ptr) {
    RTK_CHECK_ARG(ptr && bytes > 0, "rtk_p2p_alloc: null pointer or zero size");
    // Every buffer peers map is its own allocation of whole 2 MiB granules: the runtime sub-allocates smaller requests from
    // shared 2 MiB blocks, and exporting such a fragment (hipIpcGetMemHandle) hands the peers a mapping of the WHOLE block -
    // i.e. of whatever else of this process lives in it (without torch's caching allocator: its small tensors).
    const size_t granule = (size_t)2 << 20;
    // (RETAKE_P2P_EXACT_ALLOC=1, debugging aid: the request as it is - the behaviour before round 6, for A/B runs)
    const char* exact = getenv("RETAKE_P2P_EXACT_ALLOC");
    const size_t rounded = (exact && exact[0] == '1') ? bytes : (bytes + granule - 1) / granule * granule;
    hipError_t e = uncached ? hipExtMallocWithFlags(ptr, rounded, hipDeviceMallocUncached) : hipMalloc(ptr, rounded);
    if (e != hipSuccess) return hip_fail(e, "rtk_p2p_alloc");
    bytes = uncached ? rounded : bytes;   // (the flag words of the whole granule start at zero)
    // flag words start at zero; a landing buffer is only ever read where a push has landed, and zeroing it would leave
    // lines of it in this device's caches for peers' rows to race with
    if (!uncached) return RTK_OK;
    e = hipMemset(*ptr, 0, bytes);
    if (e != hipSuccess) return hip_fail(e, "rtk_p2p_alloc: memset");
    // the fill is asynchronous; peers may write as soon as they hold the handle, and nothing orders THEIR kernels behind it
    e = hipDeviceSynchronize();
    if (e != hipSuccess) return hip_fail(e, "rtk_p2p_alloc: synchronize");
    return RTK_OK;
}

extern "C" int rtk_p2p_free(void* ptr) {
    const hipError_t e = hipFree(ptr);
    return e == hipSuccess ? RTK_OK : hip_fail(e, "rtk_p2p_free");
}

extern "C" int rtk_p2p_export(const void* ptr, void* handle_out, size_t* offset_out) {
    RTK_CHECK_ARG(ptr && handle_out && offset_out, "rtk_p2p_export: null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == RTK_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    hipError_t e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr);
    if (e != hipSuccess) return hip_fail(e, "rtk_p2p_export: hipMemGetAddressRange");
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, base);
    if (e != hipSuccess) return hip_fail(e, "rtk_p2p_export: hipIpcGetMemHandle");
    std::memcpy(handle_out, &h, sizeof(h));
    *offset_out = (size_t)((const char*)ptr - (const char*)base);
    return RTK_OK;
}

extern "C" int rtk_p2p_open(const void* handle, void** base_out) {
    RTK_CHECK_ARG(handle && base_out, "rtk_p2p_open: null argument");
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof(h));
    const hipError_t e = hipIpcOpenMemHandle(base_out, h, hipIpcMemLazyEnablePeerAccess);
    return e == hipSuccess ? RTK_OK : hip_fail(e, "rtk_p2p_open: hipIpcOpenMemHandle");
}

extern "C" int rtk_p2p_close(void* base) {
    const hipError_t e = hipIpcCloseMemHandle(base);
    return e == hipSuccess ? RTK_OK : hip_fail(e, "rtk_p2p_close");
}

extern "C" int rtk_p2p_push(const void* src, size_t seg_bytes, int nseg, size_t src_stride_bytes,
                            const rtk_p2p_peers* peers, int rank, int world, size_t dst_offset_bytes,
                            size_t dst_stride_bytes, uint32_t epoch, rtk_stream_t stream) {
    RTK_CHECK_ARG(peers, "rtk_p2p_push: null argument");
    RTK_CHECK_ARG(world >= 1 && world <= RTK_P2P_MAX_RANKS && rank >= 0 && rank < world,
                  "rtk_p2p_push: rank %d / world %d outside [1, %d]", rank, world, RTK_P2P_MAX_RANKS);
    RTK_CHECK_ARG(nseg >= 0 && seg_bytes % 16 == 0 && src_stride_bytes % 16 == 0 && dst_stride_bytes % 16 == 0 &&
                      dst_offset_bytes % 16 == 0 && ((uintptr_t)src % 16 == 0 || nseg == 0 || seg_bytes == 0),
                  "rtk_p2p_push: sizes, strides, offsets and the source must be multiples of 16 bytes");
    for (int q = 0; q < world; ++q) RTK_CHECK_ARG(peers->buf[q] && peers->flag[q], "rtk_p2p_push: peer %d not mapped", q);
    const size_t vecs = seg_bytes / 16 * (size_t)nseg;
    // enough workgroups per peer to keep a link busy, few enough that world of them share the chip
    const unsigned per_peer = (unsigned)std::max<size_t>(1, std::min<size_t>((vecs + 1023) / 1024, 2048 / (unsigned)world));
    if (vecs) {
        hipLaunchKernelGGL(p2p_push_kernel, dim3(per_peer, world), dim3(256), 0, (hipStream_t)stream, (const char*)src,
                           seg_bytes, nseg, src_stride_bytes, *peers, dst_offset_bytes, dst_stride_bytes);
        RTK_LAUNCH_CHECK("rtk_p2p_push");
    }
    hipLaunchKernelGGL(p2p_flag_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *peers, rank, world, epoch);
    RTK_LAUNCH_CHECK("rtk_p2p_push: flags");
    return RTK_OK;
}

extern "C" int rtk_p2p_wait(const uint32_t* own_flags, int world, uint32_t epoch, int timeout_ms, uint32_t* status,
                            rtk_stream_t stream) {
    RTK_CHECK_ARG(own_flags && status, "rtk_p2p_wait: null argument");
    RTK_CHECK_ARG(world >= 1 && world <= RTK_P2P_MAX_RANKS, "rtk_p2p_wait: world %d outside [1, %d]", world,
                  RTK_P2P_MAX_RANKS);
    // one poll = an uncached load + s_sleep 64 (64 x 64 cycles): about 3 us
    const unsigned max_spins = (unsigned)std::min<long long>(0x7fffffffLL, (long long)std::max(1, timeout_ms) * 333);
    hipLaunchKernelGGL(p2p_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, own_flags, world, epoch, max_spins,
                       status);
    RTK_LAUNCH_CHECK("rtk_p2p_wait");
    return RTK_OK;
}

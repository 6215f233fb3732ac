// pivotkv_compact.hip - the eviction scan of a chunk's (layer, chunk) units in ONE launch, in place.
// Replaces longvideo_cache.py:278-288 (the three gathers), :297-306 (re-rotation of the kept keys at their new ids),
// :308-310 (position-cache bookkeeping) and :313-318 (the cache rebuild) - what rtk_pivotkv_evict_batched[_rope] +
// rtk_pivotkv_place_batched do in two launches with a staging hop for the rows whose source lies inside the
// destination range.
//
// Roofline: HBM.  Per unit the kernel moves what the compaction has to move and nothing else:
//   K  keep x Hkv x D x s read (the un-rotated rows, a buffer of their own) + as much written (rotated, to the tail)
//   V  keep x Hkv x D x s read + as much written, inside the same tail
//   = 4 x keep x Hkv x D x s  (6.4 MB at keep 1568, Hkv 4, D 128, bf16; 7.3 MB with the staging hop).
//
// In-place order.  Kept row r of a tail comes from chunk row keep_idx[r] >= r of the SAME tail (keep_idx ascending),
// so a destination row may still be somebody's source.  A workgroup owns R consecutive kept rows of one group of KV
// heads; it (1) takes a ticket - its row block is the ticket, so a lower block has always started, (2) issues every
// load of its rows, (3) raises its flag once the loads have landed in registers, (4) waits for the flags of ALL lower
// blocks of its (unit, head group) - the readers of its destination rows have indices <= its own, (5) stores;
// the last block to finish puts tickets and flags back to zero.
// No workgroup waits on a higher ticket or on anything a waiting workgroup holds: no deadlock at any occupancy.
//
// Rotation.  cos / sin of a kept row's new ids are computed per thread (8 correctly rounded sincos, shared by the KV
// heads of the block) while the row loads are in flight; the rotation rounds through v_cvt_pk_bf16_f32 - one
// instruction per pair instead of the integer sequence of rtk_pivotkv_evict_batched_rope, which was VALU-bound.
#include "common.cuh"
#include "variants.h"

namespace rtk {

struct CompactUnits {
    rtk_compact_unit u[RTK_COMPACT_MAX_UNITS];
};

constexpr int CMP_BLOCK = 256;
constexpr int CMP_HU = RTK_CMP_HU;            // KV heads per workgroup
constexpr int CMP_HDR = 32;          // ints before the flags of a (unit, head group): [0] ticket, [1] finished blocks

template <int DT, int KMODE>
__global__ __launch_bounds__(CMP_BLOCK, RTK_CMP_WAVES) void compact_units_kernel(CompactUnits units, int Hkv, int HG, int D, int keep,
                                                                  int P, const float* __restrict__ inv_freq,
                                                                  float scaling, RowSel rs, int round_mode,
                                                                  int32_t* __restrict__ sync,
                                                                  int sync_stride) {
    using V = Vec16<DT>;
    constexpr int VE = V::VE;
    constexpr int ES = 16 / VE;
    constexpr int HU = CMP_HU;
    __shared__ int s_b;
    const int tid = threadIdx.x;
    const int y = blockIdx.y, unit = y / HG, hg = y - unit * HG;
    const rtk_compact_unit& un = units.u[unit];
    int32_t* sy = sync + (size_t)y * sync_stride;
    if (tid == 0) s_b = atomicAdd(&sy[0], 1);
    __syncthreads();
    const int b = s_b;                      // row block = ticket: every lower block is already running
    const int nb = gridDim.x;
    const int h2 = D / 2, lpr = h2 / VE, R = CMP_BLOCK / lpr;
    const int rl = tid / lpr, c = tid - rl * lpr, d = c * VE;
    const int r = b * R + rl;
    const bool active = rl < R && r < keep;
    const int rc = min(r, keep - 1);
    const int64_t* kip = un.keep_idx + rc;
    long long id[3] = {0, 0, 0};
    if (KMODE == 0) {   // requested before the row loads that depend on keep_idx: the table is built while those fly
#pragma unroll
        for (int p = 0; p < 3; ++p) id[p] = un.pos_src[(size_t)min(p, P - 1) * un.pos_src_stride + rc];
    }
    const int64_t l = *kip;
    const int h0 = hg * HU, nh = min(HU, Hkv - h0);
    char* kt = (char*)un.k_tail;
    char* vt = (char*)un.v_tail;
    const char* ks = KMODE == 2 ? (const char*)un.k_tail : (const char*)un.k_src;
    const int64_t ks_sh = KMODE == 2 ? un.k_tail_stride_h : un.k_src_stride_h;
    // every load of the block first: V (and K) rows of the tails, then what the rotation needs
    u32x4 v_lo[HU], v_hi[HU], k_lo[HU], k_hi[HU];
#pragma unroll
    for (int u = 0; u < HU; ++u) {
        const int h = min(h0 + u, Hkv - 1);
        const char* vr = vt + ((size_t)h * un.v_tail_stride_h + (size_t)l * D) * ES;
        v_lo[u] = *(const u32x4*)(vr + (size_t)d * ES);
        v_hi[u] = *(const u32x4*)(vr + (size_t)(d + h2) * ES);
    }
#pragma unroll
    for (int u = 0; u < HU; ++u) {
        const int h = min(h0 + u, Hkv - 1);
        const char* kr = ks + ((size_t)h * ks_sh + (size_t)l * D) * ES;
        k_lo[u] = *(const u32x4*)(kr + (size_t)d * ES);
        k_hi[u] = *(const u32x4*)(kr + (size_t)(d + h2) * ES);
    }
    // every row this block reads from the tails is in registers: raise the flag the higher blocks wait for - before
    // this block's own stores are even issued (vmcnt counts stores too on gfx9)
    auto loads_landed = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&sy[CMP_HDR + b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if constexpr (KMODE == 0) {
        // ---- cos / sin of the rows' new ids: rope_table_kernel's arithmetic per thread, while the row loads fly
        // (a per-block LDS table of the rows' distinct ids was measured: no faster - with the packed-convert rotation
        // below the kernel is bound by its memory traffic, not by the 8 correctly rounded sincos per thread)
        float c1[VE], s1[VE], c2[VE], s2[VE];
        {
            const float pid[3] = {(float)id[0], (float)id[1], (float)id[2]};
            rope_chunk<VE>(inv_freq, rs, d, h2, pid, scaling, round_mode, c1, s1, c2, s2);
        }
        loads_landed();
        // ---- kept K = the un-rotated row rotated forward at its new position (:297-306): (k*cos) + (rotate_half(k)*sin),
        // one rounding per torch op, no fma contraction; straight into the tail (nobody reads K rows of the tail here)
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            if (u >= nh || !active) break;
            u32x4 olo, ohi;
            rotate_chunk_pair<DT>(k_lo[u], k_hi[u], c1, s1, c2, s2, olo, ohi);
            char* ko = kt + ((size_t)(h0 + u) * un.k_tail_stride_h + (size_t)r * D) * ES;
            *(u32x4*)(ko + (size_t)d * ES) = olo;
            *(u32x4*)(ko + (size_t)(d + h2) * ES) = ohi;
        }
    } else if constexpr (KMODE == 1) {   // the un-rotated rows themselves (deferred re-rotation), from their own buffer
        loads_landed();
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            if (u >= nh || !active) break;
            char* ko = kt + ((size_t)(h0 + u) * un.k_tail_stride_h + (size_t)r * D) * ES;
            *(u32x4*)(ko + (size_t)d * ES) = k_lo[u];
            *(u32x4*)(ko + (size_t)(d + h2) * ES) = k_hi[u];
        }
    }
    if constexpr (KMODE == 2) loads_landed();
    // ---- the in-place rows: the lower blocks' loads landed -> stores ------------------------------------------------
    if (tid < WAVE) {
        for (int bb = tid; bb < b; bb += WAVE) {
            // a lower block raises its flag a few microseconds after it starts; ~2 s of polling means the workspace was
            // not zeroed (a flag can then never be trusted): abort the launch loudly instead of hanging the queue
            unsigned polls = 0;
            while (__hip_atomic_load(&sy[CMP_HDR + bb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                __builtin_amdgcn_s_sleep(2);
                if (++polls > (1u << 21)) __builtin_trap();
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < HU; ++u) {
        if (u >= nh || !active) break;
        char* vo = vt + ((size_t)(h0 + u) * un.v_tail_stride_h + (size_t)r * D) * ES;
        *(u32x4*)(vo + (size_t)d * ES) = v_lo[u];      // torch.gather(value_states, 2, keep)  (:280) + :316-318
        *(u32x4*)(vo + (size_t)(d + h2) * ES) = v_hi[u];
        if constexpr (KMODE == 2) {                    // torch.gather(key_states, 2, keep)  (:279)
            char* ko = kt + ((size_t)(h0 + u) * un.k_tail_stride_h + (size_t)r * D) * ES;
            *(u32x4*)(ko + (size_t)d * ES) = k_lo[u];
            *(u32x4*)(ko + (size_t)(d + h2) * ES) = k_hi[u];
        }
    }
    // ids of the kept tokens -> the layer's position cache (longvideo_cache.py:308-309), once per unit
    if (hg == 0 && un.pos_dst) {
        for (int i = tid; i < P * R; i += CMP_BLOCK) {
            const int p = i / R, rr = b * R + (i - p * R);
            if (rr < keep) un.pos_dst[(size_t)p * un.pos_dst_stride + rr] = un.pos_src[(size_t)p * un.pos_src_stride + rr];
        }
    }
    // the last block to get here has seen every ticket taken and every wait over: tickets and flags back to zero - the
    // workspace is left as it was found, whatever launches next (a replayed graph included)
    if (tid == 0) s_b = atomicAdd(&sy[1], 1) == nb - 1;
    __syncthreads();
    if (s_b) {
        for (int i = tid; i < nb; i += CMP_BLOCK) __hip_atomic_store(&sy[CMP_HDR + i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(&sy[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static int compact_geometry(int D, int dtype, int keep, int* rows_per_block) {
    const int ve = dtype != RTK_F32 ? 8 : 4;
    if (D < 2 * ve || D % (2 * ve) != 0 || D / 2 / ve > CMP_BLOCK) return 0;
    const int R = CMP_BLOCK / (D / 2 / ve);
    if (rows_per_block) *rows_per_block = R;
    return (keep + R - 1) / R;
}

}  // namespace rtk

using namespace rtk;

extern "C" size_t rtk_pivotkv_compact_sync_ints(int n_units, int Hkv, int keep, int D, int dtype) {
    if (n_units < 1 || Hkv < 1 || keep < 1) return 0;
    const int nb = compact_geometry(D, dtype, keep, nullptr);
    if (nb == 0) return 0;
    const size_t stride = (size_t)((CMP_HDR + nb + 31) / 32) * 32;
    return (size_t)n_units * ((Hkv + CMP_HU - 1) / CMP_HU) * stride;
}

extern "C" int rtk_pivotkv_compact_batched(const rtk_compact_unit* units, int n_units, int Hkv, int D, int keep, int P,
                                           int dtype, int k_mode, const float* inv_freq, float attention_scaling,
                                           const int* sections_host, int nsec, int round_mode, int32_t* sync_ws,
                                           size_t sync_ws_ints, rtk_stream_t stream) {
    RTK_CHECK_ARG(units && n_units >= 1, "rtk_pivotkv_compact_batched: no units");
    RTK_CHECK_ARG(Hkv >= 1 && keep >= 1 && D >= 2, "rtk_pivotkv_compact_batched: bad shape");
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_F16, "rtk_pivotkv_compact_batched: unsupported dtype %d", dtype);
    RTK_CHECK_ARG(k_mode == RTK_COMPACT_K_ROTATE || k_mode == RTK_COMPACT_K_COPY || k_mode == RTK_COMPACT_K_INPLACE,
                  "rtk_pivotkv_compact_batched: k_mode %d", k_mode);
    RTK_CHECK_ARG(P == 0 || P == 1 || P == 3, "rtk_pivotkv_compact_batched: P must be 0, 1 or 3, got %d", P);
    RTK_CHECK_ARG(k_mode != RTK_COMPACT_K_ROTATE || (inv_freq && P > 0), "rtk_pivotkv_compact_batched: the rotation needs inv_freq and the new ids");
    RTK_CHECK_ARG(sync_ws, "rtk_pivotkv_compact_batched: needs the zero-initialised sync workspace");
    int R = 0;
    const int nb = compact_geometry(D, dtype, keep, &R);
    if (nb == 0 || D > 256) {
        set_error("rtk_pivotkv_compact_batched: head_dim %d unsupported for this dtype", D);
        return RTK_EUNSUPPORTED;
    }
    const int es = dtype != RTK_F32 ? 2 : 4;
    const int HG = (Hkv + CMP_HU - 1) / CMP_HU;
    const size_t stride = (size_t)((CMP_HDR + nb + 31) / 32) * 32;
    if (sync_ws_ints < (size_t)n_units * HG * stride) {
        set_error("rtk_pivotkv_compact_batched: sync workspace of %zu ints, need %zu", sync_ws_ints, (size_t)n_units * HG * stride);
        return RTK_EWORKSPACE;
    }
    for (int i = 0; i < n_units; ++i) {
        const rtk_compact_unit& u = units[i];
        RTK_CHECK_ARG(u.k_tail && u.v_tail && u.keep_idx, "rtk_pivotkv_compact_batched: unit %d: NULL pointer", i);
        RTK_CHECK_ARG(k_mode == RTK_COMPACT_K_INPLACE || (u.k_src && u.k_src != u.k_tail), "rtk_pivotkv_compact_batched: unit %d: k_src must be a buffer of its own", i);
        RTK_CHECK_ARG(k_mode != RTK_COMPACT_K_ROTATE || u.pos_src, "rtk_pivotkv_compact_batched: unit %d: the rotation needs pos_src", i);
        RTK_CHECK_ARG(!u.pos_dst || (u.pos_src && P > 0), "rtk_pivotkv_compact_batched: unit %d: pos_dst needs pos_src and P", i);
        const bool aligned = (u.k_tail_stride_h * es) % 16 == 0 && (u.v_tail_stride_h * es) % 16 == 0 &&
                             (k_mode == RTK_COMPACT_K_INPLACE || (u.k_src_stride_h * es) % 16 == 0) &&
                             (((uintptr_t)u.k_tail | (uintptr_t)u.v_tail | (k_mode == RTK_COMPACT_K_INPLACE ? 0 : (uintptr_t)u.k_src)) & 15) == 0;
        if (!aligned) {
            set_error("rtk_pivotkv_compact_batched: unit %d: pointers and strides must be 16-byte aligned", i);
            return RTK_EUNSUPPORTED;
        }
    }
    RowSel rs;
    if (k_mode == RTK_COMPACT_K_ROTATE) {
        const int rc = make_rowsel(rs, P, D, sections_host, nsec, "rtk_pivotkv_compact_batched");
        if (rc != RTK_OK) return rc;
    } else {
        for (int d = 0; d < 256; ++d) rs.row[d] = 0;
    }
    hipStream_t st = (hipStream_t)stream;
    for (int b = 0; b < n_units; b += RTK_COMPACT_MAX_UNITS) {
        const int n = std::min(RTK_COMPACT_MAX_UNITS, n_units - b);
        CompactUnits cu;
        for (int i = 0; i < RTK_COMPACT_MAX_UNITS; ++i) cu.u[i] = units[b + std::min(i, n - 1)];
        int32_t* sy = sync_ws + (size_t)b * HG * stride;
#define RTK_LAUNCH_CMP(DTV, KM)                                                                                              \
    RTK_LAUNCH(KID_COMPACT, (compact_units_kernel<DTV, KM>), dim3(nb, n * HG), dim3(CMP_BLOCK), 0, st, cu, Hkv, HG, D, keep, \
               P, inv_freq, attention_scaling, rs, round_mode, sy, (int)stride)
#define RTK_LAUNCH_CMP_DT(KM)                                  \
    do {                                                \
        if (dtype == RTK_BF16) RTK_LAUNCH_CMP(RTK_BF16, KM);   \
        else if (dtype == RTK_F16) RTK_LAUNCH_CMP(RTK_F16, KM); \
        else RTK_LAUNCH_CMP(RTK_F32, KM);                      \
    } while (0)
        if (k_mode == RTK_COMPACT_K_ROTATE) RTK_LAUNCH_CMP_DT(0);
        else if (k_mode == RTK_COMPACT_K_COPY) RTK_LAUNCH_CMP_DT(1);
        else RTK_LAUNCH_CMP_DT(2);
#undef RTK_LAUNCH_CMP_DT
#undef RTK_LAUNCH_CMP
        if (hipError_t e = hipGetLastError(); e != hipSuccess) {
            // an earlier launch of this call may have taken tickets: leave the workspace as the next call expects it
            (void)hipMemsetAsync(sync_ws, 0, (size_t)n_units * HG * stride * sizeof(int32_t), st);
            return rtk::hip_fail(e, "compact_units_kernel");
        }
    }
    return RTK_OK;
}

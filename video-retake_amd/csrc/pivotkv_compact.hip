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
// blocks of its (unit, head group) - the readers of its destination rows have indices <= its own, (5) stores.
// No workgroup waits on a higher ticket or on anything a waiting workgroup holds: no deadlock at any occupancy.
//
// Rotary tables.  The new ids of a block's 32 rows take few distinct values per id row (M-RoPE: 1-2 temporal, a few
// h, <= grid-width w): the block builds cos / sin for (id row, id value, channel of that row's section) once in LDS
// - rope_table_kernel's arithmetic, so the same bits - instead of 8 correctly rounded sincos per thread; blocks whose
// ids spread too far (plain 1-D ids) compute per thread as before.  The rotation rounds through v_cvt_pk_bf16_f32.
#include "common.cuh"

namespace rtk {

struct CompactUnits {
    rtk_compact_unit u[RTK_COMPACT_MAX_UNITS];
};
// the channels d < D/2 grouped by the id row (t / h / w) that feeds them: chan[start[p] + j], j < cnt[p]
struct SecMap {
    uint8_t chan[128];
    uint8_t start[4];
    uint8_t cnt[4];
};

constexpr int CMP_BLOCK = 256;
constexpr int CMP_HU = 4;            // KV heads per workgroup
constexpr int CMP_TAB = 2048;        // (cos, sin) entries of the block's table: 16 KB
constexpr int CMP_HDR = 32;          // ints before the flags of a (unit, head group): [0] ticket, [1] finished blocks

template <int DT, int KMODE>
__global__ __launch_bounds__(CMP_BLOCK, 4) void compact_units_kernel(CompactUnits units, int Hkv, int HG, int D, int keep,
                                                                  int P, const float* __restrict__ inv_freq,
                                                                  float scaling, RowSel rs, SecMap sm, int round_mode,
                                                                  int use_tab, int32_t* __restrict__ sync,
                                                                  int sync_stride, int32_t epoch) {
    using V = Vec16<DT>;
    constexpr int VE = V::VE;
    constexpr int ES = 16 / VE;
    constexpr int HU = CMP_HU;
    __shared__ int s_b;
    __shared__ int s_mm[CMP_BLOCK / WAVE][8];
    __shared__ float2 tab[KMODE == 0 ? CMP_TAB : 1];
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
        if (tid == 0) __hip_atomic_store(&sy[CMP_HDR + b], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if constexpr (KMODE == 0) {
        // ---- cos / sin of the rows' new ids -------------------------------------------------------------------
        float c1[VE], s1[VE], c2[VE], s2[VE];
        int idi[3], mn[3], mx[3];
        bool bad = false;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            bad = bad || id[p] > (1ll << 29) || id[p] < -(1ll << 29);
            idi[p] = (int)id[p];
            mn[p] = mx[p] = idi[p];
        }
        int cum[3] = {0, 0, 0};
        bool tabbed = false;
        if (use_tab) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    mn[p] = min(mn[p], __shfl_xor(mn[p], o, WAVE));
                    mx[p] = max(mx[p], __shfl_xor(mx[p], o, WAVE));
                }
            }
            const bool wbad = __ballot(bad) != 0ull;
            if ((tid & (WAVE - 1)) == 0) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    s_mm[tid / WAVE][p] = mn[p];
                    s_mm[tid / WAVE][3 + p] = mx[p];
                }
                s_mm[tid / WAVE][6] = wbad;
            }
            __syncthreads();
            int anybad = 0;
#pragma unroll
            for (int w = 0; w < CMP_BLOCK / WAVE; ++w) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    mn[p] = min(mn[p], s_mm[w][p]);
                    mx[p] = max(mx[p], s_mm[w][3 + p]);
                }
                anybad |= s_mm[w][6];
            }
            int tot[3], slots = 0, total = 0;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int rng = sm.cnt[p] ? mx[p] - mn[p] + 1 : 0;
                cum[p] = slots;
                slots += rng;
                tot[p] = rng * (int)sm.cnt[p];
                total += tot[p];
            }
            const int rows = min(R, keep - b * R);
            // block-uniform: the table pays when it is at most half the per-thread work and fits
            tabbed = !anybad && slots * h2 <= CMP_TAB && 2 * total <= rows * h2;
            if (tabbed) {
                for (int i = tid; i < total; i += CMP_BLOCK) {
                    int q = i, p = 0;
                    if (q >= tot[0]) {
                        q -= tot[0];
                        p = 1;
                        if (q >= tot[1]) {
                            q -= tot[1];
                            p = 2;
                        }
                    }
                    const int np = sm.cnt[p];
                    const int ido = q / np, j = q - ido * np;
                    const int dch = sm.chan[sm.start[p] + j];
                    const int base = p == 0 ? cum[0] : (p == 1 ? cum[1] : cum[2]);
                    const int m0 = p == 0 ? mn[0] : (p == 1 ? mn[1] : mn[2]);
                    // rope_elem()'s arithmetic: fp32 id * inv_freq, correctly rounded sin / cos, * attention_scaling
                    float sn, cs;
                    sincos_cr((float)(long long)(m0 + ido) * inv_freq[dch], sn, cs);
                    cs = round_to(cs * scaling, round_mode);
                    sn = round_to(sn * scaling, round_mode);
                    tab[(base + ido) * h2 + dch] = make_float2(cs, sn);
                }
            }
        }
        loads_landed();   // (also the barrier between the table's writers and its readers)
        if (tabbed) {
            uint8_t ra[VE];
            if constexpr (VE == 8) {
                const uint64_t wa = *(const uint64_t*)(rs.row + d);
#pragma unroll
                for (int e = 0; e < 8; ++e) ra[e] = (uint8_t)(wa >> (8 * e));
            } else {
                const uint32_t wa = *(const uint32_t*)(rs.row + d);
#pragma unroll
                for (int e = 0; e < VE; ++e) ra[e] = (uint8_t)(wa >> (8 * e));
            }
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const int p = ra[e];
                const int slot = p == 0 ? cum[0] + idi[0] - mn[0] : (p == 1 ? cum[1] + idi[1] - mn[1] : cum[2] + idi[2] - mn[2]);
                const float2 t = tab[slot * h2 + d + e];
                c1[e] = c2[e] = t.x;
                s1[e] = s2[e] = t.y;
            }
        } else {
            const float pid[3] = {(float)id[0], (float)id[1], (float)id[2]};
            rope_chunk<VE>(inv_freq, rs, d, h2, pid, scaling, round_mode, c1, s1, c2, s2);
        }
        // ---- kept K = the un-rotated row rotated forward at its new position (:297-306): (k*cos) + (rotate_half(k)*sin),
        // one rounding per torch op, no fma contraction; straight into the tail (nobody reads K rows of the tail here)
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            if (u >= nh || !active) break;
            u32x4 olo, ohi;
            if constexpr (DT != RTK_F32) {
                using Hh = H16<DT>;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float x1a = Hh::lo(k_lo[u][w]), x1b = Hh::hi(k_lo[u][w]), x2a = Hh::lo(k_hi[u][w]), x2b = Hh::hi(k_hi[u][w]);
                    const int e = 2 * w;
                    const uint32_t p1 = Hh::pack2(x1a * c1[e], x1b * c1[e + 1]);
                    const uint32_t n1 = Hh::pack2(-x2a * s1[e], -x2b * s1[e + 1]);
                    const uint32_t p2 = Hh::pack2(x2a * c2[e], x2b * c2[e + 1]);
                    const uint32_t n2 = Hh::pack2(x1a * s2[e], x1b * s2[e + 1]);
                    olo[w] = Hh::pack2(Hh::lo(p1) + Hh::lo(n1), Hh::hi(p1) + Hh::hi(n1));
                    ohi[w] = Hh::pack2(Hh::lo(p2) + Hh::lo(n2), Hh::hi(p2) + Hh::hi(n2));
                }
            } else {
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const float x1 = __uint_as_float(k_lo[u][e]), x2 = __uint_as_float(k_hi[u][e]);
                    olo[e] = __float_as_uint(__fadd_rn(__fmul_rn(x1, c1[e]), __fmul_rn(-x2, s1[e])));
                    ohi[e] = __float_as_uint(__fadd_rn(__fmul_rn(x2, c2[e]), __fmul_rn(x1, s2[e])));
                }
            }
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
        for (int bb = tid; bb < b; bb += WAVE)
            while (__hip_atomic_load(&sy[CMP_HDR + bb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch)
                __builtin_amdgcn_s_sleep(2);
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
    // the last block to get here has seen every ticket taken: counters back to zero for the next launch
    if (tid == 0) {
        const int done = atomicAdd(&sy[1], 1);
        if (done == nb - 1) {
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
                                           size_t sync_ws_ints, int32_t epoch, rtk_stream_t stream) {
    RTK_CHECK_ARG(units && n_units >= 1, "rtk_pivotkv_compact_batched: no units");
    RTK_CHECK_ARG(Hkv >= 1 && keep >= 1 && D >= 2, "rtk_pivotkv_compact_batched: bad shape");
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_F16, "rtk_pivotkv_compact_batched: unsupported dtype %d", dtype);
    RTK_CHECK_ARG(k_mode == RTK_COMPACT_K_ROTATE || k_mode == RTK_COMPACT_K_COPY || k_mode == RTK_COMPACT_K_INPLACE,
                  "rtk_pivotkv_compact_batched: k_mode %d", k_mode);
    RTK_CHECK_ARG(P == 0 || P == 1 || P == 3, "rtk_pivotkv_compact_batched: P must be 0, 1 or 3, got %d", P);
    RTK_CHECK_ARG(k_mode != RTK_COMPACT_K_ROTATE || (inv_freq && P > 0), "rtk_pivotkv_compact_batched: the rotation needs inv_freq and the new ids");
    RTK_CHECK_ARG(sync_ws && epoch != 0, "rtk_pivotkv_compact_batched: needs the zero-initialised sync workspace and a non-zero epoch");
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
    SecMap sm;
    int use_tab = 0;
    if (k_mode == RTK_COMPACT_K_ROTATE) {
        const int rc = make_rowsel(rs, P, D, sections_host, nsec, "rtk_pivotkv_compact_batched");
        if (rc != RTK_OK) return rc;
        const int h2 = D / 2;
        use_tab = h2 <= 128;
        int at = 0;
        for (int p = 0; p < 3; ++p) {
            sm.start[p] = (uint8_t)at;
            int cnt = 0;
            for (int d = 0; d < h2 && use_tab; ++d) {
                if (rs.row[d] != rs.row[d + h2]) use_tab = 0;   // a channel and its rotation partner on different id rows
                if (rs.row[d] == p) sm.chan[at + cnt++] = (uint8_t)d;
            }
            sm.cnt[p] = (uint8_t)cnt;
            at += cnt;
        }
        sm.start[3] = sm.cnt[3] = 0;
    } else {
        for (int d = 0; d < 256; ++d) rs.row[d] = 0;
        for (int d = 0; d < 128; ++d) sm.chan[d] = 0;
        for (int p = 0; p < 4; ++p) sm.start[p] = sm.cnt[p] = 0;
    }
    hipStream_t st = (hipStream_t)stream;
    for (int b = 0; b < n_units; b += RTK_COMPACT_MAX_UNITS) {
        const int n = std::min(RTK_COMPACT_MAX_UNITS, n_units - b);
        CompactUnits cu;
        for (int i = 0; i < RTK_COMPACT_MAX_UNITS; ++i) cu.u[i] = units[b + std::min(i, n - 1)];
        int32_t* sy = sync_ws + (size_t)b * HG * stride;
#define RTK_CMP(DTV, KM)                                                                                              \
    RTK_LAUNCH(KID_COMPACT, (compact_units_kernel<DTV, KM>), dim3(nb, n * HG), dim3(CMP_BLOCK), 0, st, cu, Hkv, HG, D, keep, \
               P, inv_freq, attention_scaling, rs, sm, round_mode, use_tab, sy, (int)stride, epoch)
#define RTK_CMP_DT(KM)                                  \
    do {                                                \
        if (dtype == RTK_BF16) RTK_CMP(RTK_BF16, KM);   \
        else if (dtype == RTK_F16) RTK_CMP(RTK_F16, KM); \
        else RTK_CMP(RTK_F32, KM);                      \
    } while (0)
        if (k_mode == RTK_COMPACT_K_ROTATE) RTK_CMP_DT(0);
        else if (k_mode == RTK_COMPACT_K_COPY) RTK_CMP_DT(1);
        else RTK_CMP_DT(2);
#undef RTK_CMP_DT
#undef RTK_CMP
        RTK_LAUNCH_CHECK("compact_units_kernel");
    }
    return RTK_OK;
}

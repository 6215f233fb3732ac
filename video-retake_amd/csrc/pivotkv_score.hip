// pivotkv_score.hip — PivotKV token scoring on gfx950.  Replaces longvideo_cache.py:248-270:
//   un-rotate q,k (pos_embed_reforge) -> softmax(q k^T / sqrt(D)) over the CURRENT chunk's keys, no
//   mask -> column sums over queries -> mean over the G heads of a KV group -> mean over groups.
//
// The [Hq,L,L] probability tensor (4.4 GB fp32 at L = 6272) is never materialised.  The row
// normaliser has to be known before a column sum can be accumulated, so the contraction runs twice:
//   pass 1  per (head, 128-query tile): stream key tiles, online row max / sum  -> lse[h,i]
//   pass 2  per (kv group, 128-key tile, row split): stream query tiles of the group's G heads,
//           p = exp(s - lse[h,i]), accumulate per-key column sums            -> partial[g,split,j]
//   finalize: fixed-order reduction of the partials (deterministic, no float atomics)  -> score[j]
// Roofline: MFMA-bound (2*Hq*L^2*D flop per pass, operands are a few MB and L2-resident).
// bf16: v_mfma_f32_32x32x16_bf16 (fp32 accumulate: products of bf16 are exact in fp32).
// fp32: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain) — the parity path.
// In both passes the operand whose statistics are kept (query rows in pass 1, keys in pass 2) sits
// in registers as the MFMA B operand, so every lane owns one row/column (n = lane & 31) and the
// reduction over the streamed operand is lane-local over the 16 accumulator registers plus one
// cross-half shuffle.  The streamed operand goes HBM/L2 -> registers -> XOR-swizzled LDS tile
// (conflict-free ds_read_b128) with the next tile's global loads in flight during the MFMAs.
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <type_traits>

#include "common.cuh"
#include "variants.h"   // A/B knobs (production builds define none of them)

// LLVM sched_group_barrier masks
#define SGB_VALU 0x2
#define SGB_MFMA 0x8
#define SGB_DS_READ 0x100
#define SGB_TRANS 0x400

namespace rtk {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int HD = 128;       // head_dim of the MFMA path
constexpr int TILE_ROWS = 64; // rows of the streamed LDS tile
constexpr int REG_ROWS = 128; // rows held in registers per workgroup (32 per wave)
constexpr int SC_BLOCK = 256;
constexpr int NXCD = 8;      // MI355X: 8 XCDs, workgroup b is dispatched to XCD b % 8

// ------------------------------------------------------------------------------------------------
// un-rotate + pack:  q [Hq,L,D] and k [Hkv,L,D] (strided) -> contiguous [H,L,D] copies (same dtype)
//   cos == NULL: plain copy;  else ((x*cos) - (rotate_half(x)*sin)) / a^2  with one rounding per
//   torch op (bf16: every intermediate is a bf16 tensor; fp32: no fma contraction).
// One thread owns a 16-byte chunk of the first half of a token row plus its rotation partner in the
// second half, keeps that token's cos/sin in registers and walks UNROT_HEADS heads with it.
// ------------------------------------------------------------------------------------------------
constexpr int UNROT_HEADS = 7;

// bf16 pairs through the hardware converter (v_cvt_pk_bf16_f32: round to nearest even, like c10::BFloat16)
using bf16x2_t = __attribute__((ext_vector_type(2))) __bf16;
using f32x2_t = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// RTK_BF16_FAST operands: the un-rotated bf16 values re-encoded as fp16 for v_mfma_f32_32x32x16_f16.  A bf16 value has 8
// significant bits, fp16 holds 11: k~ converts EXACTLY (inside fp16's range; saturated to +-65504 beyond it, 24-bit
// subnormals below 6e-5), and q~ * log2(e)/sqrt(D) is rounded once, to 11 bits (relative 2^-12), so the matrix pipe
// delivers the base-2 logits directly and the softmax needs no multiply.
using f16x2_t = __attribute__((ext_vector_type(2))) _Float16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
__device__ __forceinline__ uint32_t pack2_f16(float lo, float hi) {
    const f32x2_t v = {__builtin_fminf(__builtin_fmaxf(lo, -65504.f), 65504.f),
                       __builtin_fminf(__builtin_fmaxf(hi, -65504.f), 65504.f)};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));   // v_cvt_f16_f32: round to nearest even
}
__device__ __forceinline__ u32x4 bf16x8_to_f16x8(const u32x4& v, float scale) {
    return u32x4{pack2_f16(bf_lo(v.x) * scale, bf_hi(v.x) * scale), pack2_f16(bf_lo(v.y) * scale, bf_hi(v.y) * scale),
                 pack2_f16(bf_lo(v.z) * scale, bf_hi(v.z) * scale), pack2_f16(bf_lo(v.w) * scale, bf_hi(v.w) * scale)};
}

// DIV: 0 = no division (attention_scaling^2 == 1), 1 = multiply by the reciprocal (bf16 only, the host has
// verified EXHAUSTIVELY over all 65536 bf16 inputs that bf16(x * rcp) == bf16(x / a2) for this a2), 2 = IEEE
// FAST (RTK_BF16_FAST, bf16 inputs only): q~ is stored as fp16(q~ * qscale) and k~ additionally as fp16 in k_fast (the
// bf16 k~ in k_out - what the eviction re-rotates - is skipped when k_out is NULL).
template <int DT, int DIV, bool FAST = false>
__global__ __launch_bounds__(256) void unrotate_pack_vec_kernel(const char* __restrict__ q, int64_t q_sh, int64_t q_sl,
                                                                const char* __restrict__ k, int64_t k_sh, int64_t k_sl,
                                                                int Hq, int Hkv, int L, int D,
                                                                const float* __restrict__ cosv,
                                                                const float* __restrict__ sinv, float a2, float rcp_a2,
                                                                char* __restrict__ q_out, char* __restrict__ k_out,
                                                                char* __restrict__ k_fast = nullptr, float qscale = 1.f) {
    using V = Vec16<DT>;
    constexpr int VE = V::VE;
    constexpr int ES = 16 / VE;
    const int h2 = D / 2, lpr = h2 / VE;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= L * lpr) return;
    const int l = id / lpr, d = (id - l * lpr) * VE;
    // blockIdx.y walks head groups: first the q groups, then the k groups
    const int qgroups = (Hq + UNROT_HEADS - 1) / UNROT_HEADS;
    const bool is_q = (int)blockIdx.y < qgroups;
    const int hg = is_q ? blockIdx.y : blockIdx.y - qgroups;
    const int H = is_q ? Hq : Hkv;
    const char* src = is_q ? q : k;
    char* dst = is_q ? q_out : k_out;
    const int64_t sh = is_q ? q_sh : k_sh, sl = is_q ? q_sl : k_sl;
    float c1[VE], s1[VE], c2[VE], s2[VE];
    if (cosv) {
#pragma unroll
        for (int e = 0; e < VE; e += 4) {
            *(float4*)(c1 + e) = *(const float4*)(cosv + (size_t)l * D + d + e);
            *(float4*)(s1 + e) = *(const float4*)(sinv + (size_t)l * D + d + e);
            *(float4*)(c2 + e) = *(const float4*)(cosv + (size_t)l * D + d + h2 + e);
            *(float4*)(s2 + e) = *(const float4*)(sinv + (size_t)l * D + d + h2 + e);
        }
    }
    const int hb = hg * UNROT_HEADS;
    // all loads first (UNROT_HEADS independent row pairs in flight), then the arithmetic and the stores
    u32x4 lo[UNROT_HEADS], hi[UNROT_HEADS];
#pragma unroll
    for (int u = 0; u < UNROT_HEADS; ++u) {
        const int h = min(hb + u, H - 1);
        const char* row = src + ((size_t)h * sh + (size_t)l * sl) * ES;
        lo[u] = *(const u32x4*)(row + (size_t)d * ES);
        hi[u] = *(const u32x4*)(row + (size_t)(d + h2) * ES);
    }
#pragma unroll
    for (int u = 0; u < UNROT_HEADS; ++u) {
        const int h = hb + u;
        if (h >= H) break;
        char* orow = dst + ((size_t)h * L + l) * D * ES;
        // FAST: where the un-rotated chunk pair (bf16 values) goes - q as scaled fp16, k as bf16 (if wanted) + fp16
        auto store_fast = [&](const u32x4& a, const u32x4& b) {
            if (is_q) {
                *(u32x4*)(orow + (size_t)d * ES) = bf16x8_to_f16x8(a, qscale);
                *(u32x4*)(orow + (size_t)(d + h2) * ES) = bf16x8_to_f16x8(b, qscale);
            } else {
                if (dst) {
                    *(u32x4*)(orow + (size_t)d * ES) = a;
                    *(u32x4*)(orow + (size_t)(d + h2) * ES) = b;
                }
                char* frow = k_fast + ((size_t)h * L + l) * D * ES;
                *(u32x4*)(frow + (size_t)d * ES) = bf16x8_to_f16x8(a, 1.f);
                *(u32x4*)(frow + (size_t)(d + h2) * ES) = bf16x8_to_f16x8(b, 1.f);
            }
        };
        if (!cosv) {
            if constexpr (FAST) { store_fast(lo[u], hi[u]); continue; }
            *(u32x4*)(orow + (size_t)d * ES) = lo[u];
            *(u32x4*)(orow + (size_t)(d + h2) * ES) = hi[u];
            continue;
        }
        // rotate_half(x)[d] = -x2, rotate_half(x)[d+h2] = x1   (longvideo_cache.py:28-32)
        // x~ = ((x*cos) - (rotate_half(x)*sin)) / a^2, one rounding per torch op (:76-78)
        if constexpr (DT != RTK_F32) {   // bf16 / fp16: every torch op rounds to the tensor dtype
            using Hh = H16<DT>;
            const uint32_t wl[4] = {lo[u].x, lo[u].y, lo[u].z, lo[u].w}, wh[4] = {hi[u].x, hi[u].y, hi[u].z, hi[u].w};
            uint32_t r1[4], r2[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float x1a = Hh::lo(wl[w]), x1b = Hh::hi(wl[w]), x2a = Hh::lo(wh[w]), x2b = Hh::hi(wh[w]);
                const int e = 2 * w;
                const uint32_t p1 = Hh::pack2(x1a * c1[e], x1b * c1[e + 1]);          // x1*cos
                const uint32_t n1 = Hh::pack2(x2a * s1[e], x2b * s1[e + 1]);          // -(rotate_half(x)*sin) = x2*sin
                const uint32_t p2 = Hh::pack2(x2a * c2[e], x2b * c2[e + 1]);          // x2*cos
                const uint32_t n2 = Hh::pack2(x1a * s2[e], x1b * s2[e + 1]);          // rotate_half(x)*sin = x1*sin
                uint32_t t1 = Hh::pack2(Hh::lo(p1) + Hh::lo(n1), Hh::hi(p1) + Hh::hi(n1));
                uint32_t t2 = Hh::pack2(Hh::lo(p2) - Hh::lo(n2), Hh::hi(p2) - Hh::hi(n2));
                if constexpr (DIV == 1) {
                    t1 = Hh::pack2(Hh::lo(t1) * rcp_a2, Hh::hi(t1) * rcp_a2);
                    t2 = Hh::pack2(Hh::lo(t2) * rcp_a2, Hh::hi(t2) * rcp_a2);
                } else if constexpr (DIV == 2) {
                    t1 = Hh::pack2(__fdiv_rn(Hh::lo(t1), a2), __fdiv_rn(Hh::hi(t1), a2));
                    t2 = Hh::pack2(__fdiv_rn(Hh::lo(t2), a2), __fdiv_rn(Hh::hi(t2), a2));
                }
                r1[w] = t1;
                r2[w] = t2;
            }
            if constexpr (FAST) {
                store_fast(u32x4{r1[0], r1[1], r1[2], r1[3]}, u32x4{r2[0], r2[1], r2[2], r2[3]});
            } else {
                *(u32x4*)(orow + (size_t)d * ES) = u32x4{r1[0], r1[1], r1[2], r1[3]};
                *(u32x4*)(orow + (size_t)(d + h2) * ES) = u32x4{r2[0], r2[1], r2[2], r2[3]};
            }
        } else {
            float x1[VE], x2[VE], o1[VE], o2[VE];
            V::unpack(lo[u], x1);
            V::unpack(hi[u], x2);
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                o1[e] = __fsub_rn(__fmul_rn(x1[e], c1[e]), __fmul_rn(-x2[e], s1[e]));
                o2[e] = __fsub_rn(__fmul_rn(x2[e], c2[e]), __fmul_rn(x1[e], s2[e]));
                if constexpr (DIV != 0) {
                    o1[e] = __fdiv_rn(o1[e], a2);
                    o2[e] = __fdiv_rn(o2[e], a2);
                }
            }
            *(u32x4*)(orow + (size_t)d * ES) = V::pack(o1);
            *(u32x4*)(orow + (size_t)(d + h2) * ES) = V::pack(o2);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused prepare (native RoPE tables only): one pass over the chunk's q, k, v that
//   builds the token's cos/sin chunk in registers (what rope_table_kernel writes to HBM: same sincos_cr, same
//   scaling, same bf16 rounding), un-rotates q and k with it (same arithmetic as unrotate_pack_vec_kernel),
//   and appends k and v to the cache tail (what append_kernel does) — k is read once instead of twice and
//   three launches become one.  One thread = one token x one 16-byte chunk pair; blockIdx.y splits the heads
//   in two: y = 0 the first half of the q heads + k (k~ and the k tail), y = 1 the second half + the v tail.
// ------------------------------------------------------------------------------------------------
// NW = 32-bit words per thread and row half: 4 (16-byte accesses; fp32) or, for the 16-bit dtypes, 1 (2: A/B builds).  A
// wave of the 16-byte form issues ~3500 VALU instructions (the per-op rounding chains of 18 heads plus 8 correctly
// rounded sin / cos pairs) and a chunk of 2304 tokens gives barely half the chip's SIMDs one such wave: that form is bound
// by the serial instruction stream of its waves.  Narrow chunks split the same work over 4x the waves; what remains is the
// read + write traffic (45 MB per call at L = 2304) at ~3 TB/s plus the launch ramp.

constexpr int RTK_SHIFT_COUNTERS = 64;   // arrival counters of RTK_UPDATE_SHIFT_NEXT (<= RTK_PREP_BLOCK: one per watching thread)
constexpr int RTK_SHIFT_STRIDE = 32;     // ... 128 bytes apart
constexpr int RTK_SHIFT_STATUS = RTK_SHIFT_STRIDE - 1;     // word of the first line that latches a wait that ran out
// polls of the watching workgroup before it gives up (an agent-scope load, a barrier and s_sleep 4 per poll: ~2-3 s; the
// workers need microseconds).  Same policy as compact_units_kernel's bounded wait, but latched instead of trapped: the
// host can raise, reset and carry on, and the test suite can force it.
constexpr unsigned RTK_SHIFT_MAX_POLLS = 1u << 21;
static_assert(RTK_SHIFT_COUNTERS <= RTK_PREP_BLOCK, "one watching thread per counter");

template <int DT, int DIV, bool FAST = false, int NW = 4>
__global__ __launch_bounds__(RTK_PREP_BLOCK) void prepare_native_kernel(const char* __restrict__ q, int64_t q_sh, int64_t q_sl,
                                                            const char* __restrict__ k, int64_t k_sh, int64_t k_sl,
                                                            const char* __restrict__ v, int64_t v_sh, int64_t v_sl,
                                                            int Hq, int Hkv, int L, int D,
                                                            const int64_t* pos, int64_t pos_ld,
                                                            const float* __restrict__ inv_freq, float scaling, RowSel rs,
                                                            int round_bf16, float a2, float rcp_a2,
                                                            char* __restrict__ q_out, char* __restrict__ k_out,
                                                            char* __restrict__ k_tail, char* __restrict__ v_tail,
                                                            int64_t tail_sh, int P, int64_t* __restrict__ pos_copy,
                                                            char* __restrict__ k_fast = nullptr, float qscale = 1.f,
                                                            int64_t* shift_row = nullptr, const int64_t* next_prev = nullptr,
                                                            int* ticket = nullptr, int* status = nullptr) {
    using V = Vec16<DT>;
    static_assert(NW == 4 || ((NW == 2 || NW == 1) && DT != RTK_F32), "8- / 4-byte chunks: 16-bit dtypes only");
    constexpr int ES = 16 / V::VE;          // bytes per element
    constexpr int VE = 4 * NW / ES;         // elements per thread and row half
    using W = WV<NW>;
    const int h2 = D / 2, lpr = h2 / VE;
    // RTK_UPDATE_SHIFT_NEXT: the NEXT layer's continuity shift (qwen2_vl.py:68-73) rides in this launch.  Every working
    // workgroup reads the chunk's ids, so the row may only be rewritten once all of them have.  Each adds one to one of
    // RTK_SHIFT_COUNTERS counters (own cache lines) once its ids are in registers - fire and forget, nobody waits; the
    // FIRST workgroup of the grid (an extra column) does no other work: it watches the counters reach the launch's totals,
    // zeroes them for the next launch, rewrites the row and counts the launch in ticket[0], beside the others' work.
    // The wait is bounded by a POLL count (polls only advance while this wave runs: a process that is switched out, a
    // debugger, a throttled clock cannot trip it - a wall-clock bound could).  If it ever runs out - the counters were
    // not zero at launch, i.e. the words were shared or not zeroed - NOTHING is shifted and NOTHING is zeroed: the watcher
    // latches ticket[RTK_SHIFT_STATUS] (and the host-visible *status, if given) and returns, and so does the watcher of
    // every later launch until the host has seen the latch and reset the words (PivotKVCache raises: the ids of the layers
    // after the failed launch were not shifted).  A row is only ever rewritten after every reader was counted in.
    const int bx = (int)blockIdx.x - (shift_row ? 1 : 0), gx = (int)gridDim.x - (shift_row ? 1 : 0);
    if (bx < 0) {
        if (blockIdx.y != 0) return;
        __shared__ int s_latched;
        if (threadIdx.x == 0)
            s_latched = __hip_atomic_load(ticket + RTK_SHIFT_STATUS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        __syncthreads();
        if (s_latched) return;   // an earlier launch's wait ran out: the counters are not trustworthy until the host resets them
        constexpr int E = 8;   // ids per thread and round, two rounds in flight
        const int nwork = gx * (int)gridDim.y, step = E * (int)blockDim.x;
        const long long delta = (next_prev ? (long long)next_prev[0] : -1ll) + 1 - (long long)shift_row[0];
        long long v[2][E];
        auto fetch = [&](long long* r, int base) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = base + e * (int)blockDim.x + (int)threadIdx.x;
                r[e] = i < L ? (long long)shift_row[i] : 0;
            }
        };
        fetch(v[0], 0);        // the first round of the row is on its way while the others start up
        // counter c takes the workgroups whose linear index is c modulo RTK_SHIFT_COUNTERS
        const int c = (int)threadIdx.x;   // (blockDim.x >= RTK_SHIFT_COUNTERS: one counter per thread)
        unsigned* mine = (unsigned*)ticket + RTK_SHIFT_STRIDE * (1 + c);
        const unsigned want = (unsigned)(nwork / RTK_SHIFT_COUNTERS + (c < nwork % RTK_SHIFT_COUNTERS ? 1 : 0));
        unsigned polls = 0;
        for (;;) {
            int ok = 1;
            if (c < RTK_SHIFT_COUNTERS) ok = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want;
            if (__syncthreads_and(ok)) break;
            if (++polls > RTK_SHIFT_MAX_POLLS) {   // (uniform: every thread counts the same polls)
                if (threadIdx.x == 0) {
                    __hip_atomic_fetch_add(ticket + RTK_SHIFT_STATUS, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (status) {
                        __hip_atomic_fetch_add(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __threadfence_system();
                    }
                }
                return;        // ids untouched, counters untouched
            }
            __builtin_amdgcn_s_sleep(4);
        }
        if (c < RTK_SHIFT_COUNTERS) __hip_atomic_store(mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (delta != 0) {      // t[0:L] += (prev_next + 1) - t[0]
            int cur = 0;
            for (int base = 0; base < L; base += step, cur ^= 1) {
                if (base + step < L) fetch(v[cur ^ 1], base + step);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = base + e * (int)blockDim.x + (int)threadIdx.x;
                    if (i < L) shift_row[i] = v[cur][e] + delta;
                }
            }
        }
        if (threadIdx.x == 0) ticket[0] += 1;   // launches that carried a shift (diagnostics)
        return;
    }
    const int id = bx * (int)blockDim.x + (int)threadIdx.x;
    if (id >= L * lpr) return;
    const int l = id / lpr, d = (id - l * lpr) * VE;
    if (pos_copy && blockIdx.y == 0 && d == 0)   // the ids the caller may shift in place before the deferred selection runs
        for (int p = 0; p < P; ++p) pos_copy[(size_t)p * L + l] = pos[(size_t)p * pos_ld + l];
    constexpr int HU = RTK_PREP_HU;   // heads per batch: all loads of a batch are issued before its arithmetic and stores
    const int ny = gridDim.y, qper = (Hq + ny - 1) / ny;
#if RTK_PREP_UBASE
    const int qb = uniform_int(min((int)blockIdx.y * qper, Hq)), qe = uniform_int(min(qb + qper, Hq));   // (head loops in SGPRs)
#else
    const int qb = min((int)blockIdx.y * qper, Hq), qe = min(qb + qper, Hq);
#endif
    // the KV heads: y = 0 takes k (k~ for the scoring / eviction + the rotated rows for the tail), the last y takes v
    const bool has_kv = blockIdx.y == 0 || (int)blockIdx.y == ny - 1;
    const char* src = blockIdx.y == 0 ? k : v;
    const int64_t sh = blockIdx.y == 0 ? k_sh : v_sh, sl = blockIdx.y == 0 ? k_sl : v_sl;
    char* tail = blockIdx.y == 0 ? k_tail : v_tail;
    const int nkv = has_kv ? Hkv : 0;
    // Software pipeline over head batches: the rows of batch b+1 (after the last query batch: the first KV batch)
    // are requested before batch b is un-rotated and stored, and the first batch before the table arithmetic
    // (sin / cos are ~25 fp64 operations per value) - with ~1.5 waves per SIMD nothing else hides a round trip.
    W lo[HU], hi[HU], lon[HU], hin[HU];
#if RTK_PREP_UBASE
    // a row's address = descriptor (tensor base) + soffset (the head: wave-uniform, a scalar multiply) + voffset (this
    // thread's byte offset inside a head, computed once); the launcher has checked that every extent fits 31 bits
    const uint32_t off_q = (uint32_t)(((int64_t)l * q_sl + d) * ES), off_kv = (uint32_t)(((int64_t)l * sl + d) * ES);
    const uint32_t off_o = (uint32_t)(((int64_t)l * D + d) * ES), half = (uint32_t)(h2 * ES);
    const uint32_t off_q2 = off_q + half, off_kv2 = off_kv + half, off_o2 = off_o + half;
    const __amdgpu_buffer_rsrc_t r_q = buf_rsrc(q), r_src = buf_rsrc(src), r_qo = buf_rsrc(q_out), r_ko = buf_rsrc(k_out),
                                 r_tail = buf_rsrc(tail), r_kf = buf_rsrc(k_fast);
    const uint32_t hs_q = (uint32_t)(q_sh * ES), hs_kv = (uint32_t)(sh * ES), hs_o = (uint32_t)((int64_t)L * D * ES),
                   hs_t = (uint32_t)(tail_sh * ES);
#endif
    auto load_q = [&](W* a, W* b, int hb) {
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            const int h = min(hb + u, qe - 1);
#if RTK_PREP_UBASE
            const uint32_t so = (uint32_t)uniform_int((int)((uint32_t)h * hs_q));
            a[u] = buf_load<NW>(r_q, off_q, so);
            b[u] = buf_load<NW>(r_q, off_q2, so);
#else
            const char* row = q + ((size_t)h * q_sh + (size_t)l * q_sl) * ES;
            a[u] = *(const W*)(row + (size_t)d * ES);
            b[u] = *(const W*)(row + (size_t)(d + h2) * ES);
#endif
        }
    };
    auto load_kv = [&](W* a, W* b, int hb) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int h = min(hb + u, Hkv - 1);
#if RTK_PREP_UBASE
            const uint32_t so = (uint32_t)uniform_int((int)((uint32_t)h * hs_kv));
            a[u] = buf_load<NW>(r_src, off_kv, so);
            b[u] = buf_load<NW>(r_src, off_kv2, so);
#else
            const char* row = src + ((size_t)h * sh + (size_t)l * sl) * ES;
            a[u] = *(const W*)(row + (size_t)d * ES);
            b[u] = *(const W*)(row + (size_t)(d + h2) * ES);
#endif
        }
    };
    float pid[3];   // the token's ids (t / h / w rows; a 1-D id fills all three)
#pragma unroll
    for (int p = 0; p < 3; ++p) pid[p] = (float)pos[(size_t)min(p, P - 1) * pos_ld + l];
    if (qb < qe) load_q(lo, hi, qb);
    else if (nkv) load_kv(lo, hi, 0);
    float c1[VE], s1[VE], c2[VE], s2[VE];
    rope_chunk<VE>(inv_freq, rs, d, h2, pid, scaling, round_bf16, c1, s1, c2, s2);
    if (shift_row) {   // (kernel argument: uniform)  this workgroup holds its ids: count it in (see the top of the kernel)
        __syncthreads();   // every wave is past rope_chunk (pid consumed) and past the pos_copy stores (their loads returned)
        if (threadIdx.x == 0)
            __hip_atomic_fetch_add((unsigned*)ticket + RTK_SHIFT_STRIDE * (1 + ((int)blockIdx.y * gx + bx) % RTK_SHIFT_COUNTERS),
                                   1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // x~ = ((x*cos) - (rotate_half(x)*sin)) / a^2 for one head's chunk pair, one rounding per torch op (:76-78)
    auto unrot = [&](const W& lo, const W& hi, W& olo, W& ohi) {
        if constexpr (DT != RTK_F32) {
            using Hh = H16<DT>;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float x1a = Hh::lo(lo.w[w]), x1b = Hh::hi(lo.w[w]), x2a = Hh::lo(hi.w[w]), x2b = Hh::hi(hi.w[w]);
                const int e = 2 * w;
                const uint32_t p1 = Hh::pack2(x1a * c1[e], x1b * c1[e + 1]);
                const uint32_t n1 = Hh::pack2(x2a * s1[e], x2b * s1[e + 1]);
                const uint32_t p2 = Hh::pack2(x2a * c2[e], x2b * c2[e + 1]);
                const uint32_t n2 = Hh::pack2(x1a * s2[e], x1b * s2[e + 1]);
                uint32_t t1 = Hh::pack2(Hh::lo(p1) + Hh::lo(n1), Hh::hi(p1) + Hh::hi(n1));
                uint32_t t2 = Hh::pack2(Hh::lo(p2) - Hh::lo(n2), Hh::hi(p2) - Hh::hi(n2));
                if constexpr (DIV == 1) {
                    t1 = Hh::pack2(Hh::lo(t1) * rcp_a2, Hh::hi(t1) * rcp_a2);
                    t2 = Hh::pack2(Hh::lo(t2) * rcp_a2, Hh::hi(t2) * rcp_a2);
                } else if constexpr (DIV == 2) {
                    t1 = Hh::pack2(__fdiv_rn(Hh::lo(t1), a2), __fdiv_rn(Hh::hi(t1), a2));
                    t2 = Hh::pack2(__fdiv_rn(Hh::lo(t2), a2), __fdiv_rn(Hh::hi(t2), a2));
                }
                olo.w[w] = t1;
                ohi.w[w] = t2;
            }
        } else {
            float o1[VE], o2[VE];
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const float x1 = __uint_as_float(lo.w[e]), x2 = __uint_as_float(hi.w[e]);
                o1[e] = __fsub_rn(__fmul_rn(x1, c1[e]), __fmul_rn(-x2, s1[e]));
                o2[e] = __fsub_rn(__fmul_rn(x2, c2[e]), __fmul_rn(x1, s2[e]));
                if constexpr (DIV != 0) {
                    o1[e] = __fdiv_rn(o1[e], a2);
                    o2[e] = __fdiv_rn(o2[e], a2);
                }
                olo.w[e] = __float_as_uint(o1[e]);
                ohi.w[e] = __float_as_uint(o2[e]);
            }
        }
    };
    auto to_f16 = [&](const W& x, float scale) {   // bf16 pairs -> fp16 pairs of (value * scale)
        W o;
#pragma unroll
        for (int w = 0; w < NW; ++w) o.w[w] = pack2_f16(bf_lo(x.w[w]) * scale, bf_hi(x.w[w]) * scale);
        return o;
    };
    for (int hb = qb; hb < qe; hb += HU) {
        if (hb + HU < qe) load_q(lon, hin, hb + HU);
        else if (nkv) load_kv(lon, hin, 0);
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            const int h = hb + u;
            if (h >= qe) break;
            W olo, ohi;
            unrot(lo[u], hi[u], olo, ohi);
            if constexpr (FAST) {   // the score's A / B operand: fp16(q~ * log2(e)/sqrt(D))
                olo = to_f16(olo, qscale);
                ohi = to_f16(ohi, qscale);
            }
#if RTK_PREP_UBASE
            const uint32_t so = (uint32_t)uniform_int((int)((uint32_t)h * hs_o));
            buf_store<NW>(olo, r_qo, off_o, so);
            buf_store<NW>(ohi, r_qo, off_o2, so);
#else
            char* orow = q_out + ((size_t)h * L + l) * D * ES;
            *(W*)(orow + (size_t)d * ES) = olo;
            *(W*)(orow + (size_t)(d + h2) * ES) = ohi;
#endif
        }
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            lo[u] = lon[u];
            hi[u] = hin[u];
        }
    }
    for (int hb = 0; hb < nkv; hb += 4) {
        if (hb + 4 < nkv) load_kv(lon, hin, hb + 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int h = hb + u;
            if (h >= nkv) break;
#if RTK_PREP_UBASE
            const uint32_t sot = (uint32_t)uniform_int((int)((uint32_t)h * hs_t));
            buf_store<NW>(lo[u], r_tail, off_o, sot);
            buf_store<NW>(hi[u], r_tail, off_o2, sot);
#else
            char* trow = tail + ((size_t)h * tail_sh + (size_t)l * D) * ES;
            *(W*)(trow + (size_t)d * ES) = lo[u];
            *(W*)(trow + (size_t)(d + h2) * ES) = hi[u];
#endif
            if (blockIdx.y == 0) {
                W olo, ohi;
                unrot(lo[u], hi[u], olo, ohi);
#if RTK_PREP_UBASE
                const uint32_t so = (uint32_t)uniform_int((int)((uint32_t)h * hs_o));
                buf_store<NW>(olo, r_ko, off_o, so);
                buf_store<NW>(ohi, r_ko, off_o2, so);
                if constexpr (FAST) {   // the same k~ as fp16 for the score passes (exact re-encoding)
                    buf_store<NW>(to_f16(olo, 1.f), r_kf, off_o, so);
                    buf_store<NW>(to_f16(ohi, 1.f), r_kf, off_o2, so);
                }
#else
                char* orow = k_out + ((size_t)h * L + l) * D * ES;
                *(W*)(orow + (size_t)d * ES) = olo;
                *(W*)(orow + (size_t)(d + h2) * ES) = ohi;
                if constexpr (FAST) {   // the same k~ as fp16 for the score passes (exact re-encoding)
                    char* frow = k_fast + ((size_t)h * L + l) * D * ES;
                    *(W*)(frow + (size_t)d * ES) = to_f16(olo, 1.f);
                    *(W*)(frow + (size_t)(d + h2) * ES) = to_f16(ohi, 1.f);
                }
#endif
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            lo[u] = lon[u];
            hi[u] = hin[u];
        }
    }
}

// scalar fallback (any even head_dim, any alignment)
template <int DT>
__global__ __launch_bounds__(256) void unrotate_pack_kernel(const void* __restrict__ xv, int64_t stride_h,
                                                            int64_t stride_l, int H, int L, int D,
                                                            const float* __restrict__ cosv,
                                                            const float* __restrict__ sinv, float a2,
                                                            void* __restrict__ outv) {
    const int h2 = D / 2;
    const size_t total = (size_t)H * L * h2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % h2);
        const size_t hl = i / h2;
        const int l = (int)(hl % L);
        const int h = (int)(hl / L);
        const size_t src = (size_t)h * stride_h + (size_t)l * stride_l;
        const size_t dst = hl * D;
        float x1, x2;
        if constexpr (DT != RTK_F32) {
            x1 = H16<DT>::ld(xv, src + d);
            x2 = H16<DT>::ld(xv, src + d + h2);
        } else {
            x1 = ((const float*)xv)[src + d];
            x2 = ((const float*)xv)[src + d + h2];
        }
        float o1 = x1, o2 = x2;
        if (cosv) {
            const float c1 = cosv[(size_t)l * D + d], s1 = sinv[(size_t)l * D + d];
            const float c2 = cosv[(size_t)l * D + d + h2], s2 = sinv[(size_t)l * D + d + h2];
            if constexpr (DT != RTK_F32) {
                using Hh = H16<DT>;
                o1 = Hh::rnd(Hh::rnd(Hh::rnd(x1 * c1) - Hh::rnd(-x2 * s1)) / a2);
                o2 = Hh::rnd(Hh::rnd(Hh::rnd(x2 * c2) - Hh::rnd(x1 * s2)) / a2);
            } else {
                o1 = __fdiv_rn(__fsub_rn(__fmul_rn(x1, c1), __fmul_rn(-x2, s1)), a2);
                o2 = __fdiv_rn(__fsub_rn(__fmul_rn(x2, c2), __fmul_rn(x1, s2)), a2);
            }
        }
        if constexpr (DT != RTK_F32) {
            H16<DT>::st(outv, dst + d, o1);
            H16<DT>::st(outv, dst + d + h2, o2);
        } else {
            ((float*)outv)[dst + d] = o1;
            ((float*)outv)[dst + d + h2] = o2;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA building blocks (head_dim 128).  A "chunk" is 16 bytes of a row.
//   bf16: 16 chunks/row; MFMA step s (K=16) uses chunk 2s + half   (half = lane >> 5)
//   fp32: 32 chunks/row; the k axis is re-associated so that half `hf` owns k in [64hf, 64hf+64):
//         chunk 16hf + c feeds MFMAs 4c..4c+3 (K=2 each).  The same permutation is applied to both
//         operands, so every product a_k*b_k still meets its partner; only the summation order
//         differs from index order, which fp32 parity tolerates.
// ------------------------------------------------------------------------------------------------
template <int DT> struct MM;

template <> struct MM<RTK_BF16> {
    static constexpr int ESIZE = 2;
    static constexpr int CHUNKS = 16;            // per row
    static constexpr int NREG = 8;               // 16-byte registers per lane for a 32-row fragment
    __device__ static __forceinline__ int chunk_of(int r, int hf) { return 2 * r + hf; }
    __device__ static __forceinline__ void mma(f32x16& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                      acc, 0, 0, 0);
    }
};

template <> struct MM<RTK_F32> {
    static constexpr int ESIZE = 4;
    static constexpr int CHUNKS = 32;
    static constexpr int NREG = 16;
    __device__ static __forceinline__ int chunk_of(int r, int hf) { return 16 * hf + r; }
    __device__ static __forceinline__ void mma(f32x16& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

template <int DT> struct Tile {
    using M = MM<DT>;
    static constexpr int ROWB = M::CHUNKS * 16;                    // bytes per row (256 / 512)
    static constexpr int BYTES = TILE_ROWS * ROWB;                 // one LDS tile
    static constexpr int STAGE = (TILE_ROWS * M::CHUNKS) / SC_BLOCK;  // 16-byte chunks per thread per tile (4 / 8)
    static constexpr int ROWS_PER_STEP = SC_BLOCK / M::CHUNKS;     // rows between a thread's consecutive chunks (16 / 8)
};

// accumulator register r of lane (half hf) holds output row  m = (r&3) + 8*(r>>2) + 4*hf
__device__ __forceinline__ int acc_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

// Register fragment: row (lane & 31) of a 32-row block starting at `row0` of a contiguous [rows,128] matrix.
template <int DT>
__device__ __forceinline__ void load_reg_frag(const char* __restrict__ base, int row0, int nrows, int lane,
                                              u32x4* rf, int pitch = HD * MM<DT>::ESIZE) {
    using M = MM<DT>;
    const int row = row0 + (lane & 31), hf = lane >> 5;
    const bool ok = row < nrows;
    const u32x4* p = (const u32x4*)(base + (size_t)row * pitch);   // pitch: bytes between rows (a strided projection)
#pragma unroll
    for (int r = 0; r < M::NREG; ++r) rf[r] = ok ? p[M::chunk_of(r, hf)] : u32x4{0, 0, 0, 0};
}

// Per-thread constants of the streamed-tile pipeline, computed once per kernel:
//   frag_off[r]  LDS byte offset (inside a tile, block 0) of this lane's r-th A-fragment chunk
//   st_off[u]    LDS byte offset where this thread stores its u-th staged chunk
//   voff[u]      byte offset of the u-th staged chunk inside the tile's source rows (buffer-load voffset)
// 16-byte chunks are XOR-swizzled by (row & 15): the 16 lanes of every ds_read_b128 lane group address
// 16 distinct rows (mod 16) => 16 distinct 16-byte bank slots; no bank conflicts (SQ_LDS_BANK_CONFLICT = 0).
template <int DT> struct Pipe {
    using M = MM<DT>;
    using T = Tile<DT>;
    int frag_off[M::NREG];
    int st_off[T::STAGE];
    int voff[T::STAGE];  // byte offsets of this thread's staged chunks inside a tile's source rows
    int srow;            // first staged row of this thread inside the tile
    __device__ __forceinline__ void init(int tid, int lane) {
        const int row = lane & 31, hf = lane >> 5;
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) frag_off[r] = row * T::ROWB + ((M::chunk_of(r, hf) ^ (row & 15)) * 16);
        srow = tid / M::CHUNKS;
        const int ch = tid % M::CHUNKS;
#pragma unroll
        for (int u = 0; u < T::STAGE; ++u) {
            const int rr = srow + u * T::ROWS_PER_STEP;
            st_off[u] = rr * T::ROWB + ((ch ^ (rr & 15)) * 16);
            voff[u] = rr * T::ROWB + ch * 16;
        }
    }
    // global -> registers (issued early, consumed late) through a buffer descriptor: the per-thread byte
    // offsets are precomputed once, the tile offset travels in an SGPR, so a tile costs STAGE
    // buffer_load_dwordx4 and no address arithmetic.  Branch-free: rows past the end of the matrix are out
    // of the descriptor's range and read as zeros; rows past the caller's valid range hold finite filler
    // whose logits the callers mask out.  (A load inside a conditional would force vmcnt(0) at the join.)
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, int first_row, u32x4* st) const {
        const int soff = first_row * T::ROWB;
#pragma unroll
        for (int u = 0; u < T::STAGE; ++u)
            st[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[u], soff, 0));
    }
    __device__ __forceinline__ void store(char* lds_tile, const u32x4* st) const {
#pragma unroll
        for (int u = 0; u < T::STAGE; ++u) *(u32x4*)(lds_tile + st_off[u]) = st[u];
    }
    // explicit two-step form: fetch all A fragments of a block, then run the MFMAs on them
    __device__ __forceinline__ void read_frags(u32x4* a, const char* lds_tile, int blk) const {
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) a[r] = *(const u32x4*)(lds_tile + blk * 32 * T::ROWB + frag_off[r]);
    }
    __device__ __forceinline__ void mma_frags(f32x16& acc, const u32x4* a, const u32x4* rf) const {
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) M::mma(acc, a[r], rf[r]);
    }
    // acc += A(32 LDS rows of block `blk`) x B(register fragment)
    __device__ __forceinline__ void mma_block(f32x16& acc, const char* lds_tile, int blk, const u32x4* rf) const {
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) {
            const u32x4 a = *(const u32x4*)(lds_tile + blk * 32 * T::ROWB + frag_off[r]);
            M::mma(acc, a, rf[r]);
        }
    }
};

__device__ __forceinline__ float max16(const f32x16& a) {
    return fmaxf(fmaxf(fmaxf(fmaxf(a[0], a[1]), a[2]), fmaxf(fmaxf(a[3], a[4]), a[5])),
                 fmaxf(fmaxf(fmaxf(fmaxf(a[6], a[7]), a[8]), fmaxf(fmaxf(a[9], a[10]), a[11])),
                       fmaxf(fmaxf(fmaxf(a[12], a[13]), a[14]), a[15])));
}

// ------------------------------------------------------------------------------------------------
// pass 1: partial row log-sum-exp over one key split
//   lse_part[ks,h,i] = log sum_{j in split ks} exp(q_hi . k_gj / sqrt(D))   (log2 domain for bf16)
// grid (ceil(L/128), Hq, KS), 256 threads; wave w keeps query rows i0 + 32w + (lane&31) in registers.
// ------------------------------------------------------------------------------------------------
template <int DT>
struct RowStat {  // online max / sum of one query row, over the keys this lane sees
    float m, sum;  // bf16: m = raw dot-product max, sum of exp2((x - m) * c2);  fp32: m = max logit, sum of exp(x - m)
    __device__ __forceinline__ void init() { m = -INFINITY; sum = 0.f; }
    template <bool RAGGED>
    __device__ __forceinline__ void update(f32x16& a0, f32x16& a1, int j0, int j_end, int hf, float c2, float sqrt_d) {
        if (DT == RTK_F32) {  // the reference's operation order: logits = dot / sqrt(D), natural exp
#pragma unroll
            for (int r = 0; r < 16; ++r) { a0[r] = __fdiv_rn(a0[r], sqrt_d); a1[r] = __fdiv_rn(a1[r], sqrt_d); }
        }
        if (RAGGED) {  // keys >= j_end do not exist
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (j0 + acc_row(r, hf) >= j_end) a0[r] = -INFINITY;
                if (j0 + 32 + acc_row(r, hf) >= j_end) a1[r] = -INFINITY;
            }
        }
        const float mn = fmaxf(m, fmaxf(max16(a0), max16(a1)));
        if (RAGGED && mn == -INFINITY) return;
        float add = 0.f;
        if (DT == RTK_BF16) {
            const float nb = -mn * c2;
#pragma unroll
            for (int r = 0; r < 16; ++r) add += __builtin_amdgcn_exp2f(fmaf(a0[r], c2, nb));
#pragma unroll
            for (int r = 0; r < 16; ++r) add += __builtin_amdgcn_exp2f(fmaf(a1[r], c2, nb));
            sum = sum * __builtin_amdgcn_exp2f((m - mn) * c2) + add;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) add += expf(a0[r] - mn);
#pragma unroll
            for (int r = 0; r < 16; ++r) add += expf(a1[r] - mn);
            sum = sum * expf(m - mn) + add;
        }
        m = mn;
    }
    // merge with the other half-wave (disjoint key subsets of the same row) and take the log
    __device__ __forceinline__ float finish(float c2) const {
        const float m2 = __shfl_xor(m, 32, WAVE), s2 = __shfl_xor(sum, 32, WAVE);
        const float mm = fmaxf(m, m2);
        if (mm == -INFINITY) return -INFINITY;  // no key seen (cannot happen for a non-empty split)
        if (DT == RTK_BF16) {
            const float tot = sum * __builtin_amdgcn_exp2f((m - mm) * c2) + s2 * __builtin_amdgcn_exp2f((m2 - mm) * c2);
            return mm * c2 + __builtin_amdgcn_logf(tot);  // v_log_f32 = log2
        }
        const float tot = sum * expf(m - mm) + s2 * expf(m2 - mm);
        return mm + logf(tot);
    }
};

// Shape of the kernels (HISTORY.md §4 has the same-box A/B numbers behind every choice):
//   fp32 register-staged kernels: one 32-row register block per wave, loads one tile ahead (RegBlocks below; two
//       blocks need ~250 VGPRs and a second staging set bought nothing: the kernels are bound by instruction issue).
//   bf16 LDS-DMA kernels: two 32-row register blocks per wave (every A fragment read from LDS feeds two MFMAs on
//       independent accumulators: half the fragment reads, DMA issues, barriers and waits per MFMA; ~160 VGPRs -> 3 waves
//       per SIMD), lazy max in pass 1, the next tile's DMA pieces issued inside block 0's softmax, all fragment reads of
//       a block ahead of its MFMAs which alternate strictly between the two accumulators, a last tile that is at most
//       half full on the one-block body, pass 2's normalisers by LDS-DMA from wave 0.
// The only compile-time knobs left are the ones variants.h lists (A/B builds; production never defines them).
template <int DT> struct RegBlocks {
    static constexpr int NB = 1;
    static constexpr int PF = 1;
};

template <int DT, int NB>
__global__ __launch_bounds__(SC_BLOCK) void score_pass1_kernel(const char* __restrict__ q, const char* __restrict__ k,
                                                               int Hq, int Hkv, int L, int keys_per_split,
                                                               int row_tiles, int xcd_remap,
                                                               float* __restrict__ lse_part) {
    using M = MM<DT>;
    using T = Tile<DT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE, hf = lane >> 5;
    // XCD-aware decode of a 1-D grid (block b runs on XCD b % 8): all workgroups that stream the same
    // key split of the same KV group share an XCD, so the split stays resident in that XCD's L2.
    const int G = Hq / Hkv;
    int bx, h, ks;
    {
        const int per_group = row_tiles * G;                 // workgroups sharing one (g, ks) key stream
        int grp, w;
        if (xcd_remap) {  // only when the group count is a multiple of 8 (balanced XCDs)
            const int xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
            grp = xcd + NXCD * (slot / per_group);
            w = slot % per_group;
        } else {
            grp = blockIdx.x / per_group;
            w = blockIdx.x % per_group;
        }
        ks = grp / Hkv;
        h = (grp % Hkv) * G + w / row_tiles;
        bx = w % row_tiles;
    }
    const int g = h / G;
    const int i0 = bx * (REG_ROWS * NB) + wid * (32 * NB);   // this wave's NB*32 query rows
    const int jb = ks * keys_per_split, je = min(L, jb + keys_per_split);
    const char* qh = q + (size_t)h * L * HD * M::ESIZE;
    Pipe<DT> pp;
    pp.init(tid, lane);
    u32x4 qf[NB][M::NREG];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) load_reg_frag<DT>(qh, i0 + 32 * nb, L, lane, qf[nb]);

    const int nkeys = je - jb;
    const int nfull = nkeys / TILE_ROWS;              // full tiles
    const int ntiles = (nkeys + TILE_ROWS - 1) / TILE_ROWS;
    const float sqrt_d = sqrtf((float)HD);
    const float c2 = 1.4426950408889634f / sqrt_d;    // bf16: log2(e)/sqrt(D) folded into the exp2 argument
    RowStat<DT> rs[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) rs[nb].init();

    constexpr int PF = RegBlocks<DT>::PF;
    u32x4 stA[T::STAGE], stB[T::STAGE];   // staging registers: set A holds even tiles, set B odd tiles (PF == 2)
    // descriptor over this KV group's [L, 128] key matrix (wave-uniform: kernel arguments and blockIdx only)
    const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(k + (size_t)g * L * HD * M::ESIZE), 0, L * HD * M::ESIZE, 0x00020000);
#define RTK_LOAD_TILE(t, dst) pp.load(krsrc, jb + (t) * TILE_ROWS, dst)
    RTK_LOAD_TILE(0, stA);
    pp.store(smem, stA);
    if (PF == 2 && ntiles > 1) RTK_LOAD_TILE(1, stB);
    __syncthreads();

    // Every A fragment read from LDS feeds NB MFMAs (one per register block).  The MFMA -> softmax
    // dependency is hidden by the other waves on the SIMD (an explicit in-wave pipeline as in pass 2
    // measured equal here).  Tile jt is computed from LDS buffer jt & 1 while the loads of tile jt + PF are
    // in flight; tile jt + 1 (loaded one step earlier when PF == 2) is written to the other buffer.
    // ISSUE / STORE are compile-time in the steady-state loop: no load sits inside a conditional there.
#define RTK_STEP1(JT, PAR, ISSUE, STORE, MAYRAG) \
    { \
        constexpr int par = PAR; \
        const char* cur = smem + par * T::BYTES; \
        char* nxt = smem + (par ^ 1) * T::BYTES; \
        if constexpr (ISSUE) { \
            if constexpr (PF == 2 && par == 1) RTK_LOAD_TILE((JT) + PF, stB); \
            else RTK_LOAD_TILE((JT) + PF, stA); \
        } \
        f32x16 acc0[NB], acc1[NB]; \
        { \
            u32x4 a[M::NREG]; \
            pp.read_frags(a, cur, 0); \
_Pragma("unroll") \
            for (int nb = 0; nb < NB; ++nb) { acc0[nb] = f32x16{0}; pp.mma_frags(acc0[nb], a, qf[nb]); } \
            pp.read_frags(a, cur, 1); \
_Pragma("unroll") \
            for (int nb = 0; nb < NB; ++nb) { acc1[nb] = f32x16{0}; pp.mma_frags(acc1[nb], a, qf[nb]); } \
        } \
_Pragma("unroll") \
        for (int nb = 0; nb < NB; ++nb) { \
            /* steady state: tiles are full by construction (a run-time test here gets if-converted into 64 */ \
            /* v_cmp + v_cndmask per tile: 40 % more VALU issue in a kernel that is issue bound)            */ \
            if (!(MAYRAG) || (JT) < nfull) rs[nb].template update<false>(acc0[nb], acc1[nb], 0, 0, hf, c2, sqrt_d); \
            else rs[nb].template update<true>(acc0[nb], acc1[nb], (JT) * TILE_ROWS, nkeys, hf, c2, sqrt_d); \
        } \
        if constexpr (STORE) { \
            if constexpr (PF == 2 && par == 0) pp.store(nxt, stB); \
            else pp.store(nxt, stA); \
        } \
        __syncthreads(); \
    }
    int jt = 0;
    for (; jt + PF + 1 < ntiles; jt += 2) {  // steady state, two tiles per trip (parities are constants)
        RTK_STEP1(jt, 0, true, true, false)      // jt + 2 < ntiles here, and only the last tile can be ragged
        RTK_STEP1(jt + 1, 1, true, true, false)
    }
    // tail: at most PF + 1 tiles; jt is even here, so the parities are known statically (a run-time parity
    // would make the compiler select between the two staging sets through memory)
#define RTK_TAIL(PAR)                                              \
    if (jt < ntiles) {                                             \
        if (jt + PF < ntiles) RTK_STEP1(jt, PAR, true, true, true)       \
        else if (jt + 1 < ntiles) RTK_STEP1(jt, PAR, false, true, true)  \
        else RTK_STEP1(jt, PAR, false, false, true)                      \
        ++jt;                                                      \
    }
    RTK_TAIL(0)
    RTK_TAIL(1)
    RTK_TAIL(0)
#undef RTK_TAIL
#undef RTK_STEP1
#undef RTK_LOAD_TILE
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float out = rs[nb].finish(c2);
        const int i = i0 + 32 * nb + (lane & 31);
        if (hf == 0 && i < L) lse_part[((size_t)ks * Hq + h) * L + i] = out;
    }
}

// online max / sum of one query row, one 32-key block at a time (bf16 path, log2 domain)
struct RowStatB {  // online max / sum of one query row over the keys this lane sees (bf16 path, log2 domain)
    float m, sum, off;   // off = -m * c2, the exponent offset of the lazy form (kept so that a block does not recompute it)
    __device__ __forceinline__ void init() { m = -INFINITY; sum = 0.f; off = INFINITY; }
    template <bool RAGGED>
    __device__ __forceinline__ void update(f32x16& a, int j0, int j_end, int hf, float c2) {
        if (RAGGED) {  // keys >= j_end do not exist
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (j0 + acc_row(r, hf) >= j_end) a[r] = -INFINITY;
        }
        const float mn = fmaxf(m, max16(a));
        if (RAGGED && mn == -INFINITY) return;
        const float nb = -mn * c2;
        float add = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) add += __builtin_amdgcn_exp2f(fmaf(a[r], c2, nb));
        sum = sum * __builtin_amdgcn_exp2f((m - mn) * c2) + add;
        m = mn;
        off = nb;
    }
    // Lazy form: the 16 exponentials are taken against the offset of the LAST rescale (no max over the block, no
    // rescale of the running sum); only when some lane's block sum is not a finite number below 2^96 - a key beat the
    // stale offset by ~96 binary orders, or nothing has been seen yet (m = -inf makes the offset +inf) - the whole wave
    // takes the ordinary online step for this block.  Any offset gives the same sum mathematically and fp32 keeps its
    // relative precision over that range, so the result is as exact as the eager form (not bitwise equal to it).
    // Saves ~13 of the ~64 VALU instructions per 32 x 32 block in a kernel that is bound by instruction issue.
    template <bool RAGGED>
    __device__ __forceinline__ void update_lazy(f32x16& a, int j0, int j_end, int hf, float c2) {
        if (RAGGED) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (j0 + acc_row(r, hf) >= j_end) a[r] = -INFINITY;
        }
        float add = __builtin_amdgcn_exp2f(fmaf(a[0], c2, off));   // (0 + e0 would cost an instruction: -0 semantics)
#pragma unroll
        for (int r = 1; r < 16; ++r) add += __builtin_amdgcn_exp2f(fmaf(a[r], c2, off));
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(add < 0x1p96f)) == 0, 1)) {
            sum += add;
            return;
        }
        const float mn = fmaxf(m, max16(a));
        if (mn == -INFINITY) return;   // no key seen yet on this lane and none in this block
        const float nb2 = -mn * c2;
        add = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) add += __builtin_amdgcn_exp2f(fmaf(a[r], c2, nb2));
        sum = sum * __builtin_amdgcn_exp2f((m - mn) * c2) + add;
        m = mn;
        off = nb2;
    }
    __device__ __forceinline__ float finish(float c2) const {
        const float m2 = __shfl_xor(m, 32, WAVE), s2 = __shfl_xor(sum, 32, WAVE);
        const float mm = fmaxf(m, m2);
        if (mm == -INFINITY) return -INFINITY;
        const float tot = sum * __builtin_amdgcn_exp2f((m - mm) * c2) + s2 * __builtin_amdgcn_exp2f((m2 - mm) * c2);
        return mm * c2 + __builtin_amdgcn_logf(tot);  // v_log_f32 = log2
    }
};

// D = A x B + C on the bf16 (exact mode) or fp16 (RTK_BF16_FAST) matrix instruction; D and C may be different registers
template <bool FAST>
__device__ __forceinline__ void mma16(f32x16& d, const u32x4& a, const u32x4& b, const f32x16& c) {
    if constexpr (FAST)
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// RTK_BF16_FAST pass 1: the accumulators ARE base-2 logits (q~ was pre-scaled), so a logit needs no multiply.
// RowStatR - the production form - adds exp2(logit) to the row sum with NO offset and no test: two instructions per
// logit, the minimum.  That is exact whenever the row's sum stays inside fp32's comfortable range; a row whose sum
// left it (a logit beyond ~2^7 in base 2 -> inf, or every logit below ~-60 -> precision lost in subnormals) is
// detected ONCE, at the end - inf and NaN are sticky in a sum of non-negative terms - and published as NaN; the
// fix-up launch that follows (score_pass1_fixup_kernel) recomputes exactly the row tiles that own a NaN with
// RowStatF, the offset-carrying form.  Deterministic: which rows take which path depends on the data only.
struct RowStatR {
    float sum;
    __device__ __forceinline__ void init() { sum = 0.f; }
    template <bool RAGGED>
    __device__ __forceinline__ void update(f32x16& a, int j0, int j_end, int hf) {
        if (RAGGED) {  // keys >= j_end do not exist
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (j0 + acc_row(r, hf) >= j_end) a[r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f(a[r]);
    }
    __device__ __forceinline__ float finish() const {
        const float tot = sum + __shfl_xor(sum, 32, WAVE);
        const bool fine = tot < 0x1p120f && tot > 0x1p-60f;      // false for inf and NaN as well
        return fine ? __builtin_amdgcn_logf(tot) : __builtin_nanf("");   // v_log_f32 = log2
    }
};

// The same for the exact modes (bf16 / fp16 payloads, un-scaled operands): exp2(dot * c2) added to the row sum, no offset,
// no per-block test - one multiply more than RowStatR, but none of RowStatB's bookkeeping (the lazy test costs a compare,
// a ballot and a branch per 32-key block, and the offset a register): same end-of-row check, same fix-up launch, which
// then runs RowStatB.
struct RowStatRX {
    float sum;
    __device__ __forceinline__ void init() { sum = 0.f; }
    template <bool RAGGED>
    __device__ __forceinline__ void update(f32x16& a, int j0, int j_end, int hf, float c2) {
        if (RAGGED) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (j0 + acc_row(r, hf) >= j_end) a[r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f(a[r] * c2);
    }
    __device__ __forceinline__ float finish() const {
        const float tot = sum + __shfl_xor(sum, 32, WAVE);
        const bool fine = tot < 0x1p120f && tot > 0x1p-60f;
        return fine ? __builtin_amdgcn_logf(tot) : __builtin_nanf("");
    }
};

// The robust form (fix-up launch only): sum = sum_j exp2(s_j + off) over the keys this lane has seen, i.e. the true
// total is sum * 2^-off.  Lazy like RowStatB: the offset is that of the last rescale; when some lane's block sum is
// not a finite number below 2^96, or nothing has been seen yet (`primed`, wave-uniform), the wave re-bases on the
// block's maximum.
struct RowStatF {
    float sum, off;
    __device__ __forceinline__ void init() { sum = 0.f; off = 0.f; }
    template <bool RAGGED>
    __device__ __forceinline__ void update(f32x16& a, int j0, int j_end, int hf, bool& primed) {
        if (RAGGED) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (j0 + acc_row(r, hf) >= j_end) a[r] = -INFINITY;
        }
        if (primed) {
            float add = __builtin_amdgcn_exp2f(a[0] + off);
#pragma unroll
            for (int r = 1; r < 16; ++r) add += __builtin_amdgcn_exp2f(a[r] + off);
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(add < 0x1p96f)) == 0, 1)) {
                sum += add;
                return;
            }
        }
        const float mx = max16(a);
        // a lane without a key in the block (ragged tail: mx = -inf) keeps its offset: add = 0, sum unchanged
        const float noff = (mx == -INFINITY) ? off : (primed ? -fmaxf(mx, -off) : -mx);
        float add = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) add += __builtin_amdgcn_exp2f(a[r] + noff);
        // the first re-base starts from sum = 0 with a meaningless offset (0 * 2^(noff - 0) may be 0 * inf); later ones
        // only ever lower the offset (noff <= off), so the rescale factor is <= 1
        sum = primed ? sum * __builtin_amdgcn_exp2f(noff - off) + add : add;
        off = noff;
        primed = true;
    }
    __device__ __forceinline__ float finish() const {
        const float m1 = sum > 0.f ? -off : -INFINITY;   // a half that saw no key (ragged tail) carries no scale
        const float m2 = __shfl_xor(m1, 32, WAVE), s2 = __shfl_xor(sum, 32, WAVE);
        const float mm = fmaxf(m1, m2);
        if (mm == -INFINITY) return -INFINITY;            // an empty row
        const float tot = sum * __builtin_amdgcn_exp2f(m1 - mm) + s2 * __builtin_amdgcn_exp2f(m2 - mm);
        return mm + __builtin_amdgcn_logf(tot);
    }
};

// ------------------------------------------------------------------------------------------------
// pass 2: partial[g,split,j] = sum_{h in g} sum_{i in split} exp(s_hij - lse[h,i])
// grid (ceil(L/128), Hkv, RS); wave w keeps keys j0 + 32w + (lane&31) in registers.
// lse[h,i] is combined on the fly from pass 1's KS partials while the query tile is staged.
// ------------------------------------------------------------------------------------------------
// lse[h,i] = log sum_ks exp(lse_part[ks,h,i]), written over split 0 (one thread per row: no hazard)
template <int DT>
__global__ __launch_bounds__(256) void lse_combine_kernel(float* __restrict__ lse_part, size_t n, int KS,
                                                          size_t unit_floats, int negate = 0) {
    lse_part += blockIdx.y * unit_floats;   // blockIdx.y = unit of a batched launch
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    float v[8];
    float mx = -INFINITY;
    for (int s = 0; s < KS; ++s) {
        v[s] = lse_part[(size_t)s * n + idx];
        mx = fmaxf(mx, v[s]);
    }
    float tot = 0.f;
    for (int s = 0; s < KS; ++s) tot += (DT == RTK_BF16) ? __builtin_amdgcn_exp2f(v[s] - mx) : expf(v[s] - mx);
    const float out = mx + ((DT == RTK_BF16) ? __builtin_amdgcn_logf(tot) : logf(tot));
    lse_part[idx] = negate ? -out : out;   // RTK_BF16_FAST: pass 2 starts its accumulators from -lse
}

// col += sum_r exp(acc[r]*scale - ls[r]) for one 32x32 block (16 values per lane)
template <int DT>
__device__ __forceinline__ void colsum_block(float& col, const f32x16& acc, const float* ls, float c2, float sqrt_d) {
    // scalar fma / exp2 / add per logit: v_pk_fma_f32 / v_pk_add_f32 were measured 5-7 % SLOWER here
    // (packed f32 ops cost extra issue slots beside MFMAs)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (DT == RTK_BF16) col += __builtin_amdgcn_exp2f(fmaf(acc[r], c2, -ls[r]));
        else col += expf(__fdiv_rn(acc[r], sqrt_d) - ls[r]);
    }
}
// this lane's 16 row normalisers of block `blk` (rows (r&3) + 8*(r>>2) + 4*hf)
__device__ __forceinline__ void load_ls(float* ls, const float* lcur, int blk, int hf) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) *(float4*)(ls + 4 * r4) = *(const float4*)(lcur + blk * 32 + 8 * r4 + 4 * hf);
}

template <int DT, int NB>
__global__ __launch_bounds__(SC_BLOCK) void score_pass2_kernel(const char* __restrict__ q, const char* __restrict__ k,
                                                               const float* __restrict__ lse, int Hq, int Hkv, int L,
                                                               int rows_per_split, int col_tiles, int RS,
                                                               int xcd_remap, float* __restrict__ partial,
                                                               const int* __restrict__ key_index = nullptr) {
    using M = MM<DT>;
    using T = Tile<DT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lse_s = (float*)(smem + 2 * T::BYTES);  // [2][TILE_ROWS]
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE, hf = lane >> 5;
    // live keys (key_compact_kernel, see score_pass2_dma_kernel): list positions name the k~ row and the output column
    const int* kidx = nullptr;
    int Lk = L;
    if (key_index && key_index[L] >= 0) {
        kidx = key_index;
        Lk = key_index[L];
    }
    // XCD-aware decode (block b runs on XCD b % 8): the col_tiles workgroups that stream the same query
    // rows (same KV group, same row split) share an XCD and therefore its L2.
    const int G = Hq / Hkv;
    int bx, g, rs;
    {
        int grp;
        if (xcd_remap) {
            const int xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
            grp = xcd + NXCD * (slot / col_tiles);           // (g, rs) pair index
            bx = slot % col_tiles;
        } else {
            grp = blockIdx.x / col_tiles;
            bx = blockIdx.x % col_tiles;
        }
        g = grp % Hkv;
        rs = grp / Hkv;
    }
    const int j0 = bx * (REG_ROWS * NB) + wid * (32 * NB);   // this wave's NB*32 keys (positions in the live list)
    if (bx * (REG_ROWS * NB) >= Lk) return;                   // uniform per workgroup
    const char* kg = k + (size_t)g * L * HD * M::ESIZE;
    const int ib = rs * rows_per_split, ie = min(L, ib + rows_per_split);
    const int nrows = ie - ib;
    const int tiles_per_head = (nrows + TILE_ROWS - 1) / TILE_ROWS;   // >= 1: empty splits are not launched
    const int ntiles = tiles_per_head * G;

    Pipe<DT> pp;
    pp.init(tid, lane);
    u32x4 kf[NB][M::NREG];
    int jcol[NB];   // token index of this lane's key per register block, -1 past the list
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int jp = j0 + 32 * nb + (lane & 31);
        jcol[nb] = jp < Lk ? (kidx ? kidx[jp] : jp) : -1;
        const u32x4* p = (const u32x4*)(kg + (size_t)max(jcol[nb], 0) * HD * M::ESIZE);
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) kf[nb][r] = jcol[nb] >= 0 ? p[M::chunk_of(r, hf)] : u32x4{0, 0, 0, 0};
    }

    const float sqrt_d = sqrtf((float)HD);
    const float c2 = 1.4426950408889634f / sqrt_d;
    float col[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) col[nb] = 0.f;
    constexpr int PF = RegBlocks<DT>::PF;
    u32x4 stA[T::STAGE], stB[T::STAGE];   // staging register sets: even / odd tiles (only A when PF == 1)
    float lstA = 0.f, lstB = 0.f;

    // cursor of the tile being prefetched: row tile inside the split, source pointers of the current head
    int nt = 0;
    const int last_row = Hq * L - 1;  // last row of the whole q~ buffer
    const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)q, 0, Hq * L * HD * M::ESIZE, 0x00020000);
    int nrow0 = (g * G) * L + ib;     // first buffer row of the cursor head's split
    // loads the cursor tile (+ this thread's lse element) into one staging set, then advances the cursor;
    // rows past the split end get lse = +inf: exp(s - inf) = 0 whatever filler the tile holds
#define RTK_ISSUE(st, lst)                                                              \
    {                                                                                   \
        pp.load(qrsrc, nrow0 + nt * TILE_ROWS, st);                                     \
        const int r__ = nt * TILE_ROWS + (tid & (TILE_ROWS - 1));                       \
        lst = (r__ < nrows) ? lse[min(nrow0 + r__, last_row)] : INFINITY;               \
        const bool wrap__ = (nt + 1 == tiles_per_head);                                 \
        nt = wrap__ ? 0 : nt + 1;                                                       \
        nrow0 += wrap__ ? L : 0;                                                        \
    }
    // Software pipeline inside the wave: while the matrix pipe runs the 8*NB MFMAs of one 32-row block,
    // the VALU finishes the previous block (fma + exp2 + add per value; the row normalisers are shared by
    // the NB register blocks).  `pend` / `pls` carry a tile's second block across the barrier into the
    // next tile's first MFMA group.
    f32x16 pend[NB];
    float pls[16];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) pend[nb] = f32x16{0};
#pragma unroll
    for (int r = 0; r < 16; ++r) pls[r] = INFINITY;  // exp(-inf) = 0: nothing pending yet
#define RTK_STEP2(BUF, ISSUE, STORE) \
    { \
        constexpr int buf = BUF; \
        const char* cur = smem + buf * T::BYTES; \
        const float* lcur = lse_s + buf * TILE_ROWS; \
        if constexpr (ISSUE) { \
            if constexpr (PF == 2 && buf == 1) RTK_ISSUE(stB, lstB) \
            else RTK_ISSUE(stA, lstA) \
        } \
        u32x4 a0[M::NREG], a1[M::NREG]; \
        pp.read_frags(a0, cur, 0); \
        pp.read_frags(a1, cur, 1); \
        float ls0[16], ls1[16]; \
        load_ls(ls0, lcur, 0, hf); \
        load_ls(ls1, lcur, 1, hf); \
        __builtin_amdgcn_sched_barrier(0); \
        f32x16 acc0[NB]; \
_Pragma("unroll") \
        for (int nb = 0; nb < NB; ++nb) { \
            acc0[nb] = f32x16{0}; \
            pp.mma_frags(acc0[nb], a0, kf[nb]); \
            colsum_block<DT>(col[nb], pend[nb], pls, c2, sqrt_d); \
            asm volatile("" : "+v"(acc0[nb]), "+v"(col[nb])); \
        } \
        if (DT == RTK_BF16) { \
_Pragma("unroll") \
            for (int i = 0; i < 8 * NB; ++i) { \
                __builtin_amdgcn_sched_group_barrier(SGB_MFMA, 1, 0); \
                __builtin_amdgcn_sched_group_barrier(SGB_VALU, 4, 0); \
                __builtin_amdgcn_sched_group_barrier(SGB_TRANS, 2, 0); \
            } \
        } \
        __builtin_amdgcn_sched_barrier(0); \
_Pragma("unroll") \
        for (int nb = 0; nb < NB; ++nb) { \
            pend[nb] = f32x16{0}; \
            pp.mma_frags(pend[nb], a1, kf[nb]); \
            colsum_block<DT>(col[nb], acc0[nb], ls0, c2, sqrt_d); \
            asm volatile("" : "+v"(pend[nb]), "+v"(col[nb])); \
        } \
        if (DT == RTK_BF16) { \
_Pragma("unroll") \
            for (int i = 0; i < 8 * NB; ++i) { \
                __builtin_amdgcn_sched_group_barrier(SGB_MFMA, 1, 1); \
                __builtin_amdgcn_sched_group_barrier(SGB_VALU, 4, 1); \
                __builtin_amdgcn_sched_group_barrier(SGB_TRANS, 2, 1); \
            } \
        } \
        __builtin_amdgcn_sched_barrier(0); \
_Pragma("unroll") \
        for (int r = 0; r < 16; ++r) pls[r] = ls1[r]; \
        if constexpr (STORE) { \
            if constexpr (PF == 2 && buf == 0) { \
                pp.store(smem + (buf ^ 1) * T::BYTES, stB); \
                if (tid < TILE_ROWS) lse_s[(buf ^ 1) * TILE_ROWS + tid] = lstB; \
            } else { \
                pp.store(smem + (buf ^ 1) * T::BYTES, stA); \
                if (tid < TILE_ROWS) lse_s[(buf ^ 1) * TILE_ROWS + tid] = lstA; \
            } \
        } \
        __syncthreads(); \
    }
    RTK_ISSUE(stA, lstA)
    pp.store(smem, stA);
    if (tid < TILE_ROWS) lse_s[tid] = lstA;
    if (PF == 2 && ntiles > 1) RTK_ISSUE(stB, lstB)
    __syncthreads();
    int it = 0;
    for (; it + PF + 1 < ntiles; it += 2) {  // steady state: loads are unconditional => counted vmcnt waits
        RTK_STEP2(0, true, true)
        RTK_STEP2(1, true, true)
    }
    // tail: at most PF + 1 tiles; `it` is even here, so the parities are static
#define RTK_TAIL(PAR)                                        \
    if (it < ntiles) {                                       \
        if (it + PF < ntiles) RTK_STEP2(PAR, true, true)     \
        else if (it + 1 < ntiles) RTK_STEP2(PAR, false, true) \
        else RTK_STEP2(PAR, false, false)                    \
        ++it;                                                \
    }
    RTK_TAIL(0)
    RTK_TAIL(1)
    RTK_TAIL(0)
#undef RTK_TAIL
#undef RTK_STEP2
#undef RTK_ISSUE
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        colsum_block<DT>(col[nb], pend[nb], pls, c2, sqrt_d);  // drain the pipeline
        col[nb] += __shfl_xor(col[nb], 32, WAVE);
        if (hf == 0 && jcol[nb] >= 0) partial[((size_t)g * RS + rs) * L + jcol[nb]] = col[nb];
    }
}

// ------------------------------------------------------------------------------------------------
// Live keys of pass 2.  The reference overwrites the score of every key-patch token with 1.0 after the scoring
// (`score.masked_fill_(keypatches_mask_chunk, 1.0)`, longvideo_cache.py:272-274): the column masses of those tokens are
// computed and thrown away.  A key's column mass depends on its own column only, so pass 2 - whose REGISTER operand is
// the keys - can run on the compacted list of unmasked keys and leave the masked columns unwritten: identical bits for
// every column anybody reads, (mask rate) x pass 2 less work (the mask is DPSelect's peak flag: about a third of the
// tokens).  Pass 1 is untouched: the row normalisers are sums over ALL keys.
//   key_index[unit][0 .. n)  ascending indices of the unit's unmasked tokens, key_index[unit][L] = n  (-1: no mask, identity)
// One 1024-thread workgroup per unit: ordered compaction by a block scan of per-thread counts.
// ------------------------------------------------------------------------------------------------
// Where the queries of the units of a batched launch live.  row_pitch == 0: the packed un-rotated copies inside the
// units' score workspaces ([Hq, L, 128] at q + unit * q_unit_bytes).  Else: per-unit base pointers of tensors the caller
// keeps alive - the pre-RoPE projections themselves (the prologue route scores q0 as it is, so no copy is made),
// element (h, i, :) at unit[u] + h * head_stride + i * row_pitch bytes.
constexpr int MAX_Q_UNITS = 32;
struct QView {
    const char* unit[MAX_Q_UNITS];
    int head_stride, row_pitch;
};
constexpr int MAX_MASK_UNITS = 64;
struct KeyMasks {
    const uint8_t* m[MAX_MASK_UNITS];
};
__global__ __launch_bounds__(1024) void key_compact_kernel(KeyMasks masks, int L, int* __restrict__ key_index) {
    __shared__ int wsum[16];
    const uint8_t* __restrict__ mk = masks.m[blockIdx.x];
    int* __restrict__ out = key_index + (size_t)blockIdx.x * (L + 1);
    if (!mk) {
        if (threadIdx.x == 0) out[L] = -1;
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (L + 1023) / 1024;
    const int b = tid * per, e = min(L, b + per);
    int cnt = 0;
    for (int j = b; j < e; ++j) cnt += mk[j] == 0;
    int inc = cnt;   // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, WAVE);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int v = wsum[w];
        base += w < wv ? v : 0;
        total += v;
    }
    int at = base + inc - cnt;
    for (int j = b; j < e; ++j)
        if (mk[j] == 0) out[at++] = j;
    if (tid == 0) out[L] = total;
}

// ------------------------------------------------------------------------------------------------
// pass 2, LDS-DMA form (bf16, the production kernel): the decomposition of score_pass2_kernel without its in-wave
// software pipeline (one 32-row block of logits live at a time: ~95 VGPRs -> 4 waves per SIMD), and the streamed
// query tile goes HBM/L2 -> LDS directly (buffer_load_dwordx4 ... lds): no staging registers, no ds_write pass.  A wave's DMA instruction fills
// 1 KiB of LDS linearly (lane * 16 B), so the XOR swizzle of the tile is applied to the SOURCE address:
// LDS position p of row r receives chunk p ^ (r & 15), the same involution the fragment reads apply.
// NB = 32-key register blocks per wave (NB = 2: every A fragment read from LDS feeds two MFMAs).
// ------------------------------------------------------------------------------------------------
// The work of one workgroup: NB x 32 keys per wave starting at key j_base + wid * 32 * NB, the query rows of split rs.
template <int NB, bool FAST = false, bool F16 = false>   // F16: exact softmax on fp16 payloads (RTK_F16)
__device__ __forceinline__ void score_pass2_dma_body(const char* __restrict__ q, const char* __restrict__ k,
                                                     const float* __restrict__ lse, int Hq, int Hkv, int L,
                                                     int rows_per_split, int RS, float* __restrict__ partial, int j_base,
                                                     int g, int rs, const int* __restrict__ kidx, int Lk, int q_hs,
                                                     int q_pitch) {
    // q_hs / q_pitch: bytes between the heads / rows of q (packed copy: L * 256 and 256)
    // kidx / Lk: the unit's live keys (ascending token indices, Lk of them; kidx == NULL: all L tokens, Lk == L).  j_base
    // and the wave's key offsets count positions of THAT list; a position's token index names the k~ row it loads and
    // the column of `partial` it writes.
    constexpr int DT = RTK_BF16;
    using M = MM<DT>;
    using T = Tile<DT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lse_s = (float*)(smem + 2 * T::BYTES);  // [2][TILE_ROWS]
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), hf = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int G = Hq / Hkv;
    const int j0 = j_base + wid * (32 * NB);
    const char* kg = k + (size_t)g * L * HD * M::ESIZE;
    const int ib = rs * rows_per_split, ie = min(L, ib + rows_per_split);
    const int nrows = ie - ib;
    const int tiles_per_head = (nrows + TILE_ROWS - 1) / TILE_ROWS;
    const int ntiles = tiles_per_head * G;

    int frag_off[M::NREG];
    {
        const int row = lane & 31;
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) frag_off[r] = row * T::ROWB + ((M::chunk_of(r, hf) ^ (row & 15)) * 16);
    }
    u32x4 kf[NB][M::NREG];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int jp = j0 + 32 * nb + (lane & 31);           // position in the live-key list
        const bool ok = jp < Lk;
        const int row = ok ? (kidx ? kidx[jp] : jp) : 0;      // token index = k~ row
        const u32x4* p = (const u32x4*)(kg + (size_t)row * HD * M::ESIZE);
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) kf[nb][r] = ok ? p[M::chunk_of(r, hf)] : u32x4{0, 0, 0, 0};
    }
    const float sqrt_d = sqrtf((float)HD);
    const float c2 = 1.4426950408889634f / sqrt_d;
    float col[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) col[nb] = 0.f;

    // DMA addressing: piece P = 4u + wid (u = 0..3) covers tile rows 4P .. 4P+3; this lane fills position
    // (lane & 15) of row 4P + (lane >> 4) with source chunk (lane & 15) ^ (row & 15)
    const int drow = 4 * wid + (lane >> 4);                                        // row inside a 16-row group
    const int dvoff = drow * q_pitch + (((lane & 15) ^ (drow & 15)) * 16);           // + u * 16 rows via soffset
    // (a row past L lies past the buffer for a strided projection and inside the next head for the packed copy: zeros
    // or finite values, either way met by lse = +inf)
    const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)q, 0, (Hq - 1) * q_hs + (L - 1) * q_pitch + HD * M::ESIZE, 0x00020000);
    const __amdgpu_buffer_rsrc_t lrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)lse, 0, Hq * L * 4, 0x00020000);
    const bool lse_dma = (nrows % TILE_ROWS == 0);   // uniform: no row of a tile lies past the split
    float lstA = 0.f;
    int nt = 0;
    const int last_row = Hq * L - 1;
    int nrow0 = (g * G) * L + ib;           // row of lse [Hq, L] the cursor's head starts its split at
    int qoff0 = (g * G) * q_hs + ib * q_pitch;   // byte offset of that row in q
    // issues the DMA of the cursor tile into LDS buffer `b`, fetches this thread's lse element, advances the cursor
#define RTK_DMA_ISSUE(b)                                                                                  \
    {                                                                                                     \
        const int qb__ = qoff0 + nt * TILE_ROWS * q_pitch;                                                \
        _Pragma("unroll")                                                                                 \
        for (int u = 0; u < 4; ++u)                                                                       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                     \
                qrsrc, (void __attribute__((address_space(3)))*)(smem + (b) * T::BYTES + (4 * u + wid) * 1024), 16, \
                dvoff, qb__ + 16 * u * q_pitch, 0, 0);                                                    \
        const int r__ = nt * TILE_ROWS + (tid & (TILE_ROWS - 1));                                         \
        lstA = (r__ < nrows) ? lse[min(nrow0 + r__, last_row)] : (FAST ? -INFINITY : INFINITY);           \
        const bool wrap__ = (nt + 1 == tiles_per_head);                                                   \
        nt = wrap__ ? 0 : nt + 1;                                                                         \
        nrow0 += wrap__ ? L : 0;                                                                          \
        qoff0 += wrap__ ? q_hs : 0;                                                                       \
    }
#define RTK_DMA_PIECES(b, rb, U0, U1)                                                                     \
    {                                                                                                     \
        _Pragma("unroll")                                                                                 \
        for (int u = U0; u < U1; ++u)                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                     \
                qrsrc, (void __attribute__((address_space(3)))*)(smem + (b) * T::BYTES + (4 * u + wid) * 1024), 16, \
                dvoff, (rb) + 16 * u * q_pitch, 0, 0);                                                    \
    }
#define RTK_DMA_TAIL(b)                                                                                   \
    {                                                                                                     \
        const int r__ = nt * TILE_ROWS + (tid & (TILE_ROWS - 1));                                         \
        if (lse_dma) { /* whole tiles only: the 64 normalisers go HBM/L2 -> LDS like the tile itself */   \
            if (wid == 0)                                                                                 \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                 \
                    lrsrc, (void __attribute__((address_space(3)))*)(lse_s + (b) * TILE_ROWS), 4, lane * 4, \
                    (nrow0 + nt * TILE_ROWS) * 4, 0, 0);                                                  \
        } else if (wid == 0) lstA = (r__ < nrows) ? lse[min(nrow0 + r__, last_row)] : (FAST ? -INFINITY : INFINITY); \
        const bool wrap__ = (nt + 1 == tiles_per_head);                                                   \
        nt = wrap__ ? 0 : nt + 1;                                                                         \
        nrow0 += wrap__ ? L : 0;                                                                          \
        qoff0 += wrap__ ? q_hs : 0;                                                                       \
    }
#define RTK_DMA_STEP(BUF, ISSUE)                                                                          \
    {                                                                                                     \
        constexpr int buf = BUF;                                                                          \
        const char* cur = smem + buf * T::BYTES;                                                          \
        const float* lcur = lse_s + buf * TILE_ROWS;                                                      \
        const int rb__ = qoff0 + nt * TILE_ROWS * q_pitch;   /* byte offset of the next tile's rows */    \
        if constexpr (ISSUE) RTK_DMA_TAIL(buf ^ 1)                                  \
        _Pragma("unroll")                                                                                 \
        for (int blk = 0; blk < 2; ++blk) {                                                               \
            u32x4 a[M::NREG];                                                                             \
            float ls[16];                                                                                 \
            _Pragma("unroll")                                                                             \
            for (int r = 0; r < M::NREG; ++r) a[r] = *(const u32x4*)(cur + blk * 32 * T::ROWB + frag_off[r]); \
            load_ls(ls, lcur, blk, hf);                                                                   \
            f32x16 acc[NB], lsv;                                                                          \
            _Pragma("unroll")                                                                             \
            for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x16{0};                                          \
            if constexpr (FAST) { /* the accumulator chains start from -lse (what `ls` holds in this mode) */ \
                _Pragma("unroll")                                                                         \
                for (int r = 0; r < 16; ++r) lsv[r] = ls[r];                                              \
            }                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                        \
            _Pragma("unroll")                                                                             \
            for (int r = 0; r < M::NREG; ++r) {                                                           \
                _Pragma("unroll")                                                                         \
                for (int nb = 0; nb < NB; ++nb) {                                                         \
                    if constexpr (FAST) mma16<true>(acc[nb], a[r], kf[nb][r], r == 0 ? lsv : acc[nb]);   \
                    else if constexpr (F16) mma16<true>(acc[nb], a[r], kf[nb][r], acc[nb]);               \
                    else M::mma(acc[nb], a[r], kf[nb][r]);                                                \
                    __builtin_amdgcn_sched_group_barrier(SGB_MFMA, 1, 0);            \
                }                                                                                         \
            }                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                        \
            _Pragma("unroll")                                                                             \
            for (int nb = 0; nb < NB; ++nb) {                                                             \
                if constexpr (FAST) {                                                                     \
                    _Pragma("unroll")                                                                     \
                    for (int r = 0; r < 16; ++r) col[nb] += __builtin_amdgcn_exp2f(acc[nb][r]);           \
                } else colsum_block<DT>(col[nb], acc[nb], ls, c2, sqrt_d);                                \
                asm volatile("" : "+v"(col[nb]) : : "memory");                                            \
                __builtin_amdgcn_sched_barrier(0);                                                        \
                if constexpr (ISSUE) {                                              \
                    if (blk == 0) {                                                                       \
                        if (NB == 1) RTK_DMA_PIECES(buf ^ 1, rb__, 0, 4)                                  \
                        else if (nb == 0) RTK_DMA_PIECES(buf ^ 1, rb__, 0, 2)                             \
                        else if (nb == 1) RTK_DMA_PIECES(buf ^ 1, rb__, 2, 4)                             \
                        __builtin_amdgcn_sched_barrier(0);                                                \
                    }                                                                                     \
                }                                                                                         \
            }                                                                                             \
        }                                                                                                 \
        if constexpr (ISSUE) {                                                                            \
            if (!lse_dma && tid < TILE_ROWS) lse_s[(buf ^ 1) * TILE_ROWS + tid] = lstA;                   \
        }                                                                                                 \
        __syncthreads(); /* drains the DMA (vmcnt(0)) and the LDS reads of this tile */                   \
    }
    RTK_DMA_ISSUE(0)
    if (tid < TILE_ROWS) lse_s[tid] = lstA;
    __syncthreads();
    int it = 0;
    for (; it + 2 < ntiles; it += 2) {
        RTK_DMA_STEP(0, true)
        RTK_DMA_STEP(1, true)
    }
    if (it < ntiles) {
        if (it + 1 < ntiles) RTK_DMA_STEP(0, true)
        else RTK_DMA_STEP(0, false)
        ++it;
    }
    if (it < ntiles) {
        RTK_DMA_STEP(1, false)
        ++it;
    }
#undef RTK_DMA_STEP
#undef RTK_DMA_ISSUE
    int lane_late = lane;
    asm volatile("" : "+v"(lane_late));   // the output address is formed here, not carried (and spilled) through the loop
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float c = col[nb] + __shfl_xor(col[nb], 32, WAVE);
        const int jp = j0 + 32 * nb + (lane_late & 31);
        if (lane_late < 32 && jp < Lk) partial[((size_t)g * RS + rs) * L + (kidx ? kidx[jp] : jp)] = c;
    }
}

// blockIdx.x -> (key tile bx, KV head g, row split rs), blockIdx.y = unit of a batched launch.  A key tile is
// REG_ROWS * NB keys (4 waves x NB x 32).  When the LAST tile holds at most half of that (L = 6272 = 24.5 tiles of 256),
// its workgroups run the one-block body on 32 keys per wave instead of leaving two of four waves without a key: the
// tile costs half the MFMAs (2 % of the launch's arithmetic was spent on keys past L).
template <int NB, bool FAST = false, bool F16 = false>
__global__ __launch_bounds__(SC_BLOCK, (NB == 1 ? 4 : (NB == 2 ? 3 : (NB == 3 ? 2 : 2)))) void score_pass2_dma_kernel(
    const char* __restrict__ q, const char* __restrict__ k, const float* __restrict__ lse, int Hq, int Hkv, int L,
    int rows_per_split, int col_tiles, int RS, int xcd_remap, float* __restrict__ partial, size_t q_unit_bytes,
    size_t k_unit_bytes, size_t lse_unit_floats, size_t part_unit_floats, const int* __restrict__ key_index, QView qv) {
    const int q_hs = qv.row_pitch ? qv.head_stride : L * HD * 2, q_pitch = qv.row_pitch ? qv.row_pitch : HD * 2;
    q = qv.row_pitch ? qv.unit[blockIdx.y] : q + blockIdx.y * q_unit_bytes;
    k += blockIdx.y * k_unit_bytes;
    lse += blockIdx.y * lse_unit_floats;
    partial += blockIdx.y * part_unit_floats;
    // the unit's live keys (key_compact_kernel); a workgroup whose key tile lies past them has nothing to do
    const int* kidx = nullptr;
    int Lk = L;
    if (key_index) {
        const int* ki = key_index + (size_t)blockIdx.y * (L + 1);
        const int n = ki[L];
        if (n >= 0) {
            kidx = ki;
            Lk = n;
        }
    }
    int bx, g, rs;
    {
        int grp;
        if (xcd_remap) {
            const int xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
            grp = xcd + NXCD * (slot / col_tiles);
            bx = slot % col_tiles;
        } else {
            grp = blockIdx.x / col_tiles;
            bx = blockIdx.x % col_tiles;
        }
        g = grp % Hkv;
        rs = grp / Hkv;
    }
    const int j_base = bx * (REG_ROWS * NB);
    if (j_base >= Lk) return;
    if constexpr (NB == 2) {
        if (Lk - j_base <= REG_ROWS) {   // uniform per workgroup
            score_pass2_dma_body<1, FAST, F16>(q, k, lse, Hq, Hkv, L, rows_per_split, RS, partial, j_base, g, rs, kidx, Lk, q_hs,
                                               q_pitch);
            return;
        }
    }
    score_pass2_dma_body<NB, FAST, F16>(q, k, lse, Hq, Hkv, L, rows_per_split, RS, partial, j_base, g, rs, kidx, Lk, q_hs, q_pitch);
}

// ------------------------------------------------------------------------------------------------
// pass 1, LDS-DMA form (bf16): same decomposition as score_pass1_kernel (32 query rows per wave in
// registers, 64-key tiles streamed), with the key tile DMA'd straight into the swizzled LDS image and one
// 32-key block in flight per wave (~100 VGPRs -> 4 waves per SIMD).
// ------------------------------------------------------------------------------------------------
// The work of one workgroup: NB x 32 query rows of head h per wave starting at row i_base + wid * 32 * NB, key split ks.
// MODE is a set of flags: P1_F16 = the operands are fp16 (fast mode, fp16 payloads), P1_SCALED = q~ was pre-scaled by
// log2(e)/sqrt(D) (fast mode: the accumulators are base-2 logits), P1_RAW = plain row sums checked once at the end
// (a fix-up launch with the same flags minus P1_RAW follows).  Statistic: RAW ? (SCALED ? RowStatR : RowStatRX)
//                                                                              : (SCALED ? RowStatF : RowStatB).
constexpr int P1_F16 = 1, P1_SCALED = 2, P1_RAW = 4;
template <int NB, bool LAZY, int MODE = 0>
__device__ __forceinline__ void score_pass1_dma_body(const char* __restrict__ q, const char* __restrict__ k, int Hq, int Hkv,
                                                     int L, int keys_per_split, float* __restrict__ lse_part, int i_base, int h,
                                                     int ks, int neg_out, int q_hs, int q_pitch) {
    constexpr int DT = RTK_BF16;
    using M = MM<DT>;
    using T = Tile<DT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), hf = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int G = Hq / Hkv;
    const int g = h / G;
    const int i0 = i_base + wid * (32 * NB);   // this wave's NB x 32 query rows
    // A wave whose rows all lie past L (a last tile that is between half and three quarters full) keeps its DMA pieces and
    // barriers but skips the MFMAs and the softmax; wave-uniform.  (A last tile that is at most half full runs the
    // one-block body instead, see score_pass1_dma_kernel.)
    const bool live = __builtin_amdgcn_readfirstlane(i0) < L;
    const int jb = ks * keys_per_split, je = min(L, jb + keys_per_split);
    const int nkeys = je - jb;
    const int nfull = nkeys / TILE_ROWS;
    const int ntiles = (nkeys + TILE_ROWS - 1) / TILE_ROWS;
    u32x4 qf[NB][M::NREG];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) load_reg_frag<DT>(q + (size_t)h * q_hs, i0 + 32 * nb, L, lane, qf[nb], q_pitch);
    int frag_off[M::NREG];
    {
        const int row = lane & 31;
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) frag_off[r] = row * T::ROWB + ((M::chunk_of(r, hf) ^ (row & 15)) * 16);
    }
    const float c2 = 1.4426950408889634f / sqrtf((float)HD);
    constexpr bool F16OPS = (MODE & P1_F16) != 0, SCALED = (MODE & P1_SCALED) != 0, RAW = (MODE & P1_RAW) != 0;
    using Stat = std::conditional_t<RAW, std::conditional_t<SCALED, RowStatR, RowStatRX>,
                                    std::conditional_t<SCALED, RowStatF, RowStatB>>;
    Stat rs[NB];
    bool primed[NB];       // RowStatF: has this wave re-based its rows yet?  (wave-uniform)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        rs[nb].init();
        primed[nb] = false;
    }
    const int drow = 4 * wid + (lane >> 4);
    const int dvoff = drow * T::ROWB + (((lane & 15) ^ (drow & 15)) * 16);
    const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(k + (size_t)g * L * HD * M::ESIZE), 0, L * HD * M::ESIZE, 0x00020000);
#define RTK_DMA1_ISSUE(t, b)                                                                              \
    {                                                                                                     \
        _Pragma("unroll")                                                                                 \
        for (int u = 0; u < 4; ++u)                                                                       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                     \
                krsrc, (void __attribute__((address_space(3)))*)(smem + (b) * T::BYTES + (4 * u + wid) * 1024), 16, \
                dvoff, (jb + (t) * TILE_ROWS + 16 * u) * T::ROWB, 0, 0);                                  \
    }
#define RTK_DMA1_PIECES(t, b, U0, U1)                                                                     \
    {                                                                                                     \
        _Pragma("unroll")                                                                                 \
        for (int u = U0; u < U1; ++u)                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                     \
                krsrc, (void __attribute__((address_space(3)))*)(smem + (b) * T::BYTES + (4 * u + wid) * 1024), 16, \
                dvoff, (jb + (t) * TILE_ROWS + 16 * u) * T::ROWB, 0, 0);                                  \
    }
#define RTK_DMA1_STEP(JT, BUF, ISSUE, RAG)                                                                \
    {                                                                                                     \
        constexpr int buf = BUF;                                                                          \
        const char* cur = smem + buf * T::BYTES;                                                          \
        if (!live) { /* this wave's query rows lie past L: it only moves its share of the next tile */    \
            if constexpr (ISSUE) RTK_DMA1_PIECES((JT) + 1, buf ^ 1, 0, 4)           \
        } else {                                                                                          \
        _Pragma("unroll")                                                                                 \
        for (int blk = 0; blk < 2; ++blk) {                                                               \
            u32x4 a[M::NREG];                                                                             \
            _Pragma("unroll")                                                                             \
            for (int r = 0; r < M::NREG; ++r) a[r] = *(const u32x4*)(cur + blk * 32 * T::ROWB + frag_off[r]); \
            f32x16 acc[NB];                                                                               \
            _Pragma("unroll")                                                                             \
            for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x16{0};                                          \
            __builtin_amdgcn_sched_barrier(0);                                        \
            _Pragma("unroll")                                                                             \
            for (int r = 0; r < M::NREG; ++r) {                                                           \
                _Pragma("unroll")                                                                         \
                for (int nb = 0; nb < NB; ++nb) {                                                         \
                    if constexpr (F16OPS) mma16<true>(acc[nb], a[r], qf[nb][r], acc[nb]);                 \
                    else M::mma(acc[nb], a[r], qf[nb][r]);                                                \
                    __builtin_amdgcn_sched_group_barrier(SGB_MFMA, 1, 0);            \
                }                                                                                         \
            }                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                        \
            _Pragma("unroll")                                                                             \
            for (int nb = 0; nb < NB; ++nb) {                                                             \
                if constexpr (RAW && SCALED) {                                                            \
                    rs[nb].template update<RAG>(acc[nb], (JT) * TILE_ROWS + 32 * blk, nkeys, hf);         \
                    asm volatile("" : "+v"(rs[nb].sum) : : "memory");                                     \
                } else if constexpr (RAW) {                                                               \
                    rs[nb].template update<RAG>(acc[nb], (JT) * TILE_ROWS + 32 * blk, nkeys, hf, c2);     \
                    asm volatile("" : "+v"(rs[nb].sum) : : "memory");                                     \
                } else if constexpr (SCALED) {                                                            \
                    rs[nb].template update<RAG>(acc[nb], (JT) * TILE_ROWS + 32 * blk, nkeys, hf, primed[nb]); \
                    asm volatile("" : "+v"(rs[nb].sum), "+v"(rs[nb].off) : : "memory");                   \
                } else {                                                                                  \
                if constexpr (LAZY) rs[nb].template update_lazy<RAG>(acc[nb], (JT) * TILE_ROWS + 32 * blk, nkeys, hf, c2); \
                else rs[nb].template update<RAG>(acc[nb], (JT) * TILE_ROWS + 32 * blk, nkeys, hf, c2);    \
                asm volatile("" : "+v"(rs[nb].sum), "+v"(rs[nb].m) : : "memory");                         \
                }                                                                                         \
                __builtin_amdgcn_sched_barrier(0);                                                        \
                if constexpr (ISSUE) {   /* next tile's DMA pieces inside block 0's softmax */ \
                    if (blk == 0) {                                                                       \
                        if (NB == 1) RTK_DMA1_PIECES((JT) + 1, buf ^ 1, 0, 4)                             \
                        else if (nb == 0) RTK_DMA1_PIECES((JT) + 1, buf ^ 1, 0, 2)                        \
                        else if (nb == 1) RTK_DMA1_PIECES((JT) + 1, buf ^ 1, 2, 4)                        \
                        __builtin_amdgcn_sched_barrier(0);                                                \
                    }                                                                                     \
                }                                                                                         \
            }                                                                                             \
        }                                                                                                 \
        }                                                                                                 \
        __syncthreads(); /* drains the DMA (vmcnt(0)) and the LDS reads of this tile */                   \
    }
    RTK_DMA1_ISSUE(0, 0)
    __syncthreads();
    int jt = 0;
    for (; jt + 2 < nfull; jt += 2) {   // both tiles full, and a tile jt + 2 exists
        RTK_DMA1_STEP(jt, 0, true, false)
        RTK_DMA1_STEP(jt + 1, 1, true, false)
    }
    // at most three tiles left (jt even => buffer parity static); only the last one can be ragged
#define RTK_DMA1_TAIL(PAR)                                                     \
    if (jt < ntiles) {                                                         \
        if (jt + 1 < ntiles) {                                                 \
            if (jt < nfull) RTK_DMA1_STEP(jt, PAR, true, false)                \
            else RTK_DMA1_STEP(jt, PAR, true, true)                            \
        } else {                                                               \
            if (jt < nfull) RTK_DMA1_STEP(jt, PAR, false, false)               \
            else RTK_DMA1_STEP(jt, PAR, false, true)                           \
        }                                                                      \
        ++jt;                                                                  \
    }
    RTK_DMA1_TAIL(0)
    RTK_DMA1_TAIL(1)
    RTK_DMA1_TAIL(0)
#undef RTK_DMA1_TAIL
#undef RTK_DMA1_STEP
#undef RTK_DMA1_PIECES
#undef RTK_DMA1_ISSUE
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float out;
        if constexpr (RAW || SCALED) out = rs[nb].finish();
        else out = rs[nb].finish(c2);
        const int i = i0 + 32 * nb + (lane & 31);
        if (hf == 0 && i < L) lse_part[((size_t)ks * Hq + h) * L + i] = neg_out ? -out : out;
    }
}

// blockIdx.x -> (row tile bx, head h, key split ks), blockIdx.y = (layer, chunk) unit of a batched launch: same shapes,
// operands one unit stride apart.  Like pass 2, a last row tile that is at most half full (L = 6272 = 24.5 tiles) runs the
// one-block body on 32 rows per wave.
template <int NB, bool LAZY, int MODE = 0>
__global__ __launch_bounds__(SC_BLOCK, (NB == 1 ? 4 : 3)) void score_pass1_dma_kernel(
    const char* __restrict__ q, const char* __restrict__ k, int Hq, int Hkv, int L, int keys_per_split, int row_tiles,
    int xcd_remap, float* __restrict__ lse_part, size_t q_unit_bytes, size_t k_unit_bytes, size_t lse_unit_floats,
    int neg_out, QView qv) {
    const int q_hs = qv.row_pitch ? qv.head_stride : L * HD * 2, q_pitch = qv.row_pitch ? qv.row_pitch : HD * 2;
    q = qv.row_pitch ? qv.unit[blockIdx.y] : q + blockIdx.y * q_unit_bytes;
    k += blockIdx.y * k_unit_bytes;
    lse_part += blockIdx.y * lse_unit_floats;
    const int G = Hq / Hkv;
    int bx, h, ks;
    {
        const int per_group = row_tiles * G;
        int grp, w;
        if (xcd_remap) {
            const int xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
            grp = xcd + NXCD * (slot / per_group);
            w = slot % per_group;
        } else {
            grp = blockIdx.x / per_group;
            w = blockIdx.x % per_group;
        }
        ks = grp / Hkv;
        h = (grp % Hkv) * G + w / row_tiles;
        bx = w % row_tiles;
    }
    const int i_base = bx * (REG_ROWS * NB);
    if constexpr (NB == 2) {
        if (L - i_base <= REG_ROWS) {   // uniform per workgroup
            score_pass1_dma_body<1, LAZY, MODE>(q, k, Hq, Hkv, L, keys_per_split, lse_part, i_base, h, ks, neg_out, q_hs, q_pitch);
            return;
        }
    }
    score_pass1_dma_body<NB, LAZY, MODE>(q, k, Hq, Hkv, L, keys_per_split, lse_part, i_base, h, ks, neg_out, q_hs, q_pitch);
}

// RTK_BF16_FAST fix-up launch: the row tiles whose plain sums left fp32's range (published as NaN by RowStatR) are
// recomputed with the offset-carrying form.  Normally there is nothing to fix, so the launch must cost next to nothing: a
// workgroup looks at FIX_TILES consecutive row tiles at once - one load per thread and tile, all in flight together, the
// per-thread NaN bits OR-ed into one LDS word - and runs the robust body only for a tile that holds a NaN (1/32 of the
// main kernel's workgroups instead of a full-size grid whose 39 200 workgroups read 1 KB each and leave: ~60 us).
// Tiles are numbered ((ks * Hq + h) * row_tiles + bx); no XCD-aware decode (nothing streams in the common case).
constexpr int FIX_TILES = 32;   // tiles per workgroup = bits of its NaN mask; 28 units x 700 tiles -> 616 workgroups: one resident round
// NaN scan of FIX_TILES consecutive row tiles (RT rows each) of the row statistics, by the whole workgroup: bit u of the
// result (valid after the caller's barriers, OR-ed into an LDS word) says tile t0 + u holds a NaN.  Tile t =
// kh * row_tiles + bx covers lse_part[kh * L + bx * RT + (0 .. RT)).  16-byte loads, every thread busy, 4 loads per
// thread for 32 tiles of 128 rows (the first form - one scalar load per thread and tile from half the threads - cost
// 12-15 us per launch: tools/debug/fixup_probe.sh).
template <int RT>
__device__ __forceinline__ unsigned scan_nan_tiles(const float* __restrict__ lse_part, int t0, int n_tiles, int row_tiles, int L) {
    constexpr int V4 = RT / 4;                       // 16-byte groups per tile
    constexpr int TPP = SC_BLOCK / V4;               // tiles the workgroup covers per load
    static_assert(SC_BLOCK % V4 == 0 && FIX_TILES % TPP == 0, "scan shape");
    const int tid = (int)threadIdx.x;
    const int ul = tid / V4, r0 = (tid - ul * V4) * 4;
    unsigned mine = 0;
    if ((L & 3) == 0) {
        float4 v[FIX_TILES / TPP];
#pragma unroll
        for (int j = 0; j < FIX_TILES / TPP; ++j) {
            const int t = t0 + ul + j * TPP;
            const int bx = t % row_tiles, kh = t / row_tiles;
            const int i = bx * RT + r0;
            v[j] = (t < n_tiles && i < L) ? *(const float4*)(lse_part + (size_t)kh * L + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < FIX_TILES / TPP; ++j)
            mine |= (v[j].x != v[j].x || v[j].y != v[j].y || v[j].z != v[j].z || v[j].w != v[j].w) ? (1u << (ul + j * TPP)) : 0u;
    } else {
        for (int j = 0; j < FIX_TILES / TPP; ++j) {
            const int t = t0 + ul + j * TPP;
            const int bx = t % row_tiles, kh = t / row_tiles;
            bool nan = false;
            for (int e = 0; e < 4; ++e) {
                const int i = bx * RT + r0 + e;
                const float x = (t < n_tiles && i < L) ? lse_part[(size_t)kh * L + i] : 0.f;
                nan = nan || x != x;
            }
            mine |= nan ? (1u << (ul + j * TPP)) : 0u;
        }
    }
    return mine;
}

template <int NB, int MODE>   // MODE: the robust flags (no P1_RAW) of the launch being repaired
__global__ __launch_bounds__(SC_BLOCK, 2) void score_pass1_fixup_kernel(   // (2: registers, not occupancy - no spills)
    const char* __restrict__ q, const char* __restrict__ k, int Hq, int Hkv, int L, int keys_per_split, int row_tiles,
    int n_tiles, float* __restrict__ lse_part, size_t q_unit_bytes, size_t k_unit_bytes, size_t lse_unit_floats,
    int neg_out, QView qv) {
    if (RTK_FIXUP_PROBE == 1) return;
    const int q_hs = qv.row_pitch ? qv.head_stride : L * HD * 2, q_pitch = qv.row_pitch ? qv.row_pitch : HD * 2;
    q = qv.row_pitch ? qv.unit[blockIdx.y] : q + blockIdx.y * q_unit_bytes;
    k += blockIdx.y * k_unit_bytes;
    lse_part += blockIdx.y * lse_unit_floats;
    const int t0 = blockIdx.x * FIX_TILES;
    __shared__ unsigned nan_tiles;       // bit u: tile t0 + u holds a NaN
    if (threadIdx.x == 0) nan_tiles = 0;
    const unsigned mine = scan_nan_tiles<REG_ROWS * NB>(lse_part, t0, n_tiles, row_tiles, L);
    __syncthreads();
    if (mine) atomicOr(&nan_tiles, mine);
    __syncthreads();
    unsigned todo = RTK_FIXUP_PROBE >= 2 ? 0u : nan_tiles;           // uniform: the whole workgroup takes the same path
    while (todo) {
        const int u = __builtin_ctz(todo);
        todo &= todo - 1;
        const int t = t0 + u;
        const int bx = t % row_tiles, kh = t / row_tiles;
        const int h = kh % Hq, ks = kh / Hq;
        const int i_base = bx * (REG_ROWS * NB);
        if (NB == 2 && L - i_base <= REG_ROWS)
            score_pass1_dma_body<1, true, MODE>(q, k, Hq, Hkv, L, keys_per_split, lse_part, i_base, h, ks, neg_out, q_hs, q_pitch);
        else
            score_pass1_dma_body<NB, true, MODE>(q, k, Hq, Hkv, L, keys_per_split, lse_part, i_base, h, ks, neg_out, q_hs, q_pitch);
        __syncthreads();   // the next tile's prologue writes the LDS buffers this one was still reading
    }
}

}  // namespace rtk
#include "score_refround.cuh"
namespace rtk {

// ------------------------------------------------------------------------------------------------
// generic fallback (any head_dim; small problems): plain fp32 VALU, same two passes.
// ------------------------------------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ float ldx(const void* p, size_t i) {
    if constexpr (DT != RTK_F32) return H16<DT>::ld(p, i);
    else return ((const float*)p)[i];
}

template <int DT>
__global__ __launch_bounds__(256) void score_pass1_generic(const void* __restrict__ q, const void* __restrict__ k,
                                                           int Hq, int Hkv, int L, int D, float* __restrict__ lse) {
    extern __shared__ float qs[];  // [D]
    __shared__ float red_m[4], red_s[4];
    const int i = blockIdx.x, h = blockIdx.y, g = h / (Hq / Hkv), tid = threadIdx.x;
    for (int d = tid; d < D; d += blockDim.x) qs[d] = ldx<DT>(q, ((size_t)h * L + i) * D + d);
    __syncthreads();
    const float sqrt_d = sqrtf((float)D);
    float m = -INFINITY, sum = 0.f;
    for (int j = tid; j < L; j += blockDim.x) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s = fmaf(qs[d], ldx<DT>(k, ((size_t)g * L + j) * D + d), s);
        s = __fdiv_rn(s, sqrt_d);
        const float mn = fmaxf(m, s);
        sum = sum * expf(m - mn) + expf(s - mn);
        m = mn;
    }
    // wave then block combine of (m, sum)
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, WAVE), s2 = __shfl_xor(sum, o, WAVE);
        const float mm = fmaxf(m, m2);
        if (mm > -INFINITY) sum = sum * expf(m - mm) + s2 * expf(m2 - mm);
        m = mm;
    }
    if ((tid & 63) == 0) { red_m[tid / 64] = m; red_s[tid / 64] = sum; }
    __syncthreads();
    if (tid == 0) {
        float mm = red_m[0], ss = red_s[0];
        for (int w = 1; w < 4; ++w) {
            const float m2 = red_m[w], s2 = red_s[w];
            const float mx = fmaxf(mm, m2);
            if (mx > -INFINITY) ss = ss * expf(mm - mx) + s2 * expf(m2 - mx);
            mm = mx;
        }
        lse[(size_t)h * L + i] = mm + logf(ss);
    }
}

template <int DT>
__global__ __launch_bounds__(256) void score_pass2_generic(const void* __restrict__ q, const void* __restrict__ k,
                                                           const float* __restrict__ lse, int Hq, int Hkv, int L, int D,
                                                           float* __restrict__ partial) {
    extern __shared__ float qs[];  // [D]
    const int g = blockIdx.y, G = Hq / Hkv, tid = threadIdx.x;
    const int j = blockIdx.x * blockDim.x + tid;
    const float sqrt_d = sqrtf((float)D);
    float col = 0.f;
    for (int hh = 0; hh < G; ++hh) {
        const int h = g * G + hh;
        float blk = 0.f;   // two-level sum: 64-row blocks, then blocks (a plain running sum over G * L terms drifts by ~1e-5)
        for (int i = 0; i < L; ++i) {
            __syncthreads();
            for (int d = tid; d < D; d += blockDim.x) qs[d] = ldx<DT>(q, ((size_t)h * L + i) * D + d);
            __syncthreads();
            if (j < L) {
                float s = 0.f;
                for (int d = 0; d < D; ++d) s = fmaf(qs[d], ldx<DT>(k, ((size_t)g * L + j) * D + d), s);
                blk += expf(__fdiv_rn(s, sqrt_d) - lse[(size_t)h * L + i]);
            }
            if ((i & 63) == 63 || i + 1 == L) {
                col += blk;
                blk = 0.f;
            }
        }
    }
    if (j < L) partial[(size_t)g * L + j] = col;
}

// finalize: score[j] = mean_g( (sum_split partial[g,split,j]) / G )      (longvideo_cache.py:269-270)
// Fixed summation order (bit-reproducible).  A workgroup owns 64 keys; its four waves take the KV groups
// g = wave, wave + 4, ... so that up to 4 x 8 loads per key are in flight, then wave 0 adds the per-group means
// in group order.
__global__ __launch_bounds__(256) void score_finalize_kernel(const float* __restrict__ partial, int Hkv, int RS, int G,
                                                             int L, float* __restrict__ score) {
    extern __shared__ float fin_gs[];  // [Hkv][64]
    const int jl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jl;
    const int jc = min(j, L - 1);
    for (int g = part; g < Hkv; g += 4) {
        const float* p = partial + (size_t)g * RS * L + jc;
        float gs = 0.f;
        int r = 0;
        for (; r + 8 <= RS; r += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(r + u) * L];
#pragma unroll
            for (int u = 0; u < 8; ++u) gs += v[u];
        }
        for (; r < RS; ++r) gs += p[(size_t)r * L];
        fin_gs[g * 64 + jl] = gs / (float)G;
    }
    __syncthreads();
    if (part == 0 && j < L) {
        float tot = 0.f;
        for (int g = 0; g < Hkv; ++g) tot += fin_gs[g * 64 + jl];
        score[j] = tot / (float)Hkv;
    }
}

// Work decomposition.  Both passes are cut into >= ~3000 workgroups (about 4 rounds over 256 CUs x 3
// resident workgroups) so the last round's tail stays small; the splits depend on the shape only,
// so results are deterministic.
constexpr int TARGET_WGS = 3072;
static int pick_splits(int tiles_fixed, int heads, int stream_tiles, int cap, int Hkv) {
    int s = (TARGET_WGS + tiles_fixed * heads - 1) / (tiles_fixed * heads);
    s = std::max(1, std::min(std::min(s, cap), stream_tiles));
    // prefer a split count whose NON-EMPTY splits make Hkv*splits a multiple of the XCD count (balanced
    // XCD-aware mapping); splits are whole 64-row tiles, so check the effective count
    for (int t = s; t <= std::min(cap, stream_tiles) && t <= s + 8; ++t) {
        const int per = (((stream_tiles + t - 1) / t));          // tiles per split
        const int eff = (stream_tiles + per - 1) / per;
        if ((Hkv * eff) % NXCD == 0) return t;
    }
    return s;
}

// bf16(x * (1/a2)) == bf16(x / a2) for EVERY finite bf16 x?  (x is bf16-valued in the un-rotate chain, so the
// check is exhaustive: 65536 cases, cached per a2.)  True for the YaRN factor-4 scaling 1.1386^2.
bool bf16_rcp_is_exact(float a2) {
    static std::mutex mu;
    static std::map<uint32_t, bool> cache;
    uint32_t key;
    memcpy(&key, &a2, 4);
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    auto to_bf = [](float f) -> uint16_t {
        uint32_t u;
        memcpy(&u, &f, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    };
    const volatile float rcp = 1.0f / a2;
    bool ok = std::isfinite(rcp) && a2 != 0.0f;
    for (uint32_t b = 0; ok && b < 65536; ++b) {
        const uint32_t u = b << 16;
        float x;
        memcpy(&x, &u, 4);
        if (!std::isfinite(x)) continue;
        const volatile float qd = x / a2, qm = x * rcp;   // volatile: no fused / extended-precision evaluation
        // flush-to-zero differences between host and device do not matter: both sides would round tiny values the same
        if (to_bf(qd) != to_bf(qm)) ok = false;
    }
    cache[key] = ok;
    return ok;
}

struct ScoreWs {
    size_t q_off, k_off, lse_off, part_off, total;
    int RS, KS;
    bool ref;   // RTK_BF16_REFROUND / RTK_F16_REFROUND: the reference's rounding chain; column partials are per head
    bool fast;  // RTK_BF16_FAST: q~ (pre-scaled) and a second copy of k~ (at k_off) are fp16
    bool h16;   // RTK_F16 / RTK_F16_REFROUND: fp16 payloads (un-rotation rounds to fp16, the passes use the fp16 matrix instruction)
};
static ScoreWs score_ws(int Hq, int Hkv, int L, int D, int dtype) {
    const bool many = !RTK_IGNORE_MANY_UNITS && (dtype & RTK_SCORE_MANY_UNITS) != 0;   // the caller batches many units per launch
    dtype &= ~RTK_SCORE_MANY_UNITS;
    const size_t es = dtype == RTK_F32 ? 4 : 2;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    ScoreWs w;
    w.ref = dtype == RTK_BF16_REFROUND || dtype == RTK_F16_REFROUND;
    w.fast = dtype == RTK_BF16_FAST;
    w.h16 = dtype == RTK_F16 || dtype == RTK_F16_REFROUND;
    const int nbr = REG_ROWS;
    const int reg_tiles = (L + nbr - 1) / nbr, stream_tiles = (L + TILE_ROWS - 1) / TILE_ROWS;
    w.RS = (D == HD) ? pick_splits(reg_tiles, Hkv, stream_tiles, 32, Hkv) : 1;
    w.KS = (D == HD) ? pick_splits(reg_tiles, Hq, stream_tiles, 8, Hkv) : 1;
    // bf16 production path: the chunk-batched launches bring their own parallelism (28 layers), so pass 1 prefers
    // longer key streams per workgroup (measured: 2 splits -1.7 % over 4) and half the lse partials
    if (D == HD && dtype != RTK_F32) w.KS = std::min(w.KS, 2);
    if (many && D == HD && dtype != RTK_F32) {
        // Launches of many units bring their own parallelism, so the splits are chosen for the length of a workgroup's
        // stream instead of for the workgroup count of ONE unit (same-box A/Bs at L = 2304 and 6272,
        // profiles/r06_ab_splits.txt): one key split (pass 1 -1.6 % / -0.4 %, no lse_combine launch), about eight row
        // tiles per row split (pass 2 -2 % at L = 2304; 14 splits of 7 tiles at L = 6272, what pick_splits gave already).
        w.KS = 1;
        const int s0 = std::max(1, (stream_tiles + 7) / 8);
        w.RS = s0;
        for (int t = s0; t <= std::min(stream_tiles, s0 + 8); ++t) {
            const int per = (stream_tiles + t - 1) / t, eff = (stream_tiles + per - 1) / per;
            if ((Hkv * eff) % NXCD == 0) { w.RS = t; break; }
        }
    }
    if (RTK_FORCE_KS > 0 && D == HD && dtype != RTK_F32) w.KS = RTK_FORCE_KS;   // A/B builds only (variants.h)
    if (RTK_FORCE_RS > 0 && D == HD && dtype != RTK_F32) w.RS = RTK_FORCE_RS;
    w.q_off = 0;
    w.k_off = al((size_t)Hq * L * D * es);
    w.lse_off = w.k_off + al((size_t)Hkv * L * D * es);
    w.part_off = w.lse_off + al((size_t)w.KS * Hq * L * 4);
    w.total = w.part_off + al((size_t)(w.ref ? Hq : Hkv) * w.RS * L * 4);
    return w;
}

}  // namespace rtk

using namespace rtk;

extern "C" size_t rtk_pivotkv_score_workspace_bytes(int Hq, int Hkv, int L, int D, int dtype) {
    if (Hq < 1 || Hkv < 1 || L < 1 || D < 1) return 0;
    return score_ws(Hq, Hkv, L, D, dtype).total;
}

template <int DT>
static int score_impl(const void* q, int64_t qsh, int64_t qsl, const void* k, int64_t ksh, int64_t ksl, int Hq,
                      int Hkv, int L, int D, const float* cosv, const float* sinv, float a, float* score,
                      void* k_unrot, char* ws, const ScoreWs& w, int stages, float* partial_out, hipStream_t st,
                      int n_units = 1, size_t ws_stride = 0, size_t k_stride = 0, size_t part_stride = 0,
                      const int* key_index = nullptr, const QView* q_view = nullptr) {
    // q_view (bf16 LDS-DMA passes only, n_units <= MAX_Q_UNITS): the units' queries are read where the caller keeps
    // them (per-unit pointers, head stride, row pitch) instead of from the packed copies inside the workspaces
    QView qv;
    memset(&qv, 0, sizeof(qv));
    if (q_view) qv = *q_view;
    // n_units > 1 (RTK_SCORE_PASSES only): the same passes for n_units units whose workspaces / k~ / partials lie
    // ws_stride / k_stride bytes and part_stride floats apart — one launch per kernel, blockIdx.y = unit
    char* qt = ws + w.q_off;
    char* kt = (k_unrot && !w.fast) ? (char*)k_unrot : ws + w.k_off;   // FAST scores on the fp16 copy inside the workspace
    if (w.fast) k_stride = ws_stride;
    float* lse = (float*)(ws + w.lse_off);
    float* part = partial_out ? partial_out : (float*)(ws + w.part_off);
    const float a2 = (float)((double)a * (double)a);  // python float ** 2, then an fp32 tensor / scalar
    if (stages & RTK_SCORE_PREPARE) {
        constexpr int VE = Vec16<DT>::VE;
        const int es = 16 / VE;
        const bool vec_ok = (D % (2 * VE) == 0) && ((qsh * es) % 16 == 0) && ((qsl * es) % 16 == 0) &&
                            ((ksh * es) % 16 == 0) && ((ksl * es) % 16 == 0) &&
                            ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)qt | (uintptr_t)kt) & 15) == 0) &&
                            (!cosv || (((uintptr_t)cosv | (uintptr_t)sinv) & 15) == 0);
        if (w.fast && !(vec_ok && D == HD)) {
            set_error("rtk_pivotkv_score: RTK_BF16_FAST needs head_dim %d and 16-byte aligned rows", HD);
            return RTK_EUNSUPPORTED;
        }
        if (vec_ok) {
            const int threads = L * (D / 2 / VE);
            const int groups = (Hq + UNROT_HEADS - 1) / UNROT_HEADS + (Hkv + UNROT_HEADS - 1) / UNROT_HEADS;
            const dim3 grid((threads + 255) / 256, groups);
            const int div = (!cosv || a2 == 1.0f) ? 0 : ((DT == RTK_BF16 && bf16_rcp_is_exact(a2)) ? 1 : 2);
            const float rcp = 1.0f / a2;
            bool done = false;
            if constexpr (DT == RTK_BF16) {
                if (w.h16) {   // fp16 payloads: the same chain rounded to fp16, IEEE division
#define RTK_UNROT_H(DIV)                                                                                           \
    RTK_LAUNCH(KID_UNROT, (unrotate_pack_vec_kernel<RTK_F16, DIV>), grid, dim3(256), 0, st, (const char*)q, qsh, qsl, \
               (const char*)k, ksh, ksl, Hq, Hkv, L, D, cosv, sinv, a2, rcp, qt, kt)
                    if (!cosv || a2 == 1.0f) RTK_UNROT_H(0);
                    else RTK_UNROT_H(2);
#undef RTK_UNROT_H
                    done = true;
                }
                if (w.fast) {   // q~ -> fp16(q~ * log2(e)/sqrt(D)); k~ -> bf16 (only if the caller wants it) + fp16 at k_off
                    const float qscale = 1.4426950408889634f / sqrtf((float)HD);
#define RTK_UNROT_F(DIV)                                                                                           \
    RTK_LAUNCH(KID_UNROT, (unrotate_pack_vec_kernel<DT, DIV, true>), grid, dim3(256), 0, st, (const char*)q, qsh, qsl, \
               (const char*)k, ksh, ksl, Hq, Hkv, L, D, cosv, sinv, a2, rcp, qt, (char*)k_unrot, ws + w.k_off, qscale)
                    if (div == 0) RTK_UNROT_F(0);
                    else if (div == 1) RTK_UNROT_F(1);
                    else RTK_UNROT_F(2);
#undef RTK_UNROT_F
                    done = true;
                }
            }
#define RTK_UNROT(DIV)                                                                                             \
    RTK_LAUNCH(KID_UNROT, (unrotate_pack_vec_kernel<DT, DIV>), grid, dim3(256), 0, st, (const char*)q, qsh, qsl,     \
               (const char*)k, ksh, ksl, Hq, Hkv, L, D, cosv, sinv, a2, rcp, qt, kt)
            if (done) {}
            else if (div == 0) RTK_UNROT(0);
            else if (div == 1) RTK_UNROT(1);
            else RTK_UNROT(2);
#undef RTK_UNROT
        } else {
            const size_t nq = (size_t)Hq * L * (D / 2), nk = (size_t)Hkv * L * (D / 2);
            const dim3 gq((unsigned)std::min<size_t>((nq + 255) / 256, 8192)), gk((unsigned)std::min<size_t>((nk + 255) / 256, 8192));
            if (w.h16) {
                RTK_LAUNCH(KID_UNROT, unrotate_pack_kernel<RTK_F16>, gq, dim3(256), 0, st, q, qsh, qsl, Hq, L, D, cosv, sinv, a2, (void*)qt);
                RTK_LAUNCH(KID_UNROT, unrotate_pack_kernel<RTK_F16>, gk, dim3(256), 0, st, k, ksh, ksl, Hkv, L, D, cosv, sinv, a2, (void*)kt);
            } else {
                RTK_LAUNCH(KID_UNROT, unrotate_pack_kernel<DT>, gq, dim3(256), 0, st, q, qsh, qsl, Hq, L, D, cosv, sinv, a2, (void*)qt);
                RTK_LAUNCH(KID_UNROT, unrotate_pack_kernel<DT>, gk, dim3(256), 0, st, k, ksh, ksl, Hkv, L, D, cosv, sinv, a2, (void*)kt);
            }
        }
        RTK_LAUNCH_CHECK("unrotate_pack_kernel");
    }
    const int G = Hq / Hkv;
    int rs_n = 1;
    if (w.ref) {
        if constexpr (DT == RTK_BF16) {
            if (D != HD) {
                set_error("rtk_pivotkv_score: the reference-rounding modes need head_dim %d", HD);
                return RTK_EUNSUPPORTED;
            }
            constexpr int TILE_BYTES = Tile<DT>::BYTES;
            constexpr int LDS1 = 2 * TILE_BYTES, LDS2 = 2 * TILE_BYTES + 2 * TILE_ROWS * (int)sizeof(float);
            const int jt = (L + REG_ROWS * REF_NB - 1) / (REG_ROWS * REF_NB), jt2 = (L + REG_ROWS * REF_NB2 - 1) / (REG_ROWS * REF_NB2);
            auto per_split = [](int n, int parts) { return (((n + parts - 1) / parts + TILE_ROWS - 1) / TILE_ROWS) * TILE_ROWS; };
            const int kps = per_split(L, w.KS), rps = per_split(L, w.RS);
            const int ks_n = (L + kps - 1) / kps;
            rs_n = (L + rps - 1) / rps;
            const float sqrt_d = (float)sqrt((double)HD);   // python: math.sqrt(self.head_dim), then an fp32 opmath scalar
            const bool rcp_ok = !w.h16 && bf16_rcp_is_exact(sqrt_d);   // (fp16 payloads: IEEE division)
            const float rcp_sd = 1.0f / sqrt_d;
            if (stages & RTK_SCORE_PASSES) {
                const int n_tiles = Hkv * ks_n * jt * G;
                const dim3 g1(n_tiles, n_units), gf((n_tiles + FIX_TILES - 1) / FIX_TILES, n_units), g2(Hkv * rs_n * jt2, n_units);
                const int x1 = (int)((Hkv * ks_n) % NXCD == 0), x2 = (int)((Hkv * rs_n) % NXCD == 0);
                const size_t su = ws_stride / sizeof(float);
                // raw row sums + the fix-up launch for rows whose sum left fp32's range, lse combine over key splits,
                // then the column sums of the bf16 probabilities per head
#define RTK_REF_PASSES(DIV, F16)                                                                                             \
    RTK_LAUNCH(KID_PASS1, (score_pass1_ref_kernel<DIV, F16>), g1, dim3(SC_BLOCK), LDS1, st, (const char*)qt, (const char*)kt,   \
               Hq, Hkv, L, kps, jt, x1, lse, ws_stride, k_stride, su, sqrt_d, rcp_sd);                                     \
    RTK_LAUNCH(KID_FINALIZE, (score_pass1_ref_fixup_kernel<DIV, F16>), gf, dim3(SC_BLOCK), LDS1, st, (const char*)qt,           \
               (const char*)kt, Hq, Hkv, L, kps, jt, n_tiles, lse, ws_stride, k_stride, su, sqrt_d, rcp_sd);               \
    if (ks_n > 1)                                                                                                        \
        RTK_LAUNCH(KID_FINALIZE, lse_combine_kernel<RTK_BF16>, dim3((unsigned)(((size_t)Hq * L + 255) / 256), n_units),    \
                   dim3(256), 0, st, lse, (size_t)Hq * L, ks_n, su, 0);                                                    \
    RTK_LAUNCH(KID_PASS2, (score_pass2_ref_kernel<DIV, F16>), g2, dim3(SC_BLOCK), LDS2, st, (const char*)qt, (const char*)kt,   \
               (const float*)lse, Hq, Hkv, L, rps, jt2, rs_n, x2, part, ws_stride, k_stride, su, part_stride, sqrt_d,      \
               rcp_sd, key_index)
                if (w.h16) { RTK_REF_PASSES(2, true); } else if (rcp_ok) { RTK_REF_PASSES(1, false); } else { RTK_REF_PASSES(2, false); }
#undef RTK_REF_PASSES
                RTK_LAUNCH_CHECK("score_ref_passes");
            }
            if (stages & RTK_SCORE_FINALIZE) {
                if (w.h16)
                    RTK_LAUNCH(KID_FINALIZE, score_finalize_ref_kernel<true>, dim3((L + 255) / 256), dim3(256), 0, st, part, Hkv,
                               rs_n, G, L, score);
                else
                    RTK_LAUNCH(KID_FINALIZE, score_finalize_ref_kernel<false>, dim3((L + 255) / 256), dim3(256), 0, st, part, Hkv,
                               rs_n, G, L, score);
                RTK_LAUNCH_CHECK("score_finalize_ref_kernel");
            }
        }
        return RTK_OK;
    }
    if (w.fast && D != HD) {
        set_error("rtk_pivotkv_score: RTK_BF16_FAST needs head_dim %d", HD);
        return RTK_EUNSUPPORTED;
    }
    if (D == HD) {
        constexpr int TILE_BYTES = Tile<DT>::BYTES;
        constexpr int NBR = RegBlocks<DT>::NB;
        constexpr int LDS1 = 2 * TILE_BYTES, LDS2 = 2 * TILE_BYTES + 2 * TILE_ROWS * (int)sizeof(float);
        // > 64 KiB of dynamic LDS (fp32 tiles) needs an opt-in per kernel AND per device: remembered in one atomic
        // bit per (dtype, device) - racing first calls both opt in, which is harmless
        static std::atomic<uint64_t> opted[2];
        int dev_id = 0;
        (void)hipGetDevice(&dev_id);
        const uint64_t dev_bit = 1ull << (dev_id & 63);
        if (!(opted[DT == RTK_BF16].load(std::memory_order_relaxed) & dev_bit)) {
            if constexpr (DT == RTK_BF16) {
                (void)hipFuncSetAttribute((const void*)score_pass1_dma_kernel<RTK_P1_NB, RTK_P1_LAZY>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS1);
                (void)hipFuncSetAttribute((const void*)score_pass2_dma_kernel<RTK_P2_NB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
            } else {
                (void)hipFuncSetAttribute((const void*)score_pass1_kernel<DT, NBR>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS1);
                (void)hipFuncSetAttribute((const void*)score_pass2_kernel<DT, NBR>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
            }
            opted[DT == RTK_BF16].fetch_or(dev_bit, std::memory_order_relaxed);
        }
        const int jt = (L + REG_ROWS * NBR - 1) / (REG_ROWS * NBR);
        auto per_split = [](int n, int parts) { return (((n + parts - 1) / parts + TILE_ROWS - 1) / TILE_ROWS) * TILE_ROWS; };
        const int kps = per_split(L, w.KS), rps = per_split(L, w.RS);
        const int ks_n4 = (L + kps - 1) / kps;  // non-empty splits only
        rs_n = (L + rps - 1) / rps;
        const int ks_n = ks_n4;
        // bf16: the LDS-DMA kernels; fp32 (parity dtype): the register-staged kernels
        constexpr bool dma = (DT == RTK_BF16);
        if (stages & RTK_SCORE_PASSES) {
            if constexpr (dma) {
                const int jt1 = (L + REG_ROWS * RTK_P1_NB - 1) / (REG_ROWS * RTK_P1_NB);
                // raw row sums + a fix-up launch (1/16 of the grid) for the rows whose sum left fp32's range; the
                // fast mode's pass 2 starts its accumulators from -lse, so whoever writes the final lse negates it
                const int n_tiles = Hkv * ks_n * jt1 * G;
                const dim3 g1(n_tiles, n_units), gf((n_tiles + FIX_TILES - 1) / FIX_TILES, n_units);
                const int x1 = (int)((Hkv * ks_n) % NXCD == 0), neg = (int)(w.fast && ks_n == 1);
#define RTK_P1(MODEV)                                                                                                  \
    RTK_LAUNCH(KID_PASS1, (score_pass1_dma_kernel<RTK_P1_NB, RTK_P1_LAZY, MODEV>), g1, dim3(SC_BLOCK), LDS1, st,           \
               (const char*)qt, (const char*)kt, Hq, Hkv, L, kps, jt1, x1, lse, ws_stride, k_stride,                     \
               ws_stride / sizeof(float), neg, qv);                                                                     \
    if ((MODEV) & P1_RAW)                                                                                               \
        RTK_LAUNCH(KID_FINALIZE, (score_pass1_fixup_kernel<RTK_P1_NB, (MODEV) & ~P1_RAW>), gf, dim3(SC_BLOCK), (RTK_FIXUP_PROBE == 3 ? 0 : LDS1), st,   \
                   (const char*)qt, (const char*)kt, Hq, Hkv, L, kps, jt1, n_tiles, lse, ws_stride, k_stride,            \
                   ws_stride / sizeof(float), neg, qv)
                constexpr int RAWF = RTK_P1_RAW ? P1_RAW : 0;
                if (w.fast) { RTK_P1(P1_F16 | P1_SCALED | P1_RAW); }
                else if (w.h16) { RTK_P1(P1_F16 | RAWF); }
                else { RTK_P1(RAWF); }
#undef RTK_P1
            }
            else
                RTK_LAUNCH(KID_PASS1, (score_pass1_kernel<DT, NBR>), dim3(Hkv * ks_n * jt * G), dim3(SC_BLOCK), LDS1, st,
                           (const char*)qt, (const char*)kt, Hq, Hkv, L, kps, jt, (int)((Hkv * ks_n) % NXCD == 0), lse);
            RTK_LAUNCH_CHECK("score_pass1_kernel");
            if (ks_n > 1) {
                const size_t n = (size_t)Hq * L;
                RTK_LAUNCH(KID_FINALIZE, lse_combine_kernel<DT>, dim3((unsigned)((n + 255) / 256), n_units), dim3(256), 0, st, lse, n,
                           ks_n, ws_stride / sizeof(float), (int)w.fast);
            }
            if constexpr (dma) {
                const int jt2 = (L + REG_ROWS * RTK_P2_NB - 1) / (REG_ROWS * RTK_P2_NB);
                if (w.fast)
                    RTK_LAUNCH(KID_PASS2, (score_pass2_dma_kernel<RTK_P2_NB, true>), dim3(Hkv * rs_n * jt2, n_units), dim3(SC_BLOCK), LDS2, st,
                               (const char*)qt, (const char*)kt, (const float*)lse, Hq, Hkv, L, rps, jt2, rs_n,
                               (int)((Hkv * rs_n) % NXCD == 0), part, ws_stride, k_stride, ws_stride / sizeof(float),
                               part_stride, key_index, qv);
                else if (w.h16)
                    RTK_LAUNCH(KID_PASS2, (score_pass2_dma_kernel<RTK_P2_NB, false, true>), dim3(Hkv * rs_n * jt2, n_units), dim3(SC_BLOCK), LDS2, st,
                               (const char*)qt, (const char*)kt, (const float*)lse, Hq, Hkv, L, rps, jt2, rs_n,
                               (int)((Hkv * rs_n) % NXCD == 0), part, ws_stride, k_stride, ws_stride / sizeof(float),
                               part_stride, key_index, qv);
                else
                RTK_LAUNCH(KID_PASS2, (score_pass2_dma_kernel<RTK_P2_NB>), dim3(Hkv * rs_n * jt2, n_units), dim3(SC_BLOCK), LDS2, st,
                           (const char*)qt, (const char*)kt, (const float*)lse, Hq, Hkv, L, rps, jt2, rs_n,
                           (int)((Hkv * rs_n) % NXCD == 0), part, ws_stride, k_stride, ws_stride / sizeof(float),
                           part_stride, key_index, qv);
            }
            else
                RTK_LAUNCH(KID_PASS2, (score_pass2_kernel<DT, NBR>), dim3(Hkv * rs_n * jt), dim3(SC_BLOCK), LDS2, st,
                           (const char*)qt, (const char*)kt, (const float*)lse, Hq, Hkv, L, rps, jt, rs_n,
                           (int)((Hkv * rs_n) % NXCD == 0), part, key_index);
            RTK_LAUNCH_CHECK("score_pass2_kernel");
        }
    } else if (stages & RTK_SCORE_PASSES) {
        if (w.h16) {
            RTK_LAUNCH(KID_PASS1, score_pass1_generic<RTK_F16>, dim3(L, Hq), dim3(256), D * sizeof(float), st, (const void*)qt,
                               (const void*)kt, Hq, Hkv, L, D, lse);
            RTK_LAUNCH(KID_PASS2, score_pass2_generic<RTK_F16>, dim3((L + 255) / 256, Hkv), dim3(256), D * sizeof(float), st,
                               (const void*)qt, (const void*)kt, lse, Hq, Hkv, L, D, part);
        } else {
        RTK_LAUNCH(KID_PASS1, score_pass1_generic<DT>, dim3(L, Hq), dim3(256), D * sizeof(float), st, (const void*)qt,
                           (const void*)kt, Hq, Hkv, L, D, lse);
        RTK_LAUNCH(KID_PASS2, score_pass2_generic<DT>, dim3((L + 255) / 256, Hkv), dim3(256), D * sizeof(float), st,
                           (const void*)qt, (const void*)kt, lse, Hq, Hkv, L, D, part);
        }
        RTK_LAUNCH_CHECK("score_generic");
    }
    if (stages & RTK_SCORE_FINALIZE) {
        RTK_LAUNCH(KID_FINALIZE, score_finalize_kernel, dim3((L + 63) / 64), dim3(256), (size_t)Hkv * 64 * sizeof(float), st,
                   part, Hkv, rs_n, G, L, score);
        RTK_LAUNCH_CHECK("score_finalize_kernel");
    }
    return RTK_OK;
}

extern "C" int rtk_pivotkv_score_passes_batched(void* workspace0, size_t workspace_stride, void* k_unrot0,
                                                size_t k_unrot_stride, float* partial0, size_t partial_stride_floats,
                                                int n_units, int Hq, int Hkv, int L, int D, int dtype,
                                                const void* const* key_masks_host, int32_t* key_index_ws,
                                                rtk_stream_t stream) {
    const int dtype_full = dtype;              // may carry RTK_SCORE_MANY_UNITS (split policy, see score_ws)
    dtype &= ~RTK_SCORE_MANY_UNITS;
    RTK_CHECK_ARG(workspace0 && partial0 && n_units >= 1, "rtk_pivotkv_score_passes_batched: NULL pointer or no units");
    RTK_CHECK_ARG(Hq >= 1 && Hkv >= 1 && Hq % Hkv == 0 && L >= 1, "rtk_pivotkv_score_passes_batched: bad shape");
    RTK_CHECK_ARG(((uintptr_t)workspace0 & 255) == 0 && workspace_stride % 256 == 0,
                  "rtk_pivotkv_score_passes_batched: workspaces must be 256-byte aligned");
    const ScoreWs w = score_ws(Hq, Hkv, L, D, dtype_full);
    RTK_CHECK_ARG(n_units == 1 || workspace_stride >= w.total, "rtk_pivotkv_score_passes_batched: workspace stride too small");
    if ((dtype != RTK_BF16 && dtype != RTK_BF16_REFROUND && dtype != RTK_BF16_FAST && dtype != RTK_F16 && dtype != RTK_F16_REFROUND) || D != HD) {
        set_error("rtk_pivotkv_score_passes_batched: bf16 with head_dim %d only (call RTK_SCORE_PASSES per unit)", HD);
        return RTK_EUNSUPPORTED;
    }
    // live keys of pass 2: units whose key-patch mask is known skip the columns the mask override discards anyway
    const int* key_index = nullptr;
    if (key_masks_host && key_index_ws) {
        bool any = false;
        for (int u = 0; u < n_units; ++u) any = any || key_masks_host[u];
        if (any) {
            for (int u0 = 0; u0 < n_units; u0 += MAX_MASK_UNITS) {
                KeyMasks km;
                const int m = std::min(MAX_MASK_UNITS, n_units - u0);
                for (int u = 0; u < MAX_MASK_UNITS; ++u) km.m[u] = u < m ? (const uint8_t*)key_masks_host[u0 + u] : nullptr;
                RTK_LAUNCH(KID_FINALIZE, key_compact_kernel, dim3(m), dim3(1024), 0, (hipStream_t)stream, km, L,
                           key_index_ws + (size_t)u0 * (L + 1));
            }
            RTK_LAUNCH_CHECK("key_compact_kernel");
            key_index = key_index_ws;
        }
    }
    float dummy_score = 0.f;  // not touched by RTK_SCORE_PASSES
    return score_impl<RTK_BF16>(workspace0, 0, 0, workspace0, 0, 0, Hq, Hkv, L, D, nullptr, nullptr, 1.0f, &dummy_score,
                                k_unrot0, (char*)workspace0, w, RTK_SCORE_PASSES, partial0, (hipStream_t)stream, n_units,
                                workspace_stride, k_unrot0 ? k_unrot_stride : workspace_stride, partial_stride_floats,
                                key_index);
}

extern "C" int rtk_pivotkv_score_passes_batched_q(void* workspace0, size_t workspace_stride, void* k_unrot0,
                                                  size_t k_unrot_stride, float* partial0, size_t partial_stride_floats,
                                                  int n_units, int Hq, int Hkv, int L, int D, int dtype,
                                                  const void* const* key_masks_host, int32_t* key_index_ws,
                                                  const void* const* q_units_host, int64_t q_stride_h, int64_t q_stride_l,
                                                  rtk_stream_t stream) {
    if (!q_units_host)
        return rtk_pivotkv_score_passes_batched(workspace0, workspace_stride, k_unrot0, k_unrot_stride, partial0,
                                                partial_stride_floats, n_units, Hq, Hkv, L, D, dtype, key_masks_host,
                                                key_index_ws, stream);
    const int base = dtype & ~RTK_SCORE_MANY_UNITS;
    RTK_CHECK_ARG(workspace0 && partial0 && k_unrot0 && n_units >= 1, "rtk_pivotkv_score_passes_batched_q: NULL pointer or no units");
    if ((base != RTK_BF16 && base != RTK_F16) || D != HD) {
        set_error("rtk_pivotkv_score_passes_batched_q: bf16 / fp16 payloads with head_dim %d only", HD);
        return RTK_EUNSUPPORTED;
    }
    RTK_CHECK_ARG(q_stride_h > 0 && q_stride_l > 0 && (q_stride_h * 2) % 16 == 0 && (q_stride_l * 2) % 16 == 0 &&
                      (Hq - 1) * q_stride_h * 2 + (int64_t)(L - 1) * q_stride_l * 2 < (1ll << 31),
                  "rtk_pivotkv_score_passes_batched_q: bad query strides");
    for (int u = 0; u < n_units; ++u)
        RTK_CHECK_ARG(q_units_host[u] && ((uintptr_t)q_units_host[u] & 15) == 0,
                      "rtk_pivotkv_score_passes_batched_q: unit %d: queries must be 16-byte aligned", u);
    RTK_CHECK_ARG(((uintptr_t)workspace0 & 255) == 0 && workspace_stride % 256 == 0,
                  "rtk_pivotkv_score_passes_batched_q: workspaces must be 256-byte aligned");
    const ScoreWs w = score_ws(Hq, Hkv, L, D, dtype);
    RTK_CHECK_ARG(n_units == 1 || workspace_stride >= w.total, "rtk_pivotkv_score_passes_batched_q: workspace stride too small");
    const int* key_index = nullptr;
    if (key_masks_host && key_index_ws) {
        bool any = false;
        for (int u = 0; u < n_units; ++u) any = any || key_masks_host[u];
        if (any) {
            for (int u0 = 0; u0 < n_units; u0 += MAX_MASK_UNITS) {
                KeyMasks km;
                const int m = std::min(MAX_MASK_UNITS, n_units - u0);
                for (int u = 0; u < MAX_MASK_UNITS; ++u) km.m[u] = u < m ? (const uint8_t*)key_masks_host[u0 + u] : nullptr;
                RTK_LAUNCH(KID_FINALIZE, key_compact_kernel, dim3(m), dim3(1024), 0, (hipStream_t)stream, km, L,
                           key_index_ws + (size_t)u0 * (L + 1));
            }
            RTK_LAUNCH_CHECK("key_compact_kernel");
            key_index = key_index_ws;
        }
    }
    float dummy_score = 0.f;
    for (int u0 = 0; u0 < n_units; u0 += MAX_Q_UNITS) {   // the pointer table travels as a kernel argument
        const int m = std::min(MAX_Q_UNITS, n_units - u0);
        QView qv;
        memset(&qv, 0, sizeof(qv));
        for (int u = 0; u < m; ++u) qv.unit[u] = (const char*)q_units_host[u0 + u];
        qv.head_stride = (int)(q_stride_h * 2);
        qv.row_pitch = (int)(q_stride_l * 2);
        char* ws_u = (char*)workspace0 + (size_t)u0 * workspace_stride;
        const int rc = score_impl<RTK_BF16>(ws_u, 0, 0, ws_u, 0, 0, Hq, Hkv, L, D, nullptr, nullptr, 1.0f, &dummy_score,
                                            (char*)k_unrot0 + (size_t)u0 * k_unrot_stride, ws_u, w, RTK_SCORE_PASSES,
                                            partial0 + (size_t)u0 * partial_stride_floats, (hipStream_t)stream, m,
                                            workspace_stride, k_unrot_stride, partial_stride_floats,
                                            key_index ? key_index + (size_t)u0 * (L + 1) : nullptr, &qv);
        if (rc) return rc;
    }
    return RTK_OK;
}

extern "C" size_t rtk_pivotkv_score_partials(int Hq, int Hkv, int L, int D, int dtype, int* rs_out) {
    if (Hq < 1 || Hkv < 1 || L < 1 || D < 1) return 0;
    const ScoreWs w = score_ws(Hq, Hkv, L, D, dtype);
    int rs_n = 1;
    if (D == HD) {   // the non-empty row splits score_impl launches
        const int rps = ((((L + w.RS - 1) / w.RS) + TILE_ROWS - 1) / TILE_ROWS) * TILE_ROWS;
        rs_n = (L + rps - 1) / rps;
    }
    if (rs_out) *rs_out = rs_n;
    return (size_t)(w.ref ? Hq : Hkv) * rs_n * L;
}

static int score_stages_impl(const void* q, int64_t q_stride_h, int64_t q_stride_l, const void* k, int64_t k_stride_h,
                             int64_t k_stride_l, int Hq, int Hkv, int L, int D, int dtype, const float* cosv,
                             const float* sinv, float attention_scaling, float* score, void* k_unrot, void* workspace,
                             size_t workspace_bytes, int stages, float* partial_out, const void* key_mask,
                             int32_t* key_index_ws, rtk_stream_t stream);

extern "C" int rtk_pivotkv_score_stages(const void* q, int64_t q_stride_h, int64_t q_stride_l, const void* k,
                                        int64_t k_stride_h, int64_t k_stride_l, int Hq, int Hkv, int L, int D, int dtype,
                                        const float* cosv, const float* sinv, float attention_scaling, float* score,
                                        void* k_unrot, void* workspace, size_t workspace_bytes, int stages,
                                        float* partial_out, rtk_stream_t stream) {
    return score_stages_impl(q, q_stride_h, q_stride_l, k, k_stride_h, k_stride_l, Hq, Hkv, L, D, dtype, cosv, sinv,
                             attention_scaling, score, k_unrot, workspace, workspace_bytes, stages, partial_out, nullptr,
                             nullptr, stream);
}

extern "C" int rtk_pivotkv_score_stages_masked(const void* q, int64_t q_stride_h, int64_t q_stride_l, const void* k,
                                               int64_t k_stride_h, int64_t k_stride_l, int Hq, int Hkv, int L, int D,
                                               int dtype, const float* cosv, const float* sinv, float attention_scaling,
                                               float* score, void* k_unrot, void* workspace, size_t workspace_bytes,
                                               int stages, float* partial_out, const void* key_mask,
                                               int32_t* key_index_ws, rtk_stream_t stream) {
    return score_stages_impl(q, q_stride_h, q_stride_l, k, k_stride_h, k_stride_l, Hq, Hkv, L, D, dtype, cosv, sinv,
                             attention_scaling, score, k_unrot, workspace, workspace_bytes, stages, partial_out, key_mask,
                             key_index_ws, stream);
}

static int score_stages_impl(const void* q, int64_t q_stride_h, int64_t q_stride_l, const void* k, int64_t k_stride_h,
                             int64_t k_stride_l, int Hq, int Hkv, int L, int D, int dtype, const float* cosv,
                             const float* sinv, float attention_scaling, float* score, void* k_unrot, void* workspace,
                             size_t workspace_bytes, int stages, float* partial_out, const void* key_mask,
                             int32_t* key_index_ws, rtk_stream_t stream) {
    const int dtype_full = dtype;              // may carry RTK_SCORE_MANY_UNITS (split policy, see score_ws)
    dtype &= ~RTK_SCORE_MANY_UNITS;
    RTK_CHECK_ARG(q && k && score && workspace, "rtk_pivotkv_score: NULL pointer");
    RTK_CHECK_ARG(Hq >= 1 && Hkv >= 1 && Hq % Hkv == 0, "rtk_pivotkv_score: Hq=%d must be a multiple of Hkv=%d", Hq, Hkv);
    RTK_CHECK_ARG(L >= 1 && D >= 2 && D % 2 == 0, "rtk_pivotkv_score: bad shape L=%d D=%d", L, D);
    RTK_CHECK_ARG((cosv == nullptr) == (sinv == nullptr), "rtk_pivotkv_score: cos and sin must both be given or both NULL");
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_BF16_REFROUND || dtype == RTK_BF16_FAST ||
                      dtype == RTK_F16 || dtype == RTK_F16_REFROUND, "rtk_pivotkv_score: unsupported dtype %d", dtype);
    RTK_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "rtk_pivotkv_score: workspace must be 256-byte aligned");
    RTK_CHECK_ARG(stages > 0 && stages <= 7, "rtk_pivotkv_score: stages mask %d out of range", stages);
    const ScoreWs w = score_ws(Hq, Hkv, L, D, dtype_full);
    if (workspace_bytes < w.total) {
        set_error("rtk_pivotkv_score: workspace %zu < required %zu bytes", workspace_bytes, w.total);
        return RTK_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    // live keys of pass 2 (head_dim 128 kernels only): the columns of masked tokens are not computed, the caller's
    // selection overwrites their score with 1.0 anyway (longvideo_cache.py:272-274)
    const int* key_index = nullptr;
    if (key_mask && key_index_ws && D == HD && (stages & RTK_SCORE_PASSES)) {
        KeyMasks km;
        for (int u = 0; u < MAX_MASK_UNITS; ++u) km.m[u] = u == 0 ? (const uint8_t*)key_mask : nullptr;
        RTK_LAUNCH(KID_FINALIZE, key_compact_kernel, dim3(1), dim3(1024), 0, st, km, L, key_index_ws);
        RTK_LAUNCH_CHECK("key_compact_kernel");
        key_index = key_index_ws;
    }
    if (dtype != RTK_F32)
        return score_impl<RTK_BF16>(q, q_stride_h, q_stride_l, k, k_stride_h, k_stride_l, Hq, Hkv, L, D, cosv, sinv,
                                    attention_scaling, score, k_unrot, (char*)workspace, w, stages, partial_out, st, 1, 0, 0, 0,
                                    key_index);
    return score_impl<RTK_F32>(q, q_stride_h, q_stride_l, k, k_stride_h, k_stride_l, Hq, Hkv, L, D, cosv, sinv,
                               attention_scaling, score, k_unrot, (char*)workspace, w, stages, partial_out, st, 1, 0, 0, 0,
                               key_index);
}

extern "C" int rtk_pivotkv_score(const void* q, int64_t q_stride_h, int64_t q_stride_l, const void* k,
                                 int64_t k_stride_h, int64_t k_stride_l, int Hq, int Hkv, int L, int D, int dtype,
                                 const float* cosv, const float* sinv, float attention_scaling, float* score,
                                 void* k_unrot, void* workspace, size_t workspace_bytes, rtk_stream_t stream) {
    return rtk_pivotkv_score_stages(q, q_stride_h, q_stride_l, k, k_stride_h, k_stride_l, Hq, Hkv, L, D, dtype, cosv, sinv,
                                    attention_scaling, score, k_unrot, workspace, workspace_bytes,
                                    RTK_SCORE_PREPARE | RTK_SCORE_PASSES | RTK_SCORE_FINALIZE, nullptr, stream);
}

template <int DT>
static int prepare_impl(const void* q, int64_t qsh, int64_t qsl, const void* k, int64_t ksh, int64_t ksl, const void* v,
                        int64_t vsh, int64_t vsl, int Hq, int Hkv, int L, int D, const int64_t* pos, int64_t pos_stride,
                        const float* inv_freq, float a, const RowSel& rs, int round_bf16, char* qt, char* kt, void* k_tail,
                        void* v_tail, int64_t tail_sh, int P, int64_t* pos_copy, hipStream_t st, char* k_fast = nullptr,
                        int64_t* shift_row = nullptr, const int64_t* next_prev = nullptr, int* ticket = nullptr,
                        int* status = nullptr) {
    const float a2 = (float)((double)a * (double)a);
    const int div = (a2 == 1.0f) ? 0 : ((DT == RTK_BF16 && bf16_rcp_is_exact(a2)) ? 1 : 2);
    const float rcp = 1.0f / a2;
    // 16-bit dtypes: 4-byte chunks per thread - four times the waves of the 16-byte form, a quarter of the instruction
    // stream each (same-box A/B, profiles/r11_ab_prepare_chunk_width.txt: 20.0 -> 16.1 us at L = 2304, 29.6 -> 28.4 at 6272)
    int nw = 4;
    if constexpr (DT != RTK_F32) nw = RTK_PREP_NW;
    const int VE = nw * 4 / (DT == RTK_F32 ? 4 : 2);
    const int threads = L * (D / 2 / VE);
    static_assert(RTK_PREP_YSPLIT >= 2, "the first y-slice takes k and the LAST one v: one slice would never append v");
    // (+ one column of workgroups when the next layer's id shift rides along: its last one does the shift)
    const dim3 grid((threads + RTK_PREP_BLOCK - 1) / RTK_PREP_BLOCK + (shift_row ? 1 : 0), RTK_PREP_YSPLIT);
    char* kf = nullptr;
    float qscale = 1.f;
    auto launch = [&](auto kern) {
        RTK_LAUNCH(KID_UNROT, kern, grid, dim3(RTK_PREP_BLOCK), 0, st, (const char*)q, qsh, qsl, (const char*)k, ksh, ksl, (const char*)v,
                   vsh, vsl, Hq, Hkv, L, D, pos, pos_stride, inv_freq, a, rs, round_bf16, a2, rcp, qt, kt, (char*)k_tail,
                   (char*)v_tail, tail_sh, P, pos_copy, kf, qscale, shift_row, next_prev, ticket, status);
    };
#define RTK_PREP_NWSEL(DIV, FASTV)                                                            \
    do {                                                                                      \
        if constexpr (DT != RTK_F32) {                                                        \
            if (nw == 2) { launch(prepare_native_kernel<DT, DIV, FASTV, 2>); break; }         \
            if (nw == 1) { launch(prepare_native_kernel<DT, DIV, FASTV, 1>); break; }         \
        }                                                                                     \
        launch(prepare_native_kernel<DT, DIV, FASTV, 4>);                                     \
    } while (0)
    bool done = false;
    if constexpr (DT == RTK_BF16) {
        if (k_fast) {   // RTK_BF16_FAST: q~ as fp16(q~ * log2(e)/sqrt(D)), k~ as bf16 (eviction) and as fp16 (scoring)
            kf = k_fast;
            qscale = 1.4426950408889634f / sqrtf((float)D);
            if (div == 0) RTK_PREP_NWSEL(0, true);
            else if (div == 1) RTK_PREP_NWSEL(1, true);
            else RTK_PREP_NWSEL(2, true);
            done = true;
        }
    }
    if (done) {}
    else if (div == 0) RTK_PREP_NWSEL(0, false);
    else if (div == 1) RTK_PREP_NWSEL(1, false);
    else RTK_PREP_NWSEL(2, false);
#undef RTK_PREP_NWSEL
    RTK_LAUNCH_CHECK("prepare_native_kernel");
    return RTK_OK;
}

extern "C" int rtk_pivotkv_prepare(const void* q, int64_t q_stride_h, int64_t q_stride_l, const void* k,
                                   int64_t k_stride_h, int64_t k_stride_l, const void* v, int64_t v_stride_h,
                                   int64_t v_stride_l, int Hq, int Hkv, int L, int D, int dtype, const int64_t* pos,
                                   int64_t pos_stride, int P, const float* inv_freq, float attention_scaling,
                                   const int* sections_host, int nsec, int round_bf16, void* k_unrot, void* workspace,
                                   size_t workspace_bytes, void* k_tail, void* v_tail, int64_t tail_stride_h,
                                   int64_t* pos_copy, rtk_stream_t stream) {
    return rtk::pivotkv_prepare_shift(q, q_stride_h, q_stride_l, k, k_stride_h, k_stride_l, v, v_stride_h, v_stride_l, Hq, Hkv,
                                      L, D, dtype, pos, pos_stride, P, inv_freq, attention_scaling, sections_host, nsec,
                                      round_bf16, k_unrot, workspace, workspace_bytes, k_tail, v_tail, tail_stride_h, pos_copy,
                                      nullptr, nullptr, nullptr, 0, nullptr, stream);
}

// words of rtk_update_io.ticket: the launch count (word 0) and the run-out latch (word 31) in the first cache line, then
// the arrival counters of the prepare launch, a cache line each
extern "C" size_t rtk_pivotkv_shift_ticket_ints(int L, int D) {
    (void)L; (void)D;   // one line for the launch count, one per counter
    return (size_t)RTK_SHIFT_STRIDE * (1 + RTK_SHIFT_COUNTERS);
}

// rtk_pivotkv_prepare + (shift_row != NULL) the next layer's continuity shift in the same launch: rtk_pivotkv_update's
// RTK_UPDATE_SHIFT_NEXT.  shift_row is the temporal row of `pos` itself, ticket the zeroed device words (counters left zero);
// status (optional, host-visible memory) is incremented if the watcher's bounded wait runs out (see the kernel).
int rtk::pivotkv_prepare_shift(const void* q, int64_t q_stride_h, int64_t q_stride_l, const void* k, int64_t k_stride_h,
                               int64_t k_stride_l, const void* v, int64_t v_stride_h, int64_t v_stride_l, int Hq, int Hkv,
                               int L, int D, int dtype, const int64_t* pos, int64_t pos_stride, int P,
                               const float* inv_freq, float attention_scaling, const int* sections_host, int nsec,
                               int round_bf16, void* k_unrot, void* workspace, size_t workspace_bytes, void* k_tail,
                               void* v_tail, int64_t tail_stride_h, int64_t* pos_copy, int64_t* shift_row,
                               const int64_t* next_prev, int32_t* ticket, int64_t ticket_ints, int32_t* status,
                               rtk_stream_t stream) {
    RTK_CHECK_ARG(!shift_row || (ticket && ticket_ints >= (int64_t)rtk_pivotkv_shift_ticket_ints(L, D)),
                  "rtk_pivotkv_prepare: the in-launch id shift needs rtk_pivotkv_shift_ticket_ints(L, D) zeroed device words");
    const bool k_only = (dtype & RTK_PREPARE_K_ONLY) != 0;   // keep-all chunk: no q~
    dtype &= ~RTK_PREPARE_K_ONLY;
    const int dtype_full = dtype;              // may carry RTK_SCORE_MANY_UNITS: the workspace layout follows the split policy
    dtype &= ~RTK_SCORE_MANY_UNITS;
    RTK_CHECK_ARG(q && k && v && pos && inv_freq && k_unrot && workspace && k_tail && v_tail, "rtk_pivotkv_prepare: NULL pointer");
    RTK_CHECK_ARG(Hq >= 1 && Hkv >= 1 && L >= 1 && D >= 2, "rtk_pivotkv_prepare: bad shape");
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_BF16_FAST || dtype == RTK_F16,
                  "rtk_pivotkv_prepare: unsupported dtype %d", dtype);
    RTK_CHECK_ARG(pos_stride >= L, "rtk_pivotkv_prepare: pos_stride %lld < L %d", (long long)pos_stride, L);
    const bool fast = dtype == RTK_BF16_FAST && !k_only;
    if (fast && D != HD) {
        set_error("rtk_pivotkv_prepare: RTK_BF16_FAST needs head_dim %d", HD);
        return RTK_EUNSUPPORTED;
    }
    RTK_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "rtk_pivotkv_prepare: workspace must be 256-byte aligned");
    const ScoreWs w = score_ws(Hq, Hkv, L, D, dtype_full);
    if (!k_only && workspace_bytes < w.total) {
        set_error("rtk_pivotkv_prepare: workspace %zu < required %zu bytes", workspace_bytes, w.total);
        return RTK_EWORKSPACE;
    }
    if (k_only) Hq = 0;   // the kernel's query loop is empty; k / v take the same path
    const int ve = dtype != RTK_F32 ? 8 : 4, es = dtype != RTK_F32 ? 2 : 4;
    const bool ok = (D % (2 * ve) == 0) && D <= 256 && (q_stride_h * es) % 16 == 0 && (q_stride_l * es) % 16 == 0 &&
                    (k_stride_h * es) % 16 == 0 && (k_stride_l * es) % 16 == 0 && (v_stride_h * es) % 16 == 0 &&
                    (v_stride_l * es) % 16 == 0 && (tail_stride_h * es) % 16 == 0 &&
                    (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)k_unrot | (uintptr_t)k_tail | (uintptr_t)v_tail) & 15) == 0;
    if (!ok) {
        set_error("rtk_pivotkv_prepare: needs 16-byte aligned pointers / strides and head_dim a multiple of %d", 2 * ve);
        return RTK_EUNSUPPORTED;   // callers fall back to rtk_rope_table + rtk_pivotkv_score + rtk_pivotkv_append
    }
    if (!(fits_buffer_offsets(Hq, L, D, q_stride_h, q_stride_l, es) && fits_buffer_offsets(Hkv, L, D, k_stride_h, k_stride_l, es) &&
          fits_buffer_offsets(Hkv, L, D, v_stride_h, v_stride_l, es) && fits_buffer_offsets(Hkv, L, D, tail_stride_h, D, es) &&
          fits_buffer_offsets(Hq > Hkv ? Hq : Hkv, L, D, (int64_t)L * D, D, es))) {
        set_error("rtk_pivotkv_prepare: an operand spans 2 GiB or more (or has a negative stride): 32-bit row offsets do not reach");
        return RTK_EUNSUPPORTED;
    }
    RowSel rs;
    int rc = make_rowsel(rs, P, D, sections_host, nsec, "rtk_pivotkv_prepare");
    if (rc) return rc;
    char* qt = (char*)workspace + w.q_off;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RTK_F16)
        return prepare_impl<RTK_F16>(q, q_stride_h, q_stride_l, k, k_stride_h, k_stride_l, v, v_stride_h, v_stride_l, Hq, Hkv, L,
                                     D, pos, pos_stride, inv_freq, attention_scaling, rs, round_bf16, qt, (char*)k_unrot, k_tail,
                                     v_tail, tail_stride_h, P, pos_copy, st, nullptr, shift_row, next_prev, ticket, status);
    if (dtype != RTK_F32)
        return prepare_impl<RTK_BF16>(q, q_stride_h, q_stride_l, k, k_stride_h, k_stride_l, v, v_stride_h, v_stride_l, Hq, Hkv,
                                      L, D, pos, pos_stride, inv_freq, attention_scaling, rs, round_bf16, qt, (char*)k_unrot,
                                      k_tail, v_tail, tail_stride_h, P, pos_copy, st,
                                      fast ? (char*)workspace + w.k_off : nullptr, shift_row, next_prev, ticket, status);
    return prepare_impl<RTK_F32>(q, q_stride_h, q_stride_l, k, k_stride_h, k_stride_l, v, v_stride_h, v_stride_l, Hq, Hkv, L,
                                 D, pos, pos_stride, inv_freq, attention_scaling, rs, round_bf16, qt, (char*)k_unrot, k_tail,
                                 v_tail, tail_stride_h, P, pos_copy, st, nullptr, shift_row, next_prev, ticket, status);
}

// pivotkv_update.hip — PivotKVCache.update and the per-chunk flush as ONE call each (include/retake_hip.h, ABI 13),
// and the attention prologue: the one kernel that takes a layer's pre-RoPE projections to everything the layer's
// attention and the deferred PivotKV scoring need.
//
// Why: a 2048-frame video is 1,792 updates (qwen2_vl.py:670-720 calls the decoder once per chunk, every layer calls
// cache.update, longvideo_cache.py:217).  At the real Qwen2-VL geometry (L = 2304) the GPU work of one update is ~20 us;
// ~50 us of host work per update (argument marshalling for a 30-argument launch, view construction, dict bookkeeping)
// made the step host-bound.  The argument blocks are bound once per chunk geometry (rtk_pivotkv_batch) and per layer
// (rtk_layer_state); an update is then four pointer stores and one call.
#include <algorithm>
#include <vector>

#include "common.cuh"
#include "variants.h"

namespace rtk {

using f32x2_t = __attribute__((ext_vector_type(2))) float;
using f16x2_t = __attribute__((ext_vector_type(2))) _Float16;
__device__ __forceinline__ uint32_t upd_pack2_f16(float lo, float hi) {   // saturating, like pivotkv_score.hip's pack2_f16
    const f32x2_t v = {__builtin_fminf(__builtin_fmaxf(lo, -65504.f), 65504.f),
                       __builtin_fminf(__builtin_fmaxf(hi, -65504.f), 65504.f)};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}

// ------------------------------------------------------------------------------------------------
// Attention prologue (qwen2_vl.py:55-86 / llava_onevision.py:59-141 + longvideo_cache.py:238, :248-259), from the
// PRE-RoPE projections q0 [Hq,L,D], k0, v0 [Hkv,L,D] (strided: the [L, H*D] layout the projections produce):
//   ids      t' = t + (prev + 1 - t[0]) on the temporal row (the continuity shift, qwen2_vl.py:68-73); a copy of the
//            shifted ids goes to pos_copy for the deferred selection
//   tables   cos / sin of the token's ids in registers: rope_chunk (= rtk_rope_table's arithmetic: fp32 id * inv_freq,
//            correctly rounded sin / cos, * attention_scaling, rounded to the model dtype like the rotary module's
//            `.to(x.dtype)`), M-RoPE rows picked per channel (:68-74)
//   q_rot    (q0*cos) + (rotate_half(q0)*sin), one rounding per torch op (apply_multimodal_rotary_pos_emb :80-81)
//            -> the layer's attention; may alias q0 (every address is read and written by one thread only)
//   k tail   the same rotation of k0, appended to the cache tail; v tail: v0 (:238)
//   q~, k~   the score passes' operands.  The reference un-rotates the rotated tensors (:248-259); un-rotating a
//            rotation returns the pre-RoPE value up to the rounding of the round trip, so q~ := q0 and
//            k~ := k0 - copies into the contiguous [H,L,D] layout the matrix passes stream.
// One thread = one token x one NW-word chunk pair (d, d + D/2); blockIdx.y splits the query heads, y = 0 also takes
// k, the last y takes v.  Same software pipeline as prepare_native_kernel: the rows of head batch b+1 are requested
// before batch b is rotated and stored, and the first batch before the table arithmetic.
// ------------------------------------------------------------------------------------------------
template <int DT, bool FAST, int NW, bool RT = false>
__global__ __launch_bounds__(RTK_PREP_BLOCK) void prologue_kernel(const char* q, int64_t q_sh, int64_t q_sl,
                                                      const char* __restrict__ k, int64_t k_sh, int64_t k_sl,
                                                      const char* __restrict__ v, int64_t v_sh, int64_t v_sl,
                                                      int Hq, int Hkv, int L, int D,
                                                      const int64_t* pos, int64_t pos_ld,
                                                      const int64_t* __restrict__ prev,
                                                      const float* __restrict__ inv_freq, float scaling, RowSel rs,
                                                      int round_mode, char* q_rot, int64_t qr_sh, int64_t qr_sl,
                                                      char* __restrict__ q_out, char* __restrict__ k_out,
                                                      char* __restrict__ k_tail, char* __restrict__ v_tail,
                                                      int64_t tail_sh, int P, int64_t* __restrict__ pos_copy,
                                                      int64_t pos_copy_ld, char* __restrict__ k_fast, float qscale,
                                                      int64_t* shift_back, float a2, float rcp_a2, int div) {
    using V = Vec16<DT>;
    static_assert(NW == 4 || ((NW == 2 || NW == 1) && DT != RTK_F32), "8- / 4-byte chunks: 16-bit dtypes only");
    constexpr int ES = 16 / V::VE;          // bytes per element
    constexpr int VE = 4 * NW / ES;         // elements per thread and row half
    using W = WV<NW>;
    const int h2 = D / 2, lpr = h2 / VE;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= L * lpr) return;
    const int l = id / lpr, d = (id - l * lpr) * VE;
    // continuity shift: the whole temporal row moves so that its first id follows the layer's last cached id.  The row's
    // first id is read ONCE per thread (a relaxed atomic load the compiler can neither split nor repeat): a one-token
    // segment's launch stores the shifted id back into pos[0] while other workgroups may still be reading it, and both
    // the delta and the token's own id must come from the same value (old: delta = prev + 1 - t0, id = prev + 1; new:
    // delta = 0, id = prev + 1).
    const long long t0 = (long long)__hip_atomic_load(pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long delta = (prev ? (long long)prev[0] : -1ll) + 1 - t0;
    // pos_ld == 0: the three M-RoPE rows are ONE row seen three times (HF's decode ids, `.expand(3, -1, -1)`,
    // qwen2_vl.py:589): the reference's in-place shift of row 0 moves the shared storage, i.e. t, h and w together
    const bool rows_alias = pos_ld == 0;
    long long ids[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int pr = min(p, P - 1);
        const bool temporal = pr == 0 || rows_alias;
        const long long raw = (temporal && l == 0) ? t0 : (long long)pos[(size_t)pr * pos_ld + l];
        ids[p] = raw + (temporal ? delta : 0ll);
    }
    if (pos_copy && blockIdx.y == 0 && d == 0)
        for (int p = 0; p < P; ++p) pos_copy[(size_t)p * pos_copy_ld + l] = ids[p];
    // a one-token segment (decode): the caller's temporal id is shifted in place by this launch (qwen2_vl.py:73) - the
    // shift is idempotent (afterwards t[0] == prev + 1, so a thread that reads the new value computes delta 0 and the
    // same id) and a single aligned 8-byte atomic store cannot be seen torn, so no workgroup order is needed
    if (shift_back && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        __hip_atomic_store(shift_back, (int64_t)ids[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    constexpr int HU = RTK_PREP_HU;
    const int ny = gridDim.y, qper = (Hq + ny - 1) / ny;
#if RTK_PREP_UBASE
    const int qb = uniform_int(min((int)blockIdx.y * qper, Hq)), qe = uniform_int(min(qb + qper, Hq));   // (head loops in SGPRs)
#else
    const int qb = min((int)blockIdx.y * qper, Hq), qe = min(qb + qper, Hq);
#endif
    const bool is_k = blockIdx.y == 0;
    const bool has_kv = is_k || (int)blockIdx.y == ny - 1;
    const char* src = is_k ? k : v;
    const int64_t sh = is_k ? k_sh : v_sh, sl = is_k ? k_sl : v_sl;
    char* tail = is_k ? k_tail : v_tail;
    const int nkv = has_kv ? Hkv : 0;
    W lo[HU], hi[HU], lon[HU], hin[HU];
#if RTK_PREP_UBASE
    // a row's address = descriptor (tensor base) + soffset (the head: wave-uniform, a scalar multiply) + voffset (this
    // thread's byte offset inside a head, computed once); the launcher has checked that every extent fits 31 bits
    const uint32_t off_q = (uint32_t)(((int64_t)l * q_sl + d) * ES), off_kv = (uint32_t)(((int64_t)l * sl + d) * ES);
    const uint32_t off_qr = (uint32_t)(((int64_t)l * qr_sl + d) * ES);
    const uint32_t off_o = (uint32_t)(((int64_t)l * D + d) * ES), half = (uint32_t)(h2 * ES);
    const uint32_t off_q2 = off_q + half, off_kv2 = off_kv + half, off_o2 = off_o + half, off_qr2 = off_qr + half;
    const __amdgpu_buffer_rsrc_t r_q = buf_rsrc(q), r_src = buf_rsrc(src), r_qo = buf_rsrc(q_out), r_ko = buf_rsrc(k_out),
                                 r_tail = buf_rsrc(tail), r_kf = buf_rsrc(k_fast), r_qr = buf_rsrc(q_rot);
    const uint32_t hs_q = (uint32_t)(q_sh * ES), hs_kv = (uint32_t)(sh * ES), hs_o = (uint32_t)((int64_t)L * D * ES),
                   hs_t = (uint32_t)(tail_sh * ES), hs_qr = (uint32_t)(qr_sh * ES);
    auto soff = [](int h, uint32_t hs) { return (uint32_t)uniform_int((int)((uint32_t)h * hs)); };
#endif
    auto load_q = [&](W* a, W* b, int hb) {
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            const int h = min(hb + u, qe - 1);
#if RTK_PREP_UBASE
            a[u] = buf_load<NW>(r_q, off_q, soff(h, hs_q));
            b[u] = buf_load<NW>(r_q, off_q2, soff(h, hs_q));
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
            a[u] = buf_load<NW>(r_src, off_kv, soff(h, hs_kv));
            b[u] = buf_load<NW>(r_src, off_kv2, soff(h, hs_kv));
#else
            const char* row = src + ((size_t)h * sh + (size_t)l * sl) * ES;
            a[u] = *(const W*)(row + (size_t)d * ES);
            b[u] = *(const W*)(row + (size_t)(d + h2) * ES);
#endif
        }
    };
    float pid[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) pid[p] = (float)ids[p];   // position_ids.float() inside the rotary module
    if (qb < qe) load_q(lo, hi, qb);
    else if (nkv) load_kv(lo, hi, 0);
    float c1[VE], s1[VE], c2[VE], s2[VE];
    rope_chunk<VE>(inv_freq, rs, d, h2, pid, scaling, round_mode, c1, s1, c2, s2);
    // (x*cos) + (rotate_half(x)*sin) for one head's chunk pair, one rounding per torch op, no fma contraction;
    // rotate_half(x)[d] = -x2, rotate_half(x)[d + h2] = x1
    auto rot = [&](const W& lo, const W& hi, W& olo, W& ohi) {
        if constexpr (DT != RTK_F32) {
            using Hh = H16<DT>;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float x1a = Hh::lo(lo.w[w]), x1b = Hh::hi(lo.w[w]), x2a = Hh::lo(hi.w[w]), x2b = Hh::hi(hi.w[w]);
                const int e = 2 * w;
                const uint32_t p1 = Hh::pack2(x1a * c1[e], x1b * c1[e + 1]);
                const uint32_t n1 = Hh::pack2(-x2a * s1[e], -x2b * s1[e + 1]);
                const uint32_t p2 = Hh::pack2(x2a * c2[e], x2b * c2[e + 1]);
                const uint32_t n2 = Hh::pack2(x1a * s2[e], x1b * s2[e + 1]);
                olo.w[w] = Hh::pack2(Hh::lo(p1) + Hh::lo(n1), Hh::hi(p1) + Hh::hi(n1));
                ohi.w[w] = Hh::pack2(Hh::lo(p2) + Hh::lo(n2), Hh::hi(p2) + Hh::hi(n2));
            }
        } else {
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const float x1 = __uint_as_float(lo.w[e]), x2 = __uint_as_float(hi.w[e]);
                olo.w[e] = __float_as_uint(__fadd_rn(__fmul_rn(x1, c1[e]), __fmul_rn(-x2, s1[e])));
                ohi.w[e] = __float_as_uint(__fadd_rn(__fmul_rn(x2, c2[e]), __fmul_rn(x1, s2[e])));
            }
        }
    };
    // RT (reference operands): x~ = ((x*cos) - (rotate_half(x)*sin)) / a^2 of the ROTATED pair, one rounding per torch op
    // (longvideo_cache.py:76-78) - the un-rotation the reference applies to what its attention handed over; same
    // arithmetic as prepare_native_kernel's.  div (uniform): 0 a^2 == 1, 1 multiply by the reciprocal (bf16, proven
    // identical for every bf16 input: bf16_rcp_is_exact), 2 IEEE division.
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
                if (div == 1) {
                    t1 = Hh::pack2(Hh::lo(t1) * rcp_a2, Hh::hi(t1) * rcp_a2);
                    t2 = Hh::pack2(Hh::lo(t2) * rcp_a2, Hh::hi(t2) * rcp_a2);
                } else if (div == 2) {
                    t1 = Hh::pack2(__fdiv_rn(Hh::lo(t1), a2), __fdiv_rn(Hh::hi(t1), a2));
                    t2 = Hh::pack2(__fdiv_rn(Hh::lo(t2), a2), __fdiv_rn(Hh::hi(t2), a2));
                }
                olo.w[w] = t1;
                ohi.w[w] = t2;
            }
        } else {
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const float x1 = __uint_as_float(lo.w[e]), x2 = __uint_as_float(hi.w[e]);
                float o1 = __fsub_rn(__fmul_rn(x1, c1[e]), __fmul_rn(-x2, s1[e]));
                float o2 = __fsub_rn(__fmul_rn(x2, c2[e]), __fmul_rn(x1, s2[e]));
                if (div != 0) {
                    o1 = __fdiv_rn(o1, a2);
                    o2 = __fdiv_rn(o2, a2);
                }
                olo.w[e] = __float_as_uint(o1);
                ohi.w[e] = __float_as_uint(o2);
            }
        }
    };
    auto to_f16 = [&](const W& x, float scale) {   // bf16 pairs -> fp16 pairs of (value * scale): RTK_BF16_FAST operands
        W o;
#pragma unroll
        for (int w = 0; w < NW; ++w)
            o.w[w] = upd_pack2_f16(__uint_as_float(x.w[w] << 16) * scale, __uint_as_float(x.w[w] & 0xffff0000u) * scale);
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
            rot(lo[u], hi[u], olo, ohi);
            if (q_out) {   // q~ := q0, or (RT) the un-rotation of the rotated row (keep-all chunks are not scored: no q~)
                W ql = lo[u], qh = hi[u];
                if constexpr (RT) unrot(olo, ohi, ql, qh);
                if constexpr (FAST) {
                    ql = to_f16(ql, qscale);
                    qh = to_f16(qh, qscale);
                }
#if RTK_PREP_UBASE
                buf_store<NW>(ql, r_qo, off_o, soff(h, hs_o));
                buf_store<NW>(qh, r_qo, off_o2, soff(h, hs_o));
#else
                char* orow = q_out + ((size_t)h * L + l) * D * ES;
                *(W*)(orow + (size_t)d * ES) = ql;
                *(W*)(orow + (size_t)(d + h2) * ES) = qh;
#endif
            }
#if RTK_PREP_UBASE
            buf_store<NW>(olo, r_qr, off_qr, soff(h, hs_qr));
            buf_store<NW>(ohi, r_qr, off_qr2, soff(h, hs_qr));
#else
            char* rrow = q_rot + ((size_t)h * qr_sh + (size_t)l * qr_sl) * ES;
            *(W*)(rrow + (size_t)d * ES) = olo;
            *(W*)(rrow + (size_t)(d + h2) * ES) = ohi;
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
            const uint32_t so_t = soff(h, hs_t), so_o = soff(h, hs_o);
            if (is_k) {
                W olo, ohi;
                rot(lo[u], hi[u], olo, ohi);
                W kl = lo[u], kh = hi[u];
                if constexpr (RT) unrot(olo, ohi, kl, kh);
                if (k_out) {   // k~ := k0 / (RT) un-rotated tail row (a plain append - text segments, decode - scores nothing: no k~)
                    buf_store<NW>(kl, r_ko, off_o, so_o);
                    buf_store<NW>(kh, r_ko, off_o2, so_o);
                }
                if constexpr (FAST) {
                    buf_store<NW>(to_f16(kl, 1.f), r_kf, off_o, so_o);
                    buf_store<NW>(to_f16(kh, 1.f), r_kf, off_o2, so_o);
                }
                buf_store<NW>(olo, r_tail, off_o, so_t);
                buf_store<NW>(ohi, r_tail, off_o2, so_t);
            } else {
                buf_store<NW>(lo[u], r_tail, off_o, so_t);
                buf_store<NW>(hi[u], r_tail, off_o2, so_t);
            }
#else
            char* trow = tail + ((size_t)h * tail_sh + (size_t)l * D) * ES;
            if (is_k) {
                W olo, ohi;
                rot(lo[u], hi[u], olo, ohi);
                W kl = lo[u], kh = hi[u];
                if constexpr (RT) unrot(olo, ohi, kl, kh);
                if (k_out) {   // k~ := k0 / (RT) un-rotated tail row (a plain append - text segments, decode - scores nothing: no k~)
                    char* orow = k_out + ((size_t)h * L + l) * D * ES;
                    *(W*)(orow + (size_t)d * ES) = kl;
                    *(W*)(orow + (size_t)(d + h2) * ES) = kh;
                }
                if constexpr (FAST) {
                    char* frow = k_fast + ((size_t)h * L + l) * D * ES;
                    *(W*)(frow + (size_t)d * ES) = to_f16(kl, 1.f);
                    *(W*)(frow + (size_t)(d + h2) * ES) = to_f16(kh, 1.f);
                }
                *(W*)(trow + (size_t)d * ES) = olo;
                *(W*)(trow + (size_t)(d + h2) * ES) = ohi;
            } else {
                *(W*)(trow + (size_t)d * ES) = lo[u];
                *(W*)(trow + (size_t)(d + h2) * ES) = hi[u];
            }
#endif
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            lo[u] = lon[u];
            hi[u] = hin[u];
        }
    }
}

struct PrologueGeom {
    int Hq, Hkv, L, D, P, round_mode;
    const float* inv_freq;
    float scaling;
};

template <int DT>
static int prologue_launch(const PrologueGeom& g, const rtk_update_io* io, const RowSel& rs, const int64_t* prev,
                           char* q_out, char* k_out, char* k_tail, char* v_tail, int64_t tail_sh, int64_t* pos_copy,
                           int64_t pos_copy_ld, char* k_fast, hipStream_t st, int64_t* shift_back = nullptr,
                           bool roundtrip = false) {
    // reference operands (roundtrip): the divisor of the un-rotation, as torch evaluates `/ attention_scaling ** 2` on
    // a tensor of the model dtype - fp32 opmath with the Python double rounded to fp32 (prepare_impl's rule)
    const float a2 = (float)((double)g.scaling * (double)g.scaling);
    const int div = !roundtrip || a2 == 1.0f ? 0 : ((DT == RTK_BF16 && bf16_rcp_is_exact(a2)) ? 1 : 2);
    const float rcp_a2 = 1.0f / a2;
    int nw = 4;
    if constexpr (DT != RTK_F32) nw = RTK_PREP_NW;
    const int VE = nw * 4 / (DT == RTK_F32 ? 4 : 2);
    const int threads = g.L * (g.D / 2 / VE);
    const dim3 grid((threads + RTK_PREP_BLOCK - 1) / RTK_PREP_BLOCK, RTK_PREP_YSPLIT);
    const float qscale = k_fast ? 1.4426950408889634f / sqrtf((float)g.D) : 1.f;
    auto launch = [&](auto kern) {
        RTK_LAUNCH(KID_PROLOGUE, kern, grid, dim3(RTK_PREP_BLOCK), 0, st, (const char*)io->q, io->q_stride_h, io->q_stride_l,
                   (const char*)io->k, io->k_stride_h, io->k_stride_l, (const char*)io->v, io->v_stride_h, io->v_stride_l,
                   g.Hq, g.Hkv, g.L, g.D, io->pos, io->pos_stride, prev, g.inv_freq, g.scaling, rs, g.round_mode,
                   (char*)io->q_rot, io->qr_stride_h, io->qr_stride_l, q_out, k_out, k_tail, v_tail, tail_sh, g.P, pos_copy,
                   pos_copy_ld, k_fast, qscale, shift_back, a2, rcp_a2, div);
    };
#define RTK_PRO_NWSEL(FASTV, RTV)                                                         \
    do {                                                                                  \
        if constexpr (DT != RTK_F32) {                                                    \
            if (nw == 2) { launch(prologue_kernel<DT, FASTV, 2, RTV>); break; }           \
            if (nw == 1) { launch(prologue_kernel<DT, FASTV, 1, RTV>); break; }           \
        }                                                                                 \
        launch(prologue_kernel<DT, FASTV, 4, RTV>);                                       \
    } while (0)
    bool done = false;
    if constexpr (DT == RTK_BF16) {
        if (k_fast) {
            if (roundtrip) RTK_PRO_NWSEL(true, true);
            else RTK_PRO_NWSEL(true, false);
            done = true;
        }
    }
    if (!done) {
        if (roundtrip) RTK_PRO_NWSEL(false, true);
        else RTK_PRO_NWSEL(false, false);
    }
#undef RTK_PRO_NWSEL
    RTK_LAUNCH_CHECK("prologue_kernel");
    return RTK_OK;
}

static inline size_t esize(int dtype) { return dtype == RTK_F32 ? 4 : 2; }

}  // namespace rtk

using namespace rtk;

static int check_batch(const rtk_pivotkv_batch* b, const char* who) {
    RTK_CHECK_ARG(b, "%s: NULL batch", who);
    RTK_CHECK_ARG(b->Hq >= 1 && b->Hkv >= 1 && b->Hq % b->Hkv == 0 && b->L >= 1 && b->D >= 2 && b->slots >= 1,
                  "%s: bad geometry", who);
    RTK_CHECK_ARG(b->keep >= 1 && b->keep <= b->L, "%s: keep=%d out of range for L=%d", who, b->keep, b->L);
    RTK_CHECK_ARG(b->P == 0 || b->P == 1 || b->P == 3, "%s: P must be 0, 1 or 3, got %d", who, b->P);
    RTK_CHECK_ARG(b->dtype == RTK_F32 || b->dtype == RTK_BF16 || b->dtype == RTK_F16, "%s: unsupported dtype %d", who, b->dtype);
    RTK_CHECK_ARG(b->nsec >= 0 && b->nsec <= 8, "%s: nsec %d out of range", who, b->nsec);
    RTK_CHECK_ARG(b->keep_idx && (b->keep_all || b->v_stage || b->compact_sync), "%s: NULL batch buffer", who);
    RTK_CHECK_ARG(b->keep_all || (b->score_ws && b->partials && b->score && b->sel_ws), "%s: NULL scoring buffer", who);
    return RTK_OK;
}

extern "C" int rtk_pivotkv_update(rtk_pivotkv_batch* b, rtk_layer_state* ls, int slot, const rtk_update_io* io,
                                  rtk_stream_t stream) {
    int rc = check_batch(b, "rtk_pivotkv_update");
    if (rc) return rc;
    RTK_CHECK_ARG(ls && io, "rtk_pivotkv_update: NULL layer state or io block");
    RTK_CHECK_ARG(slot >= 0 && slot < b->slots, "rtk_pivotkv_update: slot %d outside the batch (%d slots)", slot, b->slots);
    RTK_CHECK_ARG(io->q && io->k && io->v, "rtk_pivotkv_update: NULL q / k / v");
    RTK_CHECK_ARG(ls->k && ls->v && ls->length >= 0 && ls->length + b->L <= ls->cap,
                  "rtk_pivotkv_update: the layer's cache has no room for the chunk (length %lld + %d > cap %lld)",
                  (long long)ls->length, b->L, (long long)ls->cap);
    RTK_CHECK_ARG(ls->pending == 0, "rtk_pivotkv_update: the layer still has a pending chunk (flush first)");
    if (!b->reforge || !b->inv_freq || b->P == 0 || !io->pos || !b->k_unrot || !b->score_ws) {
        set_error("rtk_pivotkv_update: needs pos_embed_reforge, position ids and an inv_freq rotary (use the per-stage calls)");
        return RTK_EUNSUPPORTED;
    }
    const size_t es = esize(b->dtype);
    const int L = b->L, D = b->D, Hkv = b->Hkv;
    char* k_tail = (char*)ls->k + (size_t)ls->length * D * es;
    char* v_tail = (char*)ls->v + (size_t)ls->length * D * es;
    const int64_t tail_sh = ls->cap * D;
    char* ws = (char*)b->score_ws + (size_t)slot * b->score_ws_stride;
    char* k_unrot = (char*)b->k_unrot + (size_t)slot * Hkv * L * D * es;
    int64_t* pos_copy = b->pos_old ? b->pos_old + (size_t)slot * b->P * L : nullptr;
    const int score_base = b->score_dtype & 0xFF;
    hipStream_t st = (hipStream_t)stream;
    if (io->flags & RTK_UPDATE_PRE_ROPE) {
        RTK_CHECK_ARG(io->q_rot, "rtk_pivotkv_update: RTK_UPDATE_PRE_ROPE needs q_rot");
        const bool roundtrip = (io->flags & RTK_UPDATE_ROUNDTRIP) != 0;
        if ((score_base == RTK_BF16_REFROUND || score_base == RTK_F16_REFROUND) && !roundtrip) {
            set_error("rtk_pivotkv_update: score_rounding='reference' scores the reference's round-tripped q~ / k~ "
                      "(set RTK_UPDATE_ROUNDTRIP, or rotate first and update without RTK_UPDATE_PRE_ROPE)");
            return RTK_EUNSUPPORTED;
        }
        RTK_CHECK_ARG(!(roundtrip && (io->flags & RTK_UPDATE_Q_IN_PLACE)),
                      "rtk_pivotkv_update: RTK_UPDATE_ROUNDTRIP makes q~ differ from q0: it cannot be scored in place");
        const int ve = b->dtype != RTK_F32 ? 8 : 4;
        const bool ok = (D % (2 * ve) == 0) && D <= 256 && (io->q_stride_h * es) % 16 == 0 && (io->q_stride_l * es) % 16 == 0 &&
                        (io->k_stride_h * es) % 16 == 0 && (io->k_stride_l * es) % 16 == 0 && (io->v_stride_h * es) % 16 == 0 &&
                        (io->v_stride_l * es) % 16 == 0 && (io->qr_stride_h * es) % 16 == 0 && (io->qr_stride_l * es) % 16 == 0 &&
                        (tail_sh * es) % 16 == 0 &&
                        (((uintptr_t)io->q | (uintptr_t)io->k | (uintptr_t)io->v | (uintptr_t)io->q_rot | (uintptr_t)k_unrot |
                          (uintptr_t)k_tail | (uintptr_t)v_tail | (uintptr_t)ws) & 15) == 0;
        if (!ok) {
            set_error("rtk_pivotkv_update: the prologue needs 16-byte aligned pointers / strides and head_dim a multiple of %d", 2 * ve);
            return RTK_EUNSUPPORTED;
        }
        if (!(fits_buffer_offsets(b->Hq, L, D, io->q_stride_h, io->q_stride_l, es) && fits_buffer_offsets(b->Hq, L, D, io->qr_stride_h, io->qr_stride_l, es) &&
              fits_buffer_offsets(Hkv, L, D, io->k_stride_h, io->k_stride_l, es) && fits_buffer_offsets(Hkv, L, D, io->v_stride_h, io->v_stride_l, es) &&
              fits_buffer_offsets(Hkv, L, D, tail_sh, D, es) && fits_buffer_offsets(b->Hq, L, D, (int64_t)L * D, D, es))) {
            set_error("rtk_pivotkv_update: an operand spans 2 GiB or more (or has a negative stride): 32-bit row offsets do not reach");
            return RTK_EUNSUPPORTED;
        }
        if (b->P == 3 && io->pos_stride < L && io->pos_stride != 0) {   // partially overlapping id rows: the eager route
            set_error("rtk_pivotkv_update: position-id rows overlap (pos_stride %lld < L %d)", (long long)io->pos_stride, L);
            return RTK_EUNSUPPORTED;
        }
        const size_t need = rtk_pivotkv_score_workspace_bytes(b->Hq, Hkv, L, D, b->score_dtype);
        if (!b->keep_all && b->score_ws_bytes < need) {
            set_error("rtk_pivotkv_update: workspace %zu < required %zu bytes", (size_t)b->score_ws_bytes, need);
            return RTK_EWORKSPACE;
        }
        RowSel rs;
        rc = make_rowsel(rs, b->P, D, b->nsec ? b->sections : nullptr, b->nsec, "rtk_pivotkv_update");
        if (rc) return rc;
        const int64_t* prev = (ls->pos && ls->pos_len > 0) ? ls->pos + (ls->pos_len - 1) : nullptr;
        char* q_out = b->keep_all ? nullptr : ws;   // q~ at offset 0 of the slot's score workspace
        if ((io->flags & RTK_UPDATE_Q_IN_PLACE) && !b->keep_all) {
            // q~ IS io->q: the batched passes of the flush stream it from where it lies
            if (!b->q_units || !b->batched_passes || score_base == RTK_BF16_FAST || io->q_rot == io->q) {
                set_error("rtk_pivotkv_update: RTK_UPDATE_Q_IN_PLACE needs batch.q_units, the batched passes, exact score "
                          "arithmetic and rotated queries that go elsewhere");
                return RTK_EINVAL;
            }
            for (int u = 0; u < b->slots; ++u)   // one pointer table, one pair of strides per flush
                if (u != slot && b->q_units[u] && (b->q_stride_h != io->q_stride_h || b->q_stride_l != io->q_stride_l)) {
                    set_error("rtk_pivotkv_update: the pending units' queries have other strides (flush first)");
                    return RTK_EUNSUPPORTED;
                }
            b->q_units[slot] = io->q;
            b->q_stride_h = io->q_stride_h;
            b->q_stride_l = io->q_stride_l;
            q_out = nullptr;
        } else if (b->q_units) {
            b->q_units[slot] = nullptr;
        }
        // RTK_BF16_FAST: a second, fp16 copy of k~ inside the workspace, right behind q~ (score_ws: k_off)
        char* k_fast = nullptr;
        if (score_base == RTK_BF16_FAST && !b->keep_all)
            k_fast = ws + (((size_t)b->Hq * L * D * es + 255) & ~(size_t)255);
        const PrologueGeom pg{b->Hq, Hkv, L, D, b->P, b->round_mode, b->inv_freq, b->attention_scaling};
        if (b->dtype == RTK_F16)
            rc = prologue_launch<RTK_F16>(pg, io, rs, prev, q_out, k_unrot, k_tail, v_tail, tail_sh, pos_copy, L, nullptr, st, nullptr, roundtrip);
        else if (b->dtype == RTK_BF16)
            rc = prologue_launch<RTK_BF16>(pg, io, rs, prev, q_out, k_unrot, k_tail, v_tail, tail_sh, pos_copy, L, k_fast, st, nullptr, roundtrip);
        else
            rc = prologue_launch<RTK_F32>(pg, io, rs, prev, q_out, k_unrot, k_tail, v_tail, tail_sh, pos_copy, L, nullptr, st, nullptr, roundtrip);
        if (rc) return rc;
    } else {
        if (b->q_units) b->q_units[slot] = nullptr;
        const int dt = b->prep_dtype | (b->keep_all ? RTK_PREPARE_K_ONLY : 0);
        // RTK_UPDATE_SHIFT_NEXT: the launch also leaves the caller's temporal row shifted for the NEXT layer (qwen2_vl.py:68-73)
        const bool shift_next = (io->flags & RTK_UPDATE_SHIFT_NEXT) != 0;
        if (shift_next && !io->ticket) {
            set_error("rtk_pivotkv_update: RTK_UPDATE_SHIFT_NEXT needs io->ticket (rtk_pivotkv_shift_ticket_ints zeroed device words)");
            return RTK_EINVAL;
        }
        if (shift_next && !b->keep_all && !b->batched_passes) {
            // the per-unit passes below may still decline (RTK_EUNSUPPORTED) AFTER the prepare launch has shifted the ids:
            // a caller that falls back to the stage-by-stage route would then re-prepare this layer from the NEXT layer's
            // ids.  Decline first, with nothing launched.
            set_error("rtk_pivotkv_update: RTK_UPDATE_SHIFT_NEXT needs the chunk-batched passes or a keep-all batch");
            return RTK_EUNSUPPORTED;
        }
        rc = pivotkv_prepare_shift(io->q, io->q_stride_h, io->q_stride_l, io->k, io->k_stride_h, io->k_stride_l, io->v,
                                   io->v_stride_h, io->v_stride_l, b->Hq, Hkv, L, D, dt, io->pos, io->pos_stride, b->P,
                                   b->inv_freq, b->attention_scaling, b->nsec ? b->sections : nullptr, b->nsec, b->round_mode,
                                   k_unrot, ws, b->score_ws_bytes, k_tail, v_tail, tail_sh, pos_copy,
                                   shift_next ? (int64_t*)io->pos : nullptr, io->next_prev, io->ticket, io->ticket_ints,
                                   shift_next ? io->status : nullptr, stream);
        if (rc) return rc;
    }
    if (!b->keep_all && !b->batched_passes) {
        // shapes outside the chunk-batched passes (fp32 payloads, other head dims): the unit's two matrix passes now
        float* part = b->partials + (size_t)slot * b->partial_floats;
        const bool live = b->skip_masked && ls->mask && b->key_index;
        rc = rtk_pivotkv_score_stages_masked(ws, 0, 0, ws, 0, 0, b->Hq, Hkv, L, D, b->score_dtype, nullptr, nullptr, 1.0f,
                                             b->score + (size_t)slot * L, k_unrot, ws, b->score_ws_bytes, RTK_SCORE_PASSES,
                                             part, live ? ls->mask : nullptr,
                                             live ? b->key_index + (size_t)slot * (L + 1) : nullptr, stream);
        if (rc) return rc;
    }
    ls->pending = L;
    ls->pending_keep = b->keep;
    return RTK_OK;
}

extern "C" int rtk_pivotkv_append_rope(rtk_layer_state* ls, const rtk_update_io* io, int Hq, int Hkv, int n, int D,
                                       int dtype, int P, const float* inv_freq, float attention_scaling,
                                       const int* sections_host, int nsec, int round_mode, int shift_ids_in_place,
                                       rtk_stream_t stream) {
    RTK_CHECK_ARG(ls && io && io->q && io->k && io->v && io->q_rot && io->pos && inv_freq, "rtk_pivotkv_append_rope: NULL pointer");
    RTK_CHECK_ARG(Hq >= 1 && Hkv >= 1 && n >= 1 && D >= 2, "rtk_pivotkv_append_rope: bad shape");
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_F16, "rtk_pivotkv_append_rope: unsupported dtype %d", dtype);
    RTK_CHECK_ARG(P == 1 || P == 3, "rtk_pivotkv_append_rope: P must be 1 or 3, got %d", P);
    RTK_CHECK_ARG(ls->k && ls->v && ls->pending == 0 && ls->length >= 0 && ls->length + n <= ls->cap,
                  "rtk_pivotkv_append_rope: the layer's cache has no room for %d rows (or a chunk is pending)", n);
    RTK_CHECK_ARG(ls->pos && ls->pos_len + n <= ls->pos_cap, "rtk_pivotkv_append_rope: the position cache has no room for %d ids", n);
    if (P == 3 && io->pos_stride < n && io->pos_stride != 0) {   // rows that overlap partially: not a layout torch hands out;
        set_error("rtk_pivotkv_append_rope: position-id rows overlap (pos_stride %lld < n %d)", (long long)io->pos_stride, n);
        return RTK_EUNSUPPORTED;                                     // fully aliased rows (stride 0, `.expand(3, ..)`) are served
    }
    const size_t es = esize(dtype);
    char* k_tail = (char*)ls->k + (size_t)ls->length * D * es;
    char* v_tail = (char*)ls->v + (size_t)ls->length * D * es;
    const int64_t tail_sh = ls->cap * D;
    const int ve = dtype != RTK_F32 ? 8 : 4;
    const bool ok = (D % (2 * ve) == 0) && D <= 256 && (io->q_stride_h * es) % 16 == 0 && (io->q_stride_l * es) % 16 == 0 &&
                    (io->k_stride_h * es) % 16 == 0 && (io->k_stride_l * es) % 16 == 0 && (io->v_stride_h * es) % 16 == 0 &&
                    (io->v_stride_l * es) % 16 == 0 && (io->qr_stride_h * es) % 16 == 0 && (io->qr_stride_l * es) % 16 == 0 &&
                    (tail_sh * es) % 16 == 0 &&
                    (((uintptr_t)io->q | (uintptr_t)io->k | (uintptr_t)io->v | (uintptr_t)io->q_rot | (uintptr_t)k_tail |
                      (uintptr_t)v_tail) & 15) == 0;
    if (!ok) {
        set_error("rtk_pivotkv_append_rope: needs 16-byte aligned pointers / strides and head_dim a multiple of %d", 2 * ve);
        return RTK_EUNSUPPORTED;
    }
    if (!(fits_buffer_offsets(Hq, n, D, io->q_stride_h, io->q_stride_l, es) && fits_buffer_offsets(Hq, n, D, io->qr_stride_h, io->qr_stride_l, es) &&
          fits_buffer_offsets(Hkv, n, D, io->k_stride_h, io->k_stride_l, es) && fits_buffer_offsets(Hkv, n, D, io->v_stride_h, io->v_stride_l, es) &&
          fits_buffer_offsets(Hkv, n, D, tail_sh, D, es))) {
        set_error("rtk_pivotkv_append_rope: an operand spans 2 GiB or more (or has a negative stride): 32-bit row offsets do not reach");
        return RTK_EUNSUPPORTED;
    }
    RowSel rs;
    int rc = make_rowsel(rs, P, D, sections_host, nsec, "rtk_pivotkv_append_rope");
    if (rc) return rc;
    const int64_t* prev = ls->pos_len > 0 ? ls->pos + (ls->pos_len - 1) : nullptr;
    int64_t* pos_out = ls->pos + ls->pos_len;   // the shifted ids join the layer's position cache (reference :319-321)
    const PrologueGeom pg{Hq, Hkv, n, D, P, round_mode, inv_freq, attention_scaling};
    hipStream_t st = (hipStream_t)stream;
    // one token (a decode step): the kernel shifts the caller's id itself; longer segments take the shift launch after it
    int64_t* shift_back = (shift_ids_in_place && n == 1) ? (int64_t*)io->pos : nullptr;
    if (dtype == RTK_F16)
        rc = prologue_launch<RTK_F16>(pg, io, rs, prev, nullptr, nullptr, k_tail, v_tail, tail_sh, pos_out, ls->pos_cap, nullptr, st,
                                      shift_back);
    else if (dtype == RTK_BF16)
        rc = prologue_launch<RTK_BF16>(pg, io, rs, prev, nullptr, nullptr, k_tail, v_tail, tail_sh, pos_out, ls->pos_cap, nullptr, st,
                                       shift_back);
    else
        rc = prologue_launch<RTK_F32>(pg, io, rs, prev, nullptr, nullptr, k_tail, v_tail, tail_sh, pos_out, ls->pos_cap, nullptr, st,
                                      shift_back);
    if (rc) return rc;
    if (shift_ids_in_place && !shift_back) {   // qwen2_vl.py:73: later layers (and the caller) see the shifted ids; after the kernel read them
        rc = rtk_position_shift((int64_t*)io->pos, n, prev, stream);
        if (rc) return rc;
    }
    ls->length += n;
    ls->pos_len += n;
    return RTK_OK;
}

extern "C" int rtk_pivotkv_flush(rtk_pivotkv_batch* b, rtk_layer_state* const* layers, const int32_t* slots, int n,
                                 rtk_stream_t stream) {
    int rc = check_batch(b, "rtk_pivotkv_flush");
    if (rc) return rc;
    RTK_CHECK_ARG(layers && slots && n >= 1 && n <= b->slots, "rtk_pivotkv_flush: bad layer list");
    const int L = b->L, D = b->D, Hkv = b->Hkv, keep = b->keep, P = b->P;
    const size_t es = esize(b->dtype);
    const bool reforge = b->reforge != 0;
    if (reforge && (!b->inv_freq || P == 0 || !b->k_unrot || !b->pos_old || !b->pos_new)) {
        set_error("rtk_pivotkv_flush: pos_embed_reforge needs position ids and an inv_freq rotary (use the per-stage calls)");
        return RTK_EUNSUPPORTED;
    }
    if (!reforge && !b->keep_all && !b->k_stage && !b->compact_sync) {
        set_error("rtk_pivotkv_flush: no K staging buffer");
        return RTK_EINVAL;
    }
    for (int i = 0; i < n; ++i) {
        const rtk_layer_state* ls = layers[i];
        RTK_CHECK_ARG(ls && ls->k && ls->v, "rtk_pivotkv_flush: layer %d: NULL state", i);
        RTK_CHECK_ARG(slots[i] >= 0 && slots[i] < b->slots && (i == 0 || slots[i] > slots[i - 1]),
                      "rtk_pivotkv_flush: slots must be ascending and inside the batch");
        RTK_CHECK_ARG(ls->pending == L && ls->pending_keep == keep && ls->length + L <= ls->cap,
                      "rtk_pivotkv_flush: layer %d has no pending chunk of this batch", i);
        RTK_CHECK_ARG(!(reforge && P) || (ls->pos && ls->pos_len + keep <= ls->pos_cap),
                      "rtk_pivotkv_flush: layer %d: position cache has no room for %d ids", i, keep);
    }
    if (b->shift_row) {
        // the attention patch shifts the ids tensor it was handed in place (qwen2_vl.py:73); the prologue left that to
        // here: one launch per chunk, with the last layer's rule - what the reference's loop leaves behind
        const rtk_layer_state* last = layers[n - 1];
        const int64_t* prev = (last->pos && last->pos_len > 0) ? last->pos + (last->pos_len - 1) : nullptr;
        rc = rtk_position_shift(b->shift_row, L, prev, stream);
        if (rc) return rc;
        b->shift_row = nullptr;
    }
    if (!b->keep_all) {
        if (b->batched_passes) {
            std::vector<const void*> masks((size_t)n);
            int i = 0;
            auto in_place = [&](int slot) { return b->q_units && b->q_units[slot] != nullptr; };
            while (i < n) {   // every run of consecutive slots whose queries live alike in one launch per kernel (:260-268)
                int j = i;
                while (j + 1 < n && slots[j + 1] == slots[j] + 1 && in_place(slots[j + 1]) == in_place(slots[i])) ++j;
                const int l0 = slots[i], cnt = j - i + 1;
                bool any = false;
                for (int u = 0; u < cnt; ++u) {
                    masks[u] = b->skip_masked ? layers[i + u]->mask : nullptr;
                    any = any || masks[u];
                }
                // queries scored in place (prologue route): every unit of the run, or none
                const void* const* qu = in_place(l0) ? b->q_units + l0 : nullptr;
                rc = rtk_pivotkv_score_passes_batched_q(
                    (char*)b->score_ws + (size_t)l0 * b->score_ws_stride, b->score_ws_stride,
                    reforge ? (char*)b->k_unrot + (size_t)l0 * Hkv * L * D * es : nullptr, (size_t)Hkv * L * D * es,
                    b->partials + (size_t)l0 * b->partial_floats, b->partial_floats, cnt, b->Hq, Hkv, L, D, b->score_dtype,
                    (any && b->key_index) ? masks.data() : nullptr,
                    (any && b->key_index) ? b->key_index + (size_t)l0 * (L + 1) : nullptr, qu, b->q_stride_h, b->q_stride_l,
                    stream);
                if (rc) return rc;
                if (qu)
                    for (int u = 0; u < cnt; ++u) b->q_units[l0 + u] = nullptr;
                i = j + 1;
            }
        }
        std::vector<rtk_select_unit> su((size_t)n);
        for (int i = 0; i < n; ++i) {   // mask override + top-k + id gather / rescale (:269-295)
            const int l = slots[i];
            rtk_select_unit& u = su[i];
            u.partial = b->partials + (size_t)l * b->partial_floats;
            u.score = b->score + (size_t)l * L;
            u.mask = layers[i]->mask;
            u.pos = P ? b->pos_old + (size_t)l * P * L : nullptr;
            u.keep_idx = b->keep_idx + (size_t)l * keep;
            u.rank = nullptr;
            u.pos_out = P ? b->pos_new + (size_t)l * keep : nullptr;
            u.workspace = (char*)b->sel_ws + (size_t)l * b->sel_ws_stride;
        }
        rc = rtk_pivotkv_select_batched(su.data(), n, Hkv, b->rs_n, b->Hq / Hkv, L, keep, P, (int)reforge,
                                        (int64_t)b->slots * keep, b->score_dtype, stream);
        if (rc) return rc;
    }
    if (b->compact_sync && !b->keep_all) {
        // the eviction scan as one in-place launch: kept K re-rotated (or copied) from k~ to the tail, V (and an
        // un-reforged K) compacted inside the tail, ids to the position cache (:278-318)
        std::vector<rtk_compact_unit> cu((size_t)n);
        for (int i = 0; i < n; ++i) {
            const int l = slots[i];
            rtk_layer_state* ls = layers[i];
            const size_t tail = (size_t)ls->length * D * es;
            rtk_compact_unit& u = cu[i];
            u.k_src = reforge ? (char*)b->k_unrot + (size_t)l * Hkv * L * D * es : nullptr;
            u.k_src_stride_h = (int64_t)L * D;
            u.k_tail = (char*)ls->k + tail;
            u.k_tail_stride_h = ls->cap * D;
            u.v_tail = (char*)ls->v + tail;
            u.v_tail_stride_h = ls->cap * D;
            u.keep_idx = b->keep_idx + (size_t)l * keep;
            if (reforge && P) {
                u.pos_src = b->pos_new + (size_t)l * keep;
                u.pos_src_stride = (int64_t)b->slots * keep;
                u.pos_dst = ls->pos + ls->pos_len;
                u.pos_dst_stride = ls->pos_cap;
            } else {
                u.pos_src = u.pos_dst = nullptr;
                u.pos_src_stride = u.pos_dst_stride = 0;
            }
        }
        const int k_mode = !reforge ? RTK_COMPACT_K_INPLACE : (b->defer_rot ? RTK_COMPACT_K_COPY : RTK_COMPACT_K_ROTATE);
        rc = rtk_pivotkv_compact_batched(cu.data(), n, Hkv, D, keep, reforge ? P : 0, b->dtype, k_mode, b->inv_freq,
                                         b->attention_scaling, b->nsec ? b->sections : nullptr, b->nsec, b->round_mode,
                                         b->compact_sync, (size_t)b->compact_sync_ints, stream);
        if (rc) return rc;
        for (int i = 0; i < n; ++i) {
            rtk_layer_state* ls = layers[i];
            ls->length += keep;
            ls->pending = 0;
            ls->pending_keep = 0;
            ls->mask = nullptr;
            if (reforge && P) ls->pos_len += keep;
        }
        b->pre_rope = 0;
        return RTK_OK;
    }
    std::vector<rtk_evict_unit> eu((size_t)n);
    std::vector<rtk_place_unit> pl((size_t)2 * n);
    int nc = 0;
    for (int i = 0; i < n; ++i) {
        const int l = slots[i];
        rtk_layer_state* ls = layers[i];
        const size_t tail = (size_t)ls->length * D * es;
        rtk_evict_unit& u = eu[i];
        u.cos_new = u.sin_new = nullptr;
        const int64_t* kidx = b->keep_idx + (size_t)l * keep;
        if (reforge) {   // kept K = k~ re-rotated at the NEW ids, straight into the cache (:297-306)
            u.k_src = (char*)b->k_unrot + (size_t)l * Hkv * L * D * es;
            u.k_src_stride_h = (int64_t)L * D;
            // keep-all chunks of pre-RoPE units: new ids == old ids and k~ == k0, so the tail already holds the result
            // (not with a deferred re-rotation: the cache has to hold the UN-rotated rows)
            // (pre_rope == 2: reference operands - k~ carries the round trip's roundings, the kept row is ITS re-rotation)
            u.k_dst = (b->keep_all && b->pre_rope == 1 && !b->defer_rot) ? nullptr : (char*)ls->k + tail;
            u.k_dst_stride_h = ls->cap * D;
        } else {
            u.k_src = (char*)ls->k + tail;
            u.k_src_stride_h = ls->cap * D;
            if (b->keep_all) {
                u.k_dst = nullptr;
                u.k_dst_stride_h = 0;
            } else {
                u.k_dst = (char*)b->k_stage + (size_t)l * Hkv * keep * D * es;
                u.k_dst_stride_h = (int64_t)keep * D;
                pl[nc].stage = u.k_dst;
                pl[nc].stage_stride_h_bytes = (int64_t)keep * D * es;
                pl[nc].tail = (char*)ls->k + tail;
                pl[nc].tail_stride_h_bytes = ls->cap * D * es;
                pl[nc].keep_idx = kidx;
                ++nc;
            }
        }
        u.v_src = (char*)ls->v + tail;
        u.v_src_stride_h = ls->cap * D;
        if (b->keep_all) {   // every V row already sits where it belongs
            u.v_dst = nullptr;
            u.v_dst_stride_h = 0;
        } else {
            u.v_dst = (char*)b->v_stage + (size_t)l * Hkv * keep * D * es;
            u.v_dst_stride_h = (int64_t)keep * D;
            pl[nc].stage = u.v_dst;
            pl[nc].stage_stride_h_bytes = (int64_t)keep * D * es;
            pl[nc].tail = (char*)ls->v + tail;
            pl[nc].tail_stride_h_bytes = ls->cap * D * es;
            pl[nc].keep_idx = kidx;
            ++nc;
        }
        u.keep_idx = kidx;
        if (reforge && P) {   // bookkeeping (:308-309).  keep-all: the ids x 1.0 are the ids (:288-292)
            u.pos_src = b->keep_all ? b->pos_old + (size_t)l * P * L : b->pos_new + (size_t)l * keep;
            u.pos_src_stride = b->keep_all ? (int64_t)L : (int64_t)b->slots * keep;
            u.pos_dst = ls->pos + ls->pos_len;
            u.pos_dst_stride = ls->pos_cap;
        } else {
            u.pos_src = u.pos_dst = nullptr;
            u.pos_src_stride = u.pos_dst_stride = 0;
        }
    }
    if (reforge && b->defer_rot)   // un-rotated kept rows + their (provisional) ids; the owner rotates once, later
        rc = rtk_pivotkv_evict_batched(eu.data(), n, Hkv, D, keep, P, b->dtype, 3, stream);
    else if (reforge)
        rc = rtk_pivotkv_evict_batched_rope(eu.data(), n, Hkv, D, keep, P, b->dtype, b->inv_freq, b->attention_scaling,
                                            b->nsec ? b->sections : nullptr, b->nsec, b->round_mode, 1, stream);
    else if (!b->keep_all)
        rc = rtk_pivotkv_evict_batched(eu.data(), n, Hkv, D, keep, 0, b->dtype, 1, stream);
    if (rc) return rc;
    if (nc) {
        rc = rtk_pivotkv_place_batched(pl.data(), nc, Hkv, keep, D, b->dtype, stream);
        if (rc) return rc;
    }
    for (int i = 0; i < n; ++i) {
        rtk_layer_state* ls = layers[i];
        ls->length += keep;
        ls->pending = 0;
        ls->pending_keep = 0;
        ls->mask = nullptr;
        if (reforge && P) ls->pos_len += keep;
    }
    b->pre_rope = 0;
    return RTK_OK;
}

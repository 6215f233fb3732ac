// common.cuh — shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "retake_hip.h"

namespace rtk {

constexpr int WAVE = 64;

// ---- bf16 <-> f32 (round to nearest even, NaN quieted; identical to c10::BFloat16) ------------
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

// ---- fp16 <-> f32 (IEEE, round to nearest even: v_cvt_f16_f32 / v_cvt_f32_f16, like c10::Half) ---------
__device__ __forceinline__ float hf2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t f2hf(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ float rhf(float f) { return (float)(_Float16)f; }

// The two 16-bit float formats behind one interface (DT = RTK_BF16 or RTK_F16): torch rounds the result of every
// elementwise op on such tensors to the tensor's dtype, so the kernels that restate a chain of torch ops need "round
// two fp32 values into a packed pair", "read a half of a packed pair" and "round trip one value".
using rtk_f32x2 = __attribute__((ext_vector_type(2))) float;
using rtk_bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using rtk_f16x2 = __attribute__((ext_vector_type(2))) _Float16;
template <int DT> struct H16;
template <> struct H16<RTK_BF16> {
    static constexpr uint32_t ONE2 = 0x3f803f80u;   // (1.0, 1.0)
    __device__ static __forceinline__ uint32_t pack2(float lo, float hi) {   // v_cvt_pk_bf16_f32
        const rtk_f32x2 v = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, rtk_bf16x2));
    }
    __device__ static __forceinline__ float lo(uint32_t p) { return __uint_as_float(p << 16); }
    __device__ static __forceinline__ float hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }
    __device__ static __forceinline__ float rnd(float x) { return rbf(x); }
    __device__ static __forceinline__ float ld(const void* p, size_t i) { return bf2f(((const uint16_t*)p)[i]); }
    __device__ static __forceinline__ void st(void* p, size_t i, float x) { ((uint16_t*)p)[i] = f2bf(x); }
    // acc + a.lo*b.lo + a.hi*b.hi on packed pairs (products of two 16-bit floats are exact in fp32)
    __device__ static __forceinline__ float dot2(uint32_t a, uint32_t b, float acc) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(rtk_bf16x2, a), __builtin_bit_cast(rtk_bf16x2, b), acc, false);
    }
};
template <> struct H16<RTK_F16> {
    static constexpr uint32_t ONE2 = 0x3c003c00u;
    __device__ static __forceinline__ uint32_t pack2(float lo, float hi) {   // 2 x v_cvt_f16_f32 + pack, RNE, overflow -> inf
        const rtk_f32x2 v = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, rtk_f16x2));
    }
    __device__ static __forceinline__ float lo(uint32_t p) { return (float)__builtin_bit_cast(rtk_f16x2, p)[0]; }
    __device__ static __forceinline__ float hi(uint32_t p) { return (float)__builtin_bit_cast(rtk_f16x2, p)[1]; }
    __device__ static __forceinline__ float rnd(float x) { return rhf(x); }
    __device__ static __forceinline__ float ld(const void* p, size_t i) { return hf2f(((const uint16_t*)p)[i]); }
    __device__ static __forceinline__ void st(void* p, size_t i, float x) { ((uint16_t*)p)[i] = f2hf(x); }
    __device__ static __forceinline__ float dot2(uint32_t a, uint32_t b, float acc) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(rtk_f16x2, a), __builtin_bit_cast(rtk_f16x2, b), acc, false);
    }
};
// A wave-uniform int the compiler keeps in a scalar register (readfirstlane makes the uniformity provable): the head index
// of the per-update kernels, whose row offsets then travel in the soffset operand of buffer loads / stores (buf_load).
__device__ __forceinline__ int uniform_int(int x) { return __builtin_amdgcn_readfirstlane(x); }

// round x to the 16-bit format named by `mode` (0 = keep fp32, 1 = bf16, 2 = fp16): the dtype a rotary module casts its
// cos / sin tables to (`.to(x.dtype)`)
__device__ __forceinline__ float round_to(float x, int mode) { return mode == 1 ? rbf(x) : (mode == 2 ? rhf(x) : x); }

// ---- wave reductions (all 64 lanes participate) -------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// order-preserving map float -> uint32 (larger float <=> larger key); -0 < +0 is harmless here
__device__ __forceinline__ uint32_t f2key(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// sin / cos of an fp32 angle, CORRECTLY ROUNDED to fp32 (checked against float64 libm on 384 k angles id * inv_freq,
// ids up to 2e5: 0 mismatches; torch's CPU cos/sin, which the reference's rotary module calls, are within 0.6 ulp of
// the true value, so the two differ by at most one fp32 ulp, in ~5 % of the entries).  The angle is position *
// inv_freq rounded to fp32 exactly like the rotary module's fp32 product, so the reduction must start from THAT value:
// fp64 Cody-Waite reduction by pi/2 (exact for |x| < 2^24 pi/2: n has <= 24 bits, two fused multiply-adds), Taylor
// polynomials to r^15 / r^16 on |r| <= pi/4 in fp64 (truncation < 5e-17).  ~25 fp64 operations; ocml's sincosf is
// <= 2 ulp and several times longer (its large-argument path is a Payne-Hanek reduction).
__device__ __forceinline__ void sincos_cr(float x, float& sn, float& cs) {
    const double xd = (double)x;
    const double n = __builtin_rint(xd * 0.63661977236758134308);
    double r = __builtin_fma(n, -1.57079632679489655800e+00, xd);
    r = __builtin_fma(n, -6.12323399573676603587e-17, r);
    const double r2 = r * r;
    double ps = -7.6471637318198164759e-13;                      // -1/15!
    ps = __builtin_fma(ps, r2, 1.6059043836821614599e-10);       //  1/13!
    ps = __builtin_fma(ps, r2, -2.5052108385441718775e-08);      // -1/11!
    ps = __builtin_fma(ps, r2, 2.7557319223985890653e-06);       //  1/9!
    ps = __builtin_fma(ps, r2, -1.9841269841269841270e-04);      // -1/7!
    ps = __builtin_fma(ps, r2, 8.3333333333333333333e-03);       //  1/5!
    ps = __builtin_fma(ps, r2, -1.6666666666666666667e-01);      // -1/3!
    const double s = __builtin_fma(r * r2, ps, r);
    double pc = 4.7794773323873852974e-14;                       //  1/16!
    pc = __builtin_fma(pc, r2, -1.1470745597729724714e-11);      // -1/14!
    pc = __builtin_fma(pc, r2, 2.0876756987868098979e-09);       //  1/12!
    pc = __builtin_fma(pc, r2, -2.7557319223985890653e-07);      // -1/10!
    pc = __builtin_fma(pc, r2, 2.4801587301587301587e-05);       //  1/8!
    pc = __builtin_fma(pc, r2, -1.3888888888888888889e-03);      // -1/6!
    pc = __builtin_fma(pc, r2, 4.1666666666666666667e-02);       //  1/4!
    pc = __builtin_fma(pc, r2, -0.5);                            // -1/2!
    const double c = __builtin_fma(pc, r2, 1.0);
    const int q = (int)(long long)n & 3;                         // x = r + q pi/2 (mod 2 pi)
    const double so = (q & 1) ? c : s, co = (q & 1) ? s : c;
    sn = (float)((q & 2) ? -so : so);
    cs = (float)(((q + 1) & 2) ? -co : co);
}

// the reference's three bf16 roundings after the probabilities (longvideo_cache.py:268-270) for key j:
//   partial [Hq][RS][L] fp32 sums of bf16 probabilities  ->  score (a bf16 value held in fp32)
template <bool F16 = false>
__device__ __forceinline__ float finalize_ref_column(const float* __restrict__ partial, int Hkv, int RS, int G, int L, int j) {
    auto rnd16 = [](float x) { return F16 ? rhf(x) : rbf(x); };   // the tensor dtype of the reference's sums and means
    float tot = 0.f;
    for (int g = 0; g < Hkv; ++g) {
        float gs = 0.f;
        for (int hh = 0; hh < G; ++hh) {
            const float* p = partial + (size_t)(g * G + hh) * RS * L + j;
            float hs = 0.f;
            int r = 0;
            for (; r + 7 <= RS; r += 7) {          // seven independent loads in flight, summed in split order
                float v[7];
#pragma unroll
                for (int u = 0; u < 7; ++u) v[u] = p[(size_t)(r + u) * L];
#pragma unroll
                for (int u = 0; u < 7; ++u) hs += v[u];
            }
            for (; r < RS; ++r) hs += p[(size_t)r * L];
            gs += rnd16(hs);                        // .sum(1) -> bf16 (sums of <= 7 bf16 values are exact in fp32)
        }
        tot += rnd16(__fdiv_rn(gs, (float)G));       // .mean(1) -> bf16
    }
    return rnd16(__fdiv_rn(tot, (float)Hkv));        // .mean(0) -> bf16
}

// 16-byte vector types
// native 4 x u32 vector: loads/stores are first-class 16-byte operations (a struct would be copied with
// llvm.memcpy, which keeps register staging arrays in scratch memory)
using u32x4 = uint32_t __attribute__((ext_vector_type(4)));

// ---- 16-byte chunks of a row as fp32 lanes (un-rotate / rotate kernels) ----------------------------------------
template <int DT> struct Vec16;
template <> struct Vec16<RTK_F32> {
    static constexpr int VE = 4;
    __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    __device__ static __forceinline__ u32x4 pack(const float* f) {
        return u32x4{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
    }
};
template <> struct Vec16<RTK_BF16> {
    static constexpr int VE = 8;
    __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
        f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
        f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    __device__ static __forceinline__ u32x4 pack(const float* f) {
        return u32x4{(uint32_t)f2bf(f[0]) | ((uint32_t)f2bf(f[1]) << 16), (uint32_t)f2bf(f[2]) | ((uint32_t)f2bf(f[3]) << 16),
                     (uint32_t)f2bf(f[4]) | ((uint32_t)f2bf(f[5]) << 16), (uint32_t)f2bf(f[6]) | ((uint32_t)f2bf(f[7]) << 16)};
    }
};

template <> struct Vec16<RTK_F16> {
    static constexpr int VE = 8;
    __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
        f[0] = H16<RTK_F16>::lo(v.x); f[1] = H16<RTK_F16>::hi(v.x); f[2] = H16<RTK_F16>::lo(v.y); f[3] = H16<RTK_F16>::hi(v.y);
        f[4] = H16<RTK_F16>::lo(v.z); f[5] = H16<RTK_F16>::hi(v.z); f[6] = H16<RTK_F16>::lo(v.w); f[7] = H16<RTK_F16>::hi(v.w);
    }
    __device__ static __forceinline__ u32x4 pack(const float* f) {
        return u32x4{H16<RTK_F16>::pack2(f[0], f[1]), H16<RTK_F16>::pack2(f[2], f[3]), H16<RTK_F16>::pack2(f[4], f[5]),
                     H16<RTK_F16>::pack2(f[6], f[7])};
    }
};

// (x*cos) + (rotate_half(x)*sin) for one 16-byte chunk `lo` of the first half of a row and its partner `hi` in the second
// half - apply_rotary_pos_emb's op chain with one rounding per torch op and no fma contraction.  16-bit payloads round
// through the packed converts (H16<DT>::pack2: one instruction per pair); rotate_half(x)[d] = -x2,
// rotate_half(x)[d + D/2] = x1.  c1 / s1: tables of the channels of `lo`, c2 / s2: of `hi`.
template <int DT>
__device__ __forceinline__ void rotate_chunk_pair(const u32x4& lo, const u32x4& hi, const float* c1, const float* s1,
                                                  const float* c2, const float* s2, u32x4& olo, u32x4& ohi) {
    if constexpr (DT != RTK_F32) {
        using Hh = H16<DT>;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float x1a = Hh::lo(lo[w]), x1b = Hh::hi(lo[w]), x2a = Hh::lo(hi[w]), x2b = Hh::hi(hi[w]);
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
        for (int e = 0; e < 4; ++e) {
            const float x1 = __uint_as_float(lo[e]), x2 = __uint_as_float(hi[e]);
            olo[e] = __float_as_uint(__fadd_rn(__fmul_rn(x1, c1[e]), __fmul_rn(-x2, s1[e])));
            ohi[e] = __float_as_uint(__fadd_rn(__fmul_rn(x2, c2[e]), __fmul_rn(x1, s2[e])));
        }
    }
}

// NW 32-bit words of a row as one aligned load / store (the narrow-chunk forms of the per-update kernels)
template <int NW> struct alignas(4 * NW) WV { uint32_t w[NW]; };

// NW-dword accesses through a buffer descriptor: address = descriptor base + soff (SGPR: the head's offset) + voff (VGPR: the
// thread's offset inside a head, computed once) + a compile-time immediate - no vector ALU work per access (the per-update
// kernels touch 2 x 18 rows per thread and were spending a third of their vector instructions on 64-bit row addresses).
// Offsets are 32-bit: the launchers check that a tensor's extent fits (RTK_EUNSUPPORTED otherwise).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* p) {   // p must be wave-uniform (readfirstlane says so)
    const unsigned long long a = (unsigned long long)p;
    const unsigned long long u = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    return __builtin_amdgcn_make_buffer_rsrc((void*)u, 0, 0xfffffffc, 0x00020000);
}
template <int NW, int IMM = 0> __device__ __forceinline__ WV<NW> buf_load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    WV<NW> o;
    if constexpr (NW == 1) {
        o.w[0] = __builtin_amdgcn_raw_buffer_load_b32(r, voff + IMM, soff, 0);
    } else if constexpr (NW == 2) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, voff + IMM, soff, 0);
        o.w[0] = v[0]; o.w[1] = v[1];
    } else {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff + IMM, soff, 0);
        o.w[0] = v[0]; o.w[1] = v[1]; o.w[2] = v[2]; o.w[3] = v[3];
    }
    return o;
}
template <int NW, int IMM = 0> __device__ __forceinline__ void buf_store(const WV<NW>& x, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    if constexpr (NW == 1) {
        __builtin_amdgcn_raw_buffer_store_b32(x.w[0], r, voff + IMM, soff, 0);
    } else if constexpr (NW == 2) {
        using u2 = __attribute__((ext_vector_type(2))) unsigned int;
        __builtin_amdgcn_raw_buffer_store_b64(u2{x.w[0], x.w[1]}, r, voff + IMM, soff, 0);
    } else {
        using u4 = __attribute__((ext_vector_type(4))) unsigned int;
        __builtin_amdgcn_raw_buffer_store_b128(u4{x.w[0], x.w[1], x.w[2], x.w[3]}, r, voff + IMM, soff, 0);
    }
}

}  // namespace rtk

// ---- host-side error plumbing --------------------------------------------------------------------
namespace rtk {
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
}  // namespace rtk

// ---- M-RoPE channel -> position-row selection (longvideo_cache.py:68-74), shared by rope.hip and the fused prepare kernel
namespace rtk {
struct RowSel {
    uint8_t row[256];  // which of the P position rows (t/h/w) feeds channel d
};

// cos/sin of one token for channels d .. d+VE-1 (c1, s1) and d+h2 .. d+h2+VE-1 (c2, s2): rope_table_kernel's
// arithmetic (fp32 id * inv_freq[channel mod h2], correctly rounded sin / cos, * attention_scaling, bf16 rounding on request).  The
// token's P ids are passed in registers (pid[row]) and the row selectors come in two wide loads, so no load
// depends on another one; d must be a multiple of VE (4 or 8).
// one channel pair (dch, dch + h2) of one token: (c1, s1) for channel dch, (c2, s2) for its rotation partner
__device__ __forceinline__ void rope_elem(float f, int ra, int rb, const float (&pid)[3], float scaling, int round_bf16,
                                          float& c1, float& s1, float& c2, float& s2) {
    const float p1 = ra == 0 ? pid[0] : (ra == 1 ? pid[1] : pid[2]);
    float sn, cs;
    sincos_cr(p1 * f, sn, cs);
    cs *= scaling;
    sn *= scaling;
    cs = round_to(cs, round_bf16);   // round_bf16: 0 = fp32 tables, 1 = bf16, 2 = fp16 (the model dtype)
    sn = round_to(sn, round_bf16);
    c1 = cs;
    s1 = sn;
    if (rb != ra) {
        const float p2 = rb == 0 ? pid[0] : (rb == 1 ? pid[1] : pid[2]);
        sincos_cr(p2 * f, sn, cs);
        cs *= scaling;
        sn *= scaling;
        cs = round_to(cs, round_bf16);
        sn = round_to(sn, round_bf16);
    }
    c2 = cs;
    s2 = sn;
}

template <int VE>
__device__ __forceinline__ void rope_chunk(const float* __restrict__ inv_freq, const RowSel& rs, int d, int h2,
                                           const float (&pid)[3], float scaling, int round_bf16, float* c1, float* s1,
                                           float* c2, float* s2) {
    uint8_t ra[VE], rb[VE];
    if constexpr (VE == 8) {
        const uint64_t wa = *(const uint64_t*)(rs.row + d), wb = *(const uint64_t*)(rs.row + d + h2);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ra[e] = (uint8_t)(wa >> (8 * e));
            rb[e] = (uint8_t)(wb >> (8 * e));
        }
    } else if constexpr (VE == 2) {
        const uint16_t wa = *(const uint16_t*)(rs.row + d), wb = *(const uint16_t*)(rs.row + d + h2);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            ra[e] = (uint8_t)(wa >> (8 * e));
            rb[e] = (uint8_t)(wb >> (8 * e));
        }
    } else {
        const uint32_t wa = *(const uint32_t*)(rs.row + d), wb = *(const uint32_t*)(rs.row + d + h2);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            ra[e] = (uint8_t)(wa >> (8 * e));
            rb[e] = (uint8_t)(wb >> (8 * e));
        }
    }
    float f[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) f[e] = inv_freq[d + e];
#pragma unroll
    for (int e = 0; e < VE; ++e) rope_elem(f[e], ra[e], rb[e], pid, scaling, round_bf16, c1[e], s1[e], c2[e], s2[e]);
}

// bf16(x * (1/a2)) == bf16(x / a2) for every finite bf16 x?  (pivotkv_score.hip; exhaustive, cached per a2)
bool bf16_rcp_is_exact(float a2);
// rtk_pivotkv_prepare with the next layer's id shift in the launch (pivotkv_score.hip; used by rtk_pivotkv_update)
int pivotkv_prepare_shift(const void* q, int64_t q_stride_h, int64_t q_stride_l, const void* k, int64_t k_stride_h,
                          int64_t k_stride_l, const void* v, int64_t v_stride_h, int64_t v_stride_l, int Hq, int Hkv, int L,
                          int D, int dtype, const int64_t* pos, int64_t pos_stride, int P, const float* inv_freq,
                          float attention_scaling, const int* sections_host, int nsec, int round_bf16, void* k_unrot,
                          void* workspace, size_t workspace_bytes, void* k_tail, void* v_tail, int64_t tail_stride_h,
                          int64_t* pos_copy, int64_t* shift_row, const int64_t* next_prev, int32_t* ticket,
                          int64_t ticket_ints, int32_t* status, rtk_stream_t stream);

// The per-update kernels address rows as (descriptor base + 32-bit head offset + 32-bit thread offset): every byte of a
// [H, L, D] operand with element strides (sh, sl, 1) must lie below 2 GiB from its base, strides non-negative.
inline bool fits_buffer_offsets(int64_t H, int64_t L, int64_t D, int64_t sh, int64_t sl, size_t es) {
    if (sh < 0 || sl < 0) return false;
    const unsigned long long last = ((unsigned long long)(H > 0 ? H - 1 : 0) * (unsigned long long)sh +
                                     (unsigned long long)(L > 0 ? L - 1 : 0) * (unsigned long long)sl + (unsigned long long)D) * es;
    return last < (1ull << 31);
}

inline int make_rowsel(RowSel& rs, int P, int D, const int* sections, int nsec, const char* who) {
    if (D > 256 || (D & 1)) {
        set_error("%s: head_dim %d unsupported (must be even and <= 256)", who, D);
        return RTK_EUNSUPPORTED;
    }
    for (int d = 0; d < D; ++d) rs.row[d] = 0;
    if (P == 1) return RTK_OK;
    if (P != 3 || !sections || nsec < 1) {
        set_error("%s: P=%d needs mrope sections", who, P);
        return RTK_EINVAL;
    }
    int tot = 0;
    for (int i = 0; i < nsec; ++i) tot += sections[i];
    if (2 * tot != D) {
        set_error("%s: sum(mrope_section)*2 = %d != head_dim %d", who, 2 * tot, D);
        return RTK_EINVAL;
    }
    int d = 0;
    for (int rep = 0; rep < 2; ++rep)
        for (int i = 0; i < nsec; ++i)
            for (int c = 0; c < sections[i]; ++c, ++d) rs.row[d] = (uint8_t)((rep * nsec + i) % 3);
    return RTK_OK;
}

}  // namespace rtk

// ---- optional per-kernel HIP-event timing (rtk_profile_*), used by bench.py ------------------------
namespace rtk {
enum KernelId {
    KID_DIS = 0, KID_DPSEL, KID_GATHER, KID_ROPE, KID_UNROT, KID_PASS1, KID_PASS2, KID_FINALIZE, KID_PSEL, KID_EVICT,
    KID_COPY, KID_APPEND, KID_EVICTB, KID_COMMITB, KID_SHIFT, KID_PEMIT, KID_PROLOGUE, KID_COMPACT, KID_COUNT
};
bool profile_on(int kid);
void profile_begin(int kid, hipStream_t st);
void profile_end(int kid, hipStream_t st);
}  // namespace rtk

// launch `kern` on `st`; when profiling is enabled the launch is bracketed by two hipEvents on `st`
#define RTK_LAUNCH(kid, kern, grid, block, shmem, st, ...)                       \
    do {                                                                         \
        const bool prof__ = rtk::profile_on(kid);                                   \
        if (prof__) rtk::profile_begin(kid, st);                                 \
        hipLaunchKernelGGL(kern, grid, block, shmem, st, __VA_ARGS__);           \
        if (prof__) rtk::profile_end(kid, st);                                   \
    } while (0)

#define RTK_CHECK_ARG(cond, ...)                \
    do {                                        \
        if (!(cond)) {                          \
            rtk::set_error(__VA_ARGS__);        \
            return RTK_EINVAL;                  \
        }                                       \
    } while (0)

#define RTK_LAUNCH_CHECK(what)                                  \
    do {                                                        \
        hipError_t e__ = hipGetLastError();                     \
        if (e__ != hipSuccess) return rtk::hip_fail(e__, what); \
    } while (0)

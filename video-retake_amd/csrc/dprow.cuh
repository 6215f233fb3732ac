// dprow.cuh — one embedding row across a wave: the lane layout, packed-pair helpers and wave sums shared by the
// adjacent-cosine kernel (dpselect.hip) and the MA-LLM-hard chain (mallm_chain.hip).
#pragma once
#include "common.cuh"

namespace rtk {

template <int DT> struct Elem;
template <> struct Elem<RTK_F32> {
    static constexpr int PER_VEC = 4;
    using vec_t = float4;
    __device__ static void unpack(const vec_t& v, float* f) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
};
template <> struct Elem<RTK_BF16> {
    static constexpr int PER_VEC = 8;
    using vec_t = u32x4;
    __device__ static void unpack(const vec_t& v, float* f) {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
        f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
        f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
    }
};

template <> struct Elem<RTK_F16> {
    static constexpr int PER_VEC = 8;
    using vec_t = u32x4;
    __device__ static void unpack(const vec_t& v, float* f) {
        f[0] = H16<RTK_F16>::lo(v.x); f[1] = H16<RTK_F16>::hi(v.x); f[2] = H16<RTK_F16>::lo(v.y); f[3] = H16<RTK_F16>::hi(v.y);
        f[4] = H16<RTK_F16>::lo(v.z); f[5] = H16<RTK_F16>::hi(v.z); f[6] = H16<RTK_F16>::lo(v.w); f[7] = H16<RTK_F16>::hi(v.w);
    }
};

using bf16x2_dp = __attribute__((ext_vector_type(2))) __bf16;
using f32x2_dp = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ uint32_t pack2_bf16_dp(float lo, float hi) {   // v_cvt_pk_bf16_f32
    const f32x2_dp v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_dp));
}

// acc + a.lo*b.lo + a.hi*b.hi on packed bf16 pairs (v_dot2c_f32_bf16): products of bf16 are exact in fp32
__device__ __forceinline__ float dot2_bf16_dp(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_dp, a), __builtin_bit_cast(bf16x2_dp, b), acc, false);
}

// Sum over the 64 lanes, returned wave-uniform (read from lane 63).  Six DPP adds on the VALU instead of six
// ds_bpermute round trips through the LDS crossbar: xor-1 and xor-2 inside quads, the two mirrors for 8 and 16
// lanes, then row_bcast:15 / row_bcast:31 carry the row sums up to the last row.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    // lanes of rows outside ROW_MASK, and lanes whose DPP source is invalid, add 0 (old = 0, bound_ctrl off)
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_uniform(float v) {
    v = dpp_add<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);   // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);   // row_mirror        -> every lane holds its 16-lane row sum
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ---- one adjacent pair on one wave -------------------------------------------------------------------------------
// cos(a, b) = sum_c (a_c / max(||a||, eps)) * (b_c / max(||b||, eps)) in the tensor dtype (F.cosine_similarity,
// visual_compression.py:19 / :63 / :100), for two rows of nvec 16-byte vectors: EXACTLY dis_kernel's arithmetic - the
// same lane layout (lane owns vectors k*64 + lane), the same packed sums of squares and of rounded products, the same
// DPP reduction order, the same reciprocal-product normalisation - so a pair scored here carries the bits dis_kernel
// gives it inside a strip.  Returned wave-uniform.
template <int DT, int VPL>
__device__ __forceinline__ void dprow_load(typename Elem<DT>::vec_t* raw, const typename Elem<DT>::vec_t* p, int nvec, int lane) {
    constexpr int KFULL = (VPL <= 6) ? VPL - 1 : VPL - 2;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        const int v = k * WAVE + lane;
        raw[k] = p[k < KFULL ? v : min(v, nvec - 1)];
    }
}

template <int DT, int VPL>
__device__ __forceinline__ void dprow_normalise(typename Elem<DT>::vec_t* raw, float* cur, int nvec, int lane) {
    using E = Elem<DT>;
    using vec_t = typename E::vec_t;
    constexpr int PV = E::PER_VEC;
    constexpr int NE = VPL * PV;
    constexpr int KFULL = (VPL <= 6) ? VPL - 1 : VPL - 2;
#pragma unroll
    for (int k = KFULL; k < VPL; ++k)
        if (k * WAVE + lane >= nvec) raw[k] = vec_t{};
#pragma unroll
    for (int k = 0; k < VPL; ++k) E::unpack(raw[k], cur + k * PV);
    float ss = 0.f;
    if constexpr (DT != RTK_F32) {
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            ss = H16<DT>::dot2(raw[k].x, raw[k].x, ss);
            ss = H16<DT>::dot2(raw[k].y, raw[k].y, ss);
            ss = H16<DT>::dot2(raw[k].z, raw[k].z, ss);
            ss = H16<DT>::dot2(raw[k].w, raw[k].w, ss);
        }
    } else {
#pragma unroll
        for (int e = 0; e < NE; ++e) ss = fmaf(cur[e], cur[e], ss);
    }
    ss = wave_sum_uniform(ss);
    float nrm = sqrtf(ss);
    if constexpr (DT == RTK_F16) {
        nrm = fmaxf(rhf(nrm), rhf(1e-8f));
#pragma unroll
        for (int e = 0; e < NE; ++e) cur[e] = __fdiv_rn(cur[e], nrm);
#pragma unroll
        for (int e = 0; e < NE; e += 2) {
            const uint32_t pk = H16<DT>::pack2(cur[e], cur[e + 1]);
            cur[e] = H16<DT>::lo(pk);
            cur[e + 1] = H16<DT>::hi(pk);
        }
    } else if constexpr (DT == RTK_BF16) {
        nrm = rbf(nrm);
        nrm = fmaxf(nrm, rbf(1e-8f));
        if (__builtin_amdgcn_readfirstlane(__float_as_int(nrm)) > 0x7b800000 /* 2^120 */) {
#pragma unroll
            for (int e = 0; e < NE; ++e) cur[e] = __fdiv_rn(cur[e], nrm);
        } else {
            const float rn = __frcp_rn(nrm);
#pragma unroll
            for (int e = 0; e < NE; ++e) cur[e] *= rn;
        }
#pragma unroll
        for (int e = 0; e < NE; e += 2) {
            const uint32_t pk = pack2_bf16_dp(cur[e], cur[e + 1]);
            cur[e] = __uint_as_float(pk << 16);
            cur[e + 1] = __uint_as_float(pk & 0xffff0000u);
        }
    } else {
        nrm = fmaxf(nrm, 1e-8f);
#pragma unroll
        for (int e = 0; e < NE; ++e) cur[e] = cur[e] / nrm;
    }
}

template <int DT, int VPL>
__device__ __forceinline__ float dprow_pair_cos(const typename Elem<DT>::vec_t* a, const typename Elem<DT>::vec_t* b, int nvec,
                                                int lane) {
    using E = Elem<DT>;
    using vec_t = typename E::vec_t;
    constexpr int NE = VPL * E::PER_VEC;
    vec_t ra[VPL], rb[VPL];
    float na[NE], nb[NE];
    dprow_load<DT, VPL>(ra, a, nvec, lane);
    dprow_load<DT, VPL>(rb, b, nvec, lane);
    dprow_normalise<DT, VPL>(ra, na, nvec, lane);
    dprow_normalise<DT, VPL>(rb, nb, nvec, lane);
    float dot = 0.f;
    if constexpr (DT != RTK_F32) {
#pragma unroll
        for (int e = 0; e < NE; e += 2) {
            const uint32_t pk = H16<DT>::pack2(na[e] * nb[e], na[e + 1] * nb[e + 1]);
            dot = H16<DT>::dot2(pk, H16<DT>::ONE2, dot);
        }
    } else {
#pragma unroll
        for (int e = 0; e < NE; ++e) dot = fmaf(na[e], nb[e], dot);
    }
    dot = wave_sum_uniform(dot);
    if constexpr (DT != RTK_F32) dot = H16<DT>::rnd(dot);
    return dot;
}

}  // namespace rtk

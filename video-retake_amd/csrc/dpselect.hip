// dpselect.hip — DPSelect on gfx950: adjacent-frame cosine distance, peak stencil + exact top-k,
// frame gather.  Replaces retake/visual_compression.py:100-175.
//
// Roofline: HBM-bound.  The distance kernel reads every (frame, patch) embedding row exactly once
// (plus one halo row per strip) — algorithmic bytes T*N*C*sizeof(elem) + 4*T*N — with 16-byte
// coalesced loads, keeps the previous frame's normalised row in registers, and reduces with wave
// shuffles.  The gather moves 2*t*N*C*sizeof(elem).  MFMA is not used: ~1 flop per byte.
#include "common.cuh"
#include "dprow.cuh"
#include "select.cuh"

namespace rtk {

// ------------------------------------------------------------------------------------------------
// K1-K2: distance.  One wave walks `strip` consecutive frames of one patch position.
// ------------------------------------------------------------------------------------------------
// VPL = 16-byte vectors per lane; a row has nvec = C / PER_VEC vectors, lane owns vec k*64+lane.  One wave walks
// `strip` (<= 64) consecutive frames of one patch position; the strip's results stay in a register (lane i holds
// row i) and are stored once at the end, and every load in the loop is unconditional (clamped addresses) so that
// the two rows in flight are tracked with counted waits rather than a full drain.
template <int DT, int VPL>
__global__ __launch_bounds__(256) void dis_kernel(const typename Elem<DT>::vec_t* __restrict__ x, int T, int N,
                                                  int nvec, int strip, int emit_cos, float* __restrict__ dis) {
    using E = Elem<DT>;
    using vec_t = typename E::vec_t;
    constexpr int PV = E::PER_VEC;
    constexpr int NE = VPL * PV;
    constexpr int KFULL = (VPL <= 6) ? VPL - 1 : VPL - 2;   // vectors below KFULL are full for every lane
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int nstrips = (T + strip - 1) / strip;
    const int n = wave % N;
    const int s = wave / N;
    if (s >= nstrips) return;
    const int t0 = s * strip;
    const int t1 = min(T, t0 + strip);
    const size_t row_vecs = (size_t)nvec;
    const size_t frame_vecs = (size_t)N * row_vecs;
    const vec_t* xrow = x + (size_t)n * row_vecs;

    vec_t rawA[VPL], rawB[VPL];   // two rows in flight per wave
    float nA[NE], nB[NE];         // normalised rows: odd steps in nA, even steps in nB (no copies between steps)

    // rows beyond the strip are clamped to its last row (a harmless re-read of a line that is in flight anyway)
    auto load_row = [&](vec_t* raw, int t) {
        const vec_t* p = xrow + (size_t)min(t, t1 - 1) * frame_vecs;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int v = k * WAVE + lane;
            raw[k] = p[k < KFULL ? v : min(v, nvec - 1)];
        }
    };
    // normalise the row held in raw[] into cur[] (x / max(||x||, eps)), reference rounding for bf16
    auto normalise = [&](vec_t* raw, float* cur) {
#pragma unroll
        for (int k = KFULL; k < VPL; ++k)
            if (k * WAVE + lane >= nvec) raw[k] = vec_t{};   // lanes past the end of the row contribute zeros
#pragma unroll
        for (int k = 0; k < VPL; ++k) E::unpack(raw[k], cur + k * PV);
        float ss = 0.f;
        if constexpr (DT != RTK_F32) {   // sum of squares straight from the packed words, two elements per instruction
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
            // fp16 tensors: norm -> fp16, clamp_min(eps) with eps = fp16(1e-8) = 0 (a zero vector divides 0 by 0 like the
            // reference does), the quotient rounded to fp16.  True IEEE division: the reciprocal-product argument below
            // is a statement about 8-bit significands.
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
            // bf16(fl32(x / nrm)) without a division per element: x and nrm are bf16 values, and the exact quotient
            // of two 8-bit significands is never closer than 128 fp32 ulp to a bf16 rounding midpoint (unless it is
            // exactly representable), while x * fl32(1/nrm) is within 2 fp32 ulp of it - so the reciprocal product
            // rounds to the same bf16 as the reference's division, always (tools/bf16_quotient_check.py walks every
            // significand pair, subnormal results included).  Only a norm so large that 1/nrm is subnormal takes
            // the true division; the branch is wave-uniform.
            if (__builtin_amdgcn_readfirstlane(__float_as_int(nrm)) > 0x7b800000 /* 2^120 */) {
#pragma unroll
                for (int e = 0; e < NE; ++e) cur[e] = __fdiv_rn(cur[e], nrm);
            } else {
                const float rn = __frcp_rn(nrm);
#pragma unroll
                for (int e = 0; e < NE; ++e) cur[e] *= rn;
            }
#pragma unroll
            for (int e = 0; e < NE; e += 2) {   // hardware pack-convert = round to nearest even, like c10::BFloat16
                const uint32_t pk = pack2_bf16_dp(cur[e], cur[e + 1]);
                cur[e] = __uint_as_float(pk << 16);
                cur[e + 1] = __uint_as_float(pk & 0xffff0000u);
            }
        } else {
            nrm = fmaxf(nrm, 1e-8f);
#pragma unroll
            for (int e = 0; e < NE; ++e) cur[e] = cur[e] / nrm;
        }
    };
    const int tb = (t0 == 0) ? 1 : t0;   // first row this wave produces a distance for
    float res = 1.0f;                    // lane i: result of row tb + i (or of row 0 for the lane that gets it)
    // cos / distance of row t (in cur[]) against the previous row (prevn[])
    auto emit = [&](int t, const float* prevn, const float* cur) {
        float dot = 0.f;
        if constexpr (DT != RTK_F32) {
#pragma unroll
            for (int e = 0; e < NE; e += 2) {
                const uint32_t pk = H16<DT>::pack2(prevn[e] * cur[e], prevn[e + 1] * cur[e + 1]);
                dot = H16<DT>::dot2(pk, H16<DT>::ONE2, dot);   // + round(product) for both halves
            }
        } else {
#pragma unroll
            for (int e = 0; e < NE; ++e) dot = fmaf(prevn[e], cur[e], dot);
        }
        dot = wave_sum_uniform(dot);
        if constexpr (DT != RTK_F32) dot = H16<DT>::rnd(dot);
        if (lane == t - tb) res = emit_cos ? dot : 1.0f - dot;
    };

    load_row(rawA, t0 == 0 ? 0 : t0 - 1);   // first row, or the halo row of the previous strip
    load_row(rawB, tb);
    normalise(rawA, nA);
    load_row(rawA, tb + 1);
    // rows tb, tb+2, ... go through rawB -> nB, rows tb+1, tb+3, ... through rawA -> nA; a raw buffer is
    // re-issued (two rows ahead) as soon as its row has been unpacked
    int t = tb;
    for (; t + 1 < t1; t += 2) {   // whole pairs only: the loads outstanding at the loop head are always B then A
        normalise(rawB, nB);
        load_row(rawB, t + 2);
        emit(t, nA, nB);
        normalise(rawA, nA);
        load_row(rawA, t + 3);
        emit(t + 1, nB, nA);
    }
    if (t < t1) {
        normalise(rawB, nB);
        emit(t, nA, nB);
    }
    // one store per produced row: MA-LLM keeps the similarity itself in [T-1, N]; DPSelect keeps 1 - cos in
    // [T, N] with torch.ones_like(dis[:1]) for frame 0   (visual_compression.py:103-106)
    const int tr = tb + lane;
    if (tr < t1) dis[(size_t)(emit_cos ? tr - 1 : tr) * N + n] = res;
    if (t0 == 0 && lane == 0 && !emit_cos) dis[n] = 1.0f;
}

// Generic fallback: any C (scalar loads, two passes over each row through L2).  One wave per (t,n).
template <int DT>
__global__ __launch_bounds__(256) void dis_kernel_generic(const void* __restrict__ xv, int T, int N, int C,
                                                          int emit_cos, float* __restrict__ dis) {
    const int lane = threadIdx.x & (WAVE - 1);
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    if (wave >= (size_t)T * N) return;
    const int t = (int)(wave / N), n = (int)(wave % N);
    if (t == 0) {
        if (lane == 0 && !emit_cos) dis[n] = 1.0f;
        return;
    }
    auto ld = [&](size_t off) -> float {
        if constexpr (DT != RTK_F32) return H16<DT>::ld(xv, off);
        else return ((const float*)xv)[off];
    };
    const size_t a0 = ((size_t)(t - 1) * N + n) * C, b0 = ((size_t)t * N + n) * C;
    float sa = 0.f, sb = 0.f;
    for (int c = lane; c < C; c += WAVE) {
        const float a = ld(a0 + c), b = ld(b0 + c);
        sa = fmaf(a, a, sa);
        sb = fmaf(b, b, sb);
    }
    float na = sqrtf(wave_sum(sa)), nb = sqrtf(wave_sum(sb));
    float dot = 0.f;
    if constexpr (DT != RTK_F32) {
        using Hh = H16<DT>;
        na = fmaxf(Hh::rnd(na), Hh::rnd(1e-8f));
        nb = fmaxf(Hh::rnd(nb), Hh::rnd(1e-8f));
        for (int c = lane; c < C; c += WAVE) dot += Hh::rnd(Hh::rnd(ld(a0 + c) / na) * Hh::rnd(ld(b0 + c) / nb));
        dot = Hh::rnd(wave_sum(dot));
    } else {
        na = fmaxf(na, 1e-8f);
        nb = fmaxf(nb, 1e-8f);
        for (int c = lane; c < C; c += WAVE) dot = fmaf(ld(a0 + c) / na, ld(b0 + c) / nb, dot);
        dot = wave_sum(dot);
    }
    if (lane == 0) {
        if (emit_cos) dis[(size_t)(t - 1) * N + n] = dot;
        else dis[(size_t)t * N + n] = 1.0f - dot;
    }
}

// ------------------------------------------------------------------------------------------------
// K3-K7, K9: per-row peak stencil + bonus + exact top-k + ordered emit.  One workgroup per row
// (sync: the single patch-mean row; async: one row per patch position).
// ------------------------------------------------------------------------------------------------
// i is a peak iff arg-max over [i - w/2, i - w/2 + w - 1] ∩ [0,T) is i with first-index-wins ties
// (max_pool1d_with_indices + `window_maxima[c] == c`, visual_compression.py:121-123,153-156).
template <typename DFn>
__device__ __forceinline__ bool is_peak(DFn d, int i, int T, int window) {
    const float di = d(i);
    int a = i - window / 2, b = a + window - 1;
    a = max(a, 0);
    b = min(b, T - 1);
    bool pk = true;
    for (int j = a; j <= b; ++j) {
        if (j == i) continue;
        const float dj = d(j);
        pk = pk && (j < i ? (di > dj) : (di >= dj));
    }
    return pk;
}

constexpr int SEL_BLOCK = 256;

__global__ __launch_bounds__(SEL_BLOCK) void dpselect_select_kernel(const float* __restrict__ dis, int T, int N,
                                                                    int tgt, int window, int sync,
                                                                    int64_t* __restrict__ idx,
                                                                    uint8_t* __restrict__ mask,
                                                                    float* __restrict__ keys) {
    __shared__ SelectSmem sm;
    const int tid = threadIdx.x;
    const int row = blockIdx.x;  // patch index in async mode, 0 in sync mode
    float* krow = keys + (size_t)row * T;
    float* raw = keys + (size_t)T;  // sync only: patch-mean distance
    if (sync) {
        // dis.mean(1)  (:110): one wave per frame row, fixed-order lane partials + shuffle tree
        const int lane = tid & (WAVE - 1), wid = tid / WAVE;
        for (int t = wid; t < T; t += SEL_BLOCK / WAVE) {
            float s = 0.f;
            for (int n = lane; n < N; n += WAVE) s += dis[(size_t)t * N + n];
            s = wave_sum(s);
            if (lane == 0) raw[t] = s / (float)N;
        }
        __syncthreads();
    }
    auto dfn = [&](int t) -> float { return sync ? raw[t] : dis[(size_t)t * N + row]; };
    for (int t = tid; t < T; t += SEL_BLOCK) {
        const float d = dfn(t);
        krow[t] = is_peak(dfn, t, T, window) ? d + 2.0f : d;  // dis[peaks] += 2  (:133, :160)
    }
    __syncthreads();
    auto key = [&](int t) -> uint32_t { return f2key(krow[t]); };
    uint32_t thr;
    int need_eq;
    block_radix_threshold<SEL_BLOCK>(key, T, tgt, sm, thr, need_eq);
    block_ordered_compact<SEL_BLOCK>(key, T, thr, need_eq, sm, [&](int r, int t) {
        const uint8_t pk = is_peak(dfn, t, T, window) ? 1 : 0;
        if (sync) {
            idx[r] = t;
            for (int n = 0; n < N; ++n) mask[(size_t)r * N + n] = pk;  // mask[:,None].repeat(1,N) (:140)
        } else {
            idx[(size_t)r * N + row] = t;   // peaks.transpose(0,1)   (:169)
            mask[(size_t)r * N + row] = pk; // mask.gather(0, peaks)  (:175)
        }
    });
}

// ------------------------------------------------------------------------------------------------
// K8: frame gather.  One wave per output row (t', n); 16-byte vectors, loads batched before stores.
// ------------------------------------------------------------------------------------------------
template <int UNROLL>
__global__ __launch_bounds__(256) void gather_rows16_kernel(const u32x4* __restrict__ x, int T, int N, int rowvec,
                                                            const int64_t* __restrict__ idx, int tgt, int sync,
                                                            u32x4* __restrict__ out) {
    const int lane = threadIdx.x & (WAVE - 1);
    const size_t nw = ((size_t)gridDim.x * blockDim.x) / WAVE;
    const size_t rows = (size_t)tgt * N;
    for (size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE; r < rows; r += nw) {
        const int tt = (int)(r / N), n = (int)(r % N);
        const int64_t f = sync ? idx[tt] : idx[r];
        const u32x4* src = x + ((size_t)f * N + n) * rowvec;
        u32x4* dst = out + r * rowvec;
        for (int v0 = 0; v0 < rowvec; v0 += UNROLL * WAVE) {
            u32x4 buf[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int v = v0 + u * WAVE + lane;
                if (v < rowvec) buf[u] = src[v];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int v = v0 + u * WAVE + lane;
                if (v < rowvec) dst[v] = buf[u];
            }
        }
    }
}

__global__ __launch_bounds__(256) void gather_rows_bytes_kernel(const uint8_t* __restrict__ x, int T, int N,
                                                                size_t rowbytes, const int64_t* __restrict__ idx,
                                                                int tgt, int sync, uint8_t* __restrict__ out) {
    const int lane = threadIdx.x & (WAVE - 1);
    const size_t nw = ((size_t)gridDim.x * blockDim.x) / WAVE;
    const size_t rows = (size_t)tgt * N;
    for (size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE; r < rows; r += nw) {
        const int tt = (int)(r / N), n = (int)(r % N);
        const int64_t f = sync ? idx[tt] : idx[r];
        const uint8_t* src = x + ((size_t)f * N + n) * rowbytes;
        uint8_t* dst = out + r * rowbytes;
        for (size_t b = lane; b < rowbytes; b += WAVE) dst[b] = src[b];
    }
}

// ------------------------------------------------------------------------------------------------
// MA-LLM / MA-LLM-hard (visual_compression.py:5-83): one merge step = adjacent cosine (dis_kernel in
// cosine mode) -> arg-max frame pair per patch position -> merge + shift into a [T-1,N,C] bank.
// ------------------------------------------------------------------------------------------------
// idx[n] = first arg-max over t of sim[t,n]; sync: of the patch-mean row, the same for every n (:20-24)
__global__ __launch_bounds__(256) void mallm_argmax_kernel(const float* __restrict__ cosv, int T1, int N, int sync,
                                                           int round_bf16, int64_t* __restrict__ idx) {
    extern __shared__ float mrow[];  // sync: [T1] patch means
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
    const int n = blockIdx.x;
    if (sync) {
        for (int t = wid; t < T1; t += 4) {  // similarity_matrix.mean(-1): fp32 accumulate, one rounding
            float s = 0.f;
            for (int j = lane; j < N; j += WAVE) s += cosv[(size_t)t * N + j];
            s = wave_sum(s);
            if (lane == 0) {
                float m = s / (float)N;
                mrow[t] = round_to(m, round_bf16);   // 0 = fp32, 1 = bf16, 2 = fp16 (the bank's dtype)
            }
        }
        __syncthreads();
    }
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int t = tid; t < T1; t += 256) {
        const float v = sync ? mrow[t] : cosv[(size_t)t * N + n];
        if (v > best || (v == best && t < bidx) || bidx == 0x7fffffff) {
            best = v;
            bidx = t;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(best, o, WAVE);
        const int i2 = __shfl_xor(bidx, o, WAVE);
        if (i2 != 0x7fffffff && (bidx == 0x7fffffff || v2 > best || (v2 == best && i2 < bidx))) {
            best = v2;
            bidx = i2;
        }
    }
    if (lane == 0) { bv[wid] = best; bi[wid] = bidx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (bi[w] != 0x7fffffff && (bi[0] == 0x7fffffff || bv[w] > bv[0] || (bv[w] == bv[0] && bi[w] < bi[0]))) {
                bv[0] = bv[w];
                bi[0] = bi[w];
            }
        if (sync) for (int j = 0; j < N; ++j) idx[j] = bi[0];
        else idx[n] = bi[0];
    }
}

// out[t,n,:] for the T-1 output frames; one thread per element (the op is a streaming copy with at most two
// source rows per output row).  Soft merge keeps the reference's op order and roundings (:35-45):
//   (x[d]*s[d]) / s[d]   and at the merged slot   ((x[i]*s[i]) + (x[i+1]*s[i+1])) / (s[i] + s[i+1])
template <int DT>
__global__ __launch_bounds__(256) void mallm_merge_kernel(const void* __restrict__ xv, const void* __restrict__ sv,
                                                          const int64_t* __restrict__ idx, int T, int N, int C, int hard,
                                                          void* __restrict__ outv, void* __restrict__ sov) {
    const size_t total = (size_t)(T - 1) * N * C;
    auto ldx = [&](size_t i) -> float {
        if constexpr (DT != RTK_F32) return H16<DT>::ld(xv, i);
        else return ((const float*)xv)[i];
    };
    auto lds = [&](size_t i) -> float {
        if constexpr (DT != RTK_F32) return H16<DT>::ld(sv, i);
        else return ((const float*)sv)[i];
    };
    auto rnd = [&](float v) -> float {
        if constexpr (DT != RTK_F32) return H16<DT>::rnd(v);
        else return v;
    };
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const size_t tn = e / C;
        const int n = (int)(tn % N), t = (int)(tn / N);
        const int ix = (int)idx[n];
        float o;
        if (hard) {
            const int d = t + (t >= ix);
            if (DT != RTK_F32) ((uint16_t*)outv)[e] = ((const uint16_t*)xv)[((size_t)d * N + n) * C + c];
            else ((float*)outv)[e] = ((const float*)xv)[((size_t)d * N + n) * C + c];
            continue;
        }
        const int d = t + (t > ix);
        const float sd = lds((size_t)d * N + n);
        const float xd = ldx(((size_t)d * N + n) * C + c);
        float den = sd;
        if (t != ix) {
            o = rnd(__fdiv_rn(rnd(__fmul_rn(xd, sd)), sd));
        } else {
            const float ss = lds((size_t)(ix + 1) * N + n);
            const float xs = ldx(((size_t)(ix + 1) * N + n) * C + c);
            den = rnd(__fadd_rn(sd, ss));
            o = rnd(__fdiv_rn(rnd(__fadd_rn(rnd(__fmul_rn(xd, sd)), rnd(__fmul_rn(xs, ss)))), den));
        }
        if constexpr (DT != RTK_F32) H16<DT>::st(outv, e, o);
        else ((float*)outv)[e] = o;
        if (c == 0) {
            if constexpr (DT != RTK_F32) H16<DT>::st(sov, tn, den);
            else ((float*)sov)[tn] = den;
        }
    }
}

}  // namespace rtk

using namespace rtk;

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
template <int DT>
static int launch_dis(const void* x, int T, int N, int C, int emit_cos, float* dis, hipStream_t st) {
    constexpr int PV = Elem<DT>::PER_VEC;
    using vec_t = typename Elem<DT>::vec_t;
    const bool vec_ok = (C % PV == 0) && (((uintptr_t)x & 15) == 0);
    const int nvec = C / PV;
    const int vpl = (nvec + WAVE - 1) / WAVE;
    if (!vec_ok || vpl > 16) {
        const size_t waves = (size_t)T * N;
        const unsigned grid = (unsigned)((waves * WAVE + 255) / 256);
        RTK_LAUNCH(KID_DIS, dis_kernel_generic<DT>, dim3(grid), dim3(256), 0, st, x, T, N, C, emit_cos, dis);
        RTK_LAUNCH_CHECK("dis_kernel_generic");
        return RTK_OK;
    }
    // Strip length: the whole job should be ONE resident set of waves.  Every wave does the same work per row,
    // so a second, partly filled round of workgroups costs as much as a full one (measured: 12.5k waves on 8k
    // slots ran at 57 % VALU utilisation); with strips sized so that N * nstrips just fits the resident slots the
    // halo re-read is 1/strip and no SIMD idles while others finish.  Small jobs fall back to strips of >= 4 rows.
    const vec_t* xv = (const vec_t*)x;
    size_t slots = 8192;
#define RTK_DIS_CASE(V)                                                                                  \
    {                                                                                                    \
        static int resident = 0; /* waves of this instantiation the chip holds at once */               \
        if (!resident) {                                                                                 \
            int dev = 0, cus = 256, nb = 0;                                                              \
            (void)hipGetDevice(&dev);                                                                    \
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);               \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, dis_kernel<DT, V>, 256, 0) != hipSuccess || nb < 1) \
                nb = 1;                                                                                  \
            resident = nb * cus * 4;                                                                     \
        }                                                                                                \
        slots = (size_t)resident;                                                                        \
        int nstr = (int)std::max<size_t>(1, std::min<size_t>(slots / (size_t)N, (size_t)(T + 3) / 4));      \
        nstr = std::max(nstr, (T + 63) / 64); /* a strip's results live in one register: <= 64 rows */     \
        const int strip = (T + nstr - 1) / nstr;                                                         \
        const size_t waves = (size_t)N * ((T + strip - 1) / strip);                                      \
        const unsigned grid = (unsigned)((waves * WAVE + 255) / 256);                                    \
        RTK_LAUNCH(KID_DIS, (dis_kernel<DT, V>), dim3(grid), dim3(256), 0, st, xv, T, N, nvec, strip, emit_cos, dis); \
    }                                                                                                    \
    break;
    switch (vpl) {
        case 1: RTK_DIS_CASE(1)
        case 2: RTK_DIS_CASE(2)
        case 3: RTK_DIS_CASE(3)
        case 4: RTK_DIS_CASE(4)
        case 5: RTK_DIS_CASE(5)
        case 6: RTK_DIS_CASE(6)
        case 7: case 8: RTK_DIS_CASE(8)
        case 9: case 10: RTK_DIS_CASE(10)
        case 11: case 12: RTK_DIS_CASE(12)
        case 13: case 14: RTK_DIS_CASE(14)
        default: RTK_DIS_CASE(16)
    }
#undef RTK_DIS_CASE
    RTK_LAUNCH_CHECK("dis_kernel");
    return RTK_OK;
}

extern "C" int rtk_dpselect_dis(const void* x, int T, int N, int C, int dtype, float* dis, rtk_stream_t stream) {
    RTK_CHECK_ARG(x && dis, "rtk_dpselect_dis: NULL pointer");
    RTK_CHECK_ARG(T >= 1 && N >= 1 && C >= 1, "rtk_dpselect_dis: bad shape T=%d N=%d C=%d", T, N, C);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RTK_F32) return launch_dis<RTK_F32>(x, T, N, C, 0, dis, st);
    if (dtype == RTK_BF16) return launch_dis<RTK_BF16>(x, T, N, C, 0, dis, st);
    if (dtype == RTK_F16) return launch_dis<RTK_F16>(x, T, N, C, 0, dis, st);
    set_error("rtk_dpselect_dis: unsupported dtype %d", dtype);
    return RTK_EINVAL;
}

extern "C" int rtk_adjacent_cosine(const void* x, int T, int N, int C, int dtype, float* cos_out, rtk_stream_t stream) {
    RTK_CHECK_ARG(x && cos_out, "rtk_adjacent_cosine: NULL pointer");
    RTK_CHECK_ARG(T >= 2 && N >= 1 && C >= 1, "rtk_adjacent_cosine: bad shape T=%d N=%d C=%d", T, N, C);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RTK_F32) return launch_dis<RTK_F32>(x, T, N, C, 1, cos_out, st);
    if (dtype == RTK_BF16) return launch_dis<RTK_BF16>(x, T, N, C, 1, cos_out, st);
    if (dtype == RTK_F16) return launch_dis<RTK_F16>(x, T, N, C, 1, cos_out, st);
    set_error("rtk_adjacent_cosine: unsupported dtype %d", dtype);
    return RTK_EINVAL;
}

extern "C" int rtk_mallm_argmax(const float* cosv, int T1, int N, int sync, int round_bf16, int64_t* idx,
                                rtk_stream_t stream) {
    RTK_CHECK_ARG(cosv && idx, "rtk_mallm_argmax: NULL pointer");
    RTK_CHECK_ARG(T1 >= 1 && N >= 1, "rtk_mallm_argmax: bad shape T1=%d N=%d", T1, N);
    const size_t lds = sync ? (size_t)T1 * sizeof(float) : 0;
    if (lds > 64 * 1024) {
        set_error("rtk_mallm_argmax: %d frames exceed the sync path's LDS row", T1);
        return RTK_EUNSUPPORTED;
    }
    RTK_LAUNCH(KID_DPSEL, mallm_argmax_kernel, dim3(sync ? 1 : N), dim3(256), lds, (hipStream_t)stream, cosv, T1, N, sync,
               round_bf16, idx);
    RTK_LAUNCH_CHECK("mallm_argmax_kernel");
    return RTK_OK;
}

extern "C" int rtk_mallm_merge(const void* x, const void* sizes, const int64_t* idx, int T, int N, int C, int dtype,
                               int hard, void* out, void* sizes_out, rtk_stream_t stream) {
    RTK_CHECK_ARG(x && idx && out, "rtk_mallm_merge: NULL pointer");
    RTK_CHECK_ARG(hard || (sizes && sizes_out), "rtk_mallm_merge: the soft merge needs sizes and sizes_out");
    RTK_CHECK_ARG(T >= 2 && N >= 1 && C >= 1, "rtk_mallm_merge: bad shape T=%d N=%d C=%d", T, N, C);
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_F16, "rtk_mallm_merge: unsupported dtype %d", dtype);
    const size_t total = (size_t)(T - 1) * N * C;
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 65535u * 16u);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RTK_BF16)
        RTK_LAUNCH(KID_GATHER, mallm_merge_kernel<RTK_BF16>, dim3(grid), dim3(256), 0, st, x, sizes, idx, T, N, C, hard, out,
                   sizes_out);
    else if (dtype == RTK_F16)
        RTK_LAUNCH(KID_GATHER, mallm_merge_kernel<RTK_F16>, dim3(grid), dim3(256), 0, st, x, sizes, idx, T, N, C, hard, out,
                   sizes_out);
    else
        RTK_LAUNCH(KID_GATHER, mallm_merge_kernel<RTK_F32>, dim3(grid), dim3(256), 0, st, x, sizes, idx, T, N, C, hard, out,
                   sizes_out);
    RTK_LAUNCH_CHECK("mallm_merge_kernel");
    return RTK_OK;
}

extern "C" int rtk_dpselect_select(const float* dis, int T, int N, int tgt, int window, int sync, int64_t* idx,
                                   uint8_t* mask, float* keys, rtk_stream_t stream) {
    RTK_CHECK_ARG(dis && idx && mask && keys, "rtk_dpselect_select: NULL pointer");
    RTK_CHECK_ARG(T >= 1 && N >= 1 && window >= 1, "rtk_dpselect_select: bad shape T=%d N=%d window=%d", T, N, window);
    RTK_CHECK_ARG(tgt >= 1 && tgt <= T, "rtk_dpselect_select: tgt_mem_len %d out of range [1,%d]", tgt, T);
    if (!sync && N == 1) {
        set_error("DPSelect async mode with N == 1: the reference raises IndexError (visual_compression.py:153-156)");
        return RTK_EREFCRASH;
    }
    RTK_LAUNCH(KID_DPSEL, dpselect_select_kernel, dim3(sync ? 1 : N), dim3(SEL_BLOCK), 0, (hipStream_t)stream, dis, T, N,
                       tgt, window, sync, idx, mask, keys);
    RTK_LAUNCH_CHECK("dpselect_select_kernel");
    return RTK_OK;
}

extern "C" int rtk_gather_frames(const void* x, int T, int N, int C, int dtype, const int64_t* idx, int tgt, int sync,
                                 void* out, rtk_stream_t stream) {
    RTK_CHECK_ARG(x && idx && out, "rtk_gather_frames: NULL pointer");
    RTK_CHECK_ARG(T >= 1 && N >= 1 && C >= 1 && tgt >= 1, "rtk_gather_frames: bad shape");
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_F16, "rtk_gather_frames: unsupported dtype %d", dtype);
    const size_t esize = dtype == RTK_F32 ? 4 : 2;
    const size_t rowbytes = (size_t)C * esize;
    const size_t rows = (size_t)tgt * N;
    const unsigned grid = (unsigned)std::min<size_t>((rows + 3) / 4, 8192);
    hipStream_t st = (hipStream_t)stream;
    if (rowbytes % 16 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
        RTK_LAUNCH(KID_GATHER, gather_rows16_kernel<4>, dim3(grid), dim3(256), 0, st, (const u32x4*)x, T, N,
                           (int)(rowbytes / 16), idx, tgt, sync, (u32x4*)out);
    } else {
        RTK_LAUNCH(KID_GATHER, gather_rows_bytes_kernel, dim3(grid), dim3(256), 0, st, (const uint8_t*)x, T, N, rowbytes,
                           idx, tgt, sync, (uint8_t*)out);
    }
    RTK_LAUNCH_CHECK("gather_rows_kernel");
    return RTK_OK;
}

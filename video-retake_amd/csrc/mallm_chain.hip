// mallm_chain.hip — MA-LLM-hard run to its target length in ONE pass over the bank.
//
// The reference shrinks the bank one frame per call (visual_compression.py:50-83, looped at qwen2_vl.py:406-408):
// every call recomputes ALL adjacent cosines of the current bank and rebuilds the whole [T-1, N, C] bank - an
// O((T - t) * T * N * C) job that re-reads ~1 GB per step at T = 2048.  But a hard merge only DROPS the first frame of
// the most similar pair (out[t] = x[t + (t >= idx)], :69-82): the surviving rows are never modified, so of the next
// call's cosines all but one are the ones just computed - only the pair that closes over the dropped frame is new.
//   async (per patch position, the shipped patch_sync: False): one wave per patch keeps its T - 1 cosines and the
//     frame list (next / previous links) in LDS; a step is an arg-max (first index wins, torch.max :66), an unlink and
//     ONE new pair cosine (two rows of the bank, dprow_pair_cos = dis_kernel's arithmetic bit for bit);
//   sync (one frame list for all patches): the similarity is the patch mean (:64-65), so a step scores the new pair at
//     all N patch positions (one workgroup, 16 waves) and takes their mean in rtk_mallm_argmax's summation order.
// Output: the surviving frame indices ([t, N] / [t]); rtk_gather_frames copies the rows.  Same values as looping
// memory_bank_compress_MALLM_hard, bit for bit (tests/test_hip_parity.py::test_mallm_hard_chain_*).
#include "dprow.cuh"

namespace rtk {

constexpr int CHAIN_NONE = 0x7fffffff;

// lane-local best over slots lane, lane + 64, ...: ascending scan with a strict '>' keeps the first maximum
__device__ __forceinline__ void chain_rescan(const float* cs, int n_slots, int lane, float& bv, int& bi) {
    bv = -INFINITY;
    bi = CHAIN_NONE;
    for (int i = lane; i < n_slots; i += WAVE) {
        const float v = cs[i];
        if (v > bv) {
            bv = v;
            bi = i;
        }
    }
}

template <int DT, int VPL>
__global__ __launch_bounds__(WAVE) void mallm_hard_chain_kernel(const typename Elem<DT>::vec_t* __restrict__ x,
                                                                const float* __restrict__ cos0, int T, int N, int nvec,
                                                                int tgt, int64_t* __restrict__ idx_out) {
    using vec_t = typename Elem<DT>::vec_t;
    extern __shared__ __attribute__((aligned(16))) char chain_sm[];
    float* cs = (float*)chain_sm;   // [T]   cs[i] = cos(frame i, its next surviving frame); -inf: no such pair
    int* nx = (int*)(cs + T);       // [T]   next surviving frame (T: none)
    int* pv = nx + T;               // [T]   previous surviving frame (-1: none, -2: frame dropped)
    const int lane = threadIdx.x;
    const int n = blockIdx.x;
    for (int i = lane; i < T; i += WAVE) {
        cs[i] = (i < T - 1) ? cos0[(size_t)i * N + n] : -INFINITY;
        nx[i] = i + 1;
        pv[i] = i - 1;
    }
    __syncthreads();
    float bv;
    int bi;
    chain_rescan(cs, T - 1, lane, bv, bi);
    const size_t frame_vecs = (size_t)N * nvec;
    const vec_t* xn = x + (size_t)n * nvec;
    int head = 0;
    for (int step = T; step > tgt; --step) {
        float v = bv;
        int j = bi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(v, o, WAVE);
            const int j2 = __shfl_xor(j, o, WAVE);
            if (v2 > v || (v2 == v && j2 < j)) {
                v = v2;
                j = j2;
            }
        }
        if (j == CHAIN_NONE) j = head;   // nothing compares greater than -inf (NaN similarities): the first pair
        j = __builtin_amdgcn_readfirstlane(j);
        const int p = __builtin_amdgcn_readfirstlane(pv[j]), q = __builtin_amdgcn_readfirstlane(nx[j]);  // frame j leaves; q exists
        float newc = 0.f;
        if (p >= 0) newc = dprow_pair_cos<DT, VPL>(xn + (size_t)p * frame_vecs, xn + (size_t)q * frame_vecs, nvec, lane);
        if (lane == 0) {
            cs[j] = -INFINITY;
            pv[j] = -2;
            pv[q] = p;
            if (p >= 0) {
                nx[p] = q;
                cs[p] = newc;
            }
        }
        if (p < 0) head = q;
        __syncthreads();
        if (lane == (j & (WAVE - 1)) || (p >= 0 && lane == (p & (WAVE - 1)))) chain_rescan(cs, T - 1, lane, bv, bi);
    }
    // ordered emit of the survivors: contiguous frame blocks per lane, wave prefix sum of the counts
    const int per = (T + WAVE - 1) / WAVE;
    const int b = lane * per, e = min(T, b + per);
    int cnt = 0;
    for (int i = b; i < e; ++i) cnt += pv[i] != -2;
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const int t2 = __shfl_up(inc, o, WAVE);
        if (lane >= o) inc += t2;
    }
    int at = inc - cnt;
    for (int i = b; i < e; ++i)
        if (pv[i] != -2) idx_out[(size_t)(at++) * N + n] = i;
}

constexpr int SYNC_BLOCK = 1024;
template <int DT, int VPL>
__global__ __launch_bounds__(SYNC_BLOCK) void mallm_hard_chain_sync_kernel(const typename Elem<DT>::vec_t* __restrict__ x,
                                                                           const float* __restrict__ cos0, int T, int N,
                                                                           int nvec, int tgt, int round_mode,
                                                                           int64_t* __restrict__ idx_out) {
    extern __shared__ __attribute__((aligned(16))) char chain_sm[];
    float* ms = (float*)chain_sm;   // [T]  patch-mean similarity of (frame i, next surviving frame)
    int* nx = (int*)(ms + T);
    int* pv = nx + T;
    float* cn = (float*)(pv + T);   // [N]  the new pair's cosine at every patch position
    __shared__ float wbv[SYNC_BLOCK / WAVE];
    __shared__ int wbi[SYNC_BLOCK / WAVE];
    __shared__ int sel;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
    constexpr int NW = SYNC_BLOCK / WAVE;
    // similarity_matrix.mean(-1): fp32 accumulation in rtk_mallm_argmax's order, one rounding to the bank's dtype
    for (int t = wid; t < T - 1; t += NW) {
        float s = 0.f;
        for (int j = lane; j < N; j += WAVE) s += cos0[(size_t)t * N + j];
        s = wave_sum(s);
        if (lane == 0) ms[t] = round_to(s / (float)N, round_mode);
    }
    for (int i = tid; i < T; i += SYNC_BLOCK) {
        if (i == T - 1) ms[i] = -INFINITY;
        nx[i] = i + 1;
        pv[i] = i - 1;
    }
    if (tid == 0) sel = 0;
    __syncthreads();
    const size_t frame_vecs = (size_t)N * nvec;
    int head = 0;
    for (int step = T; step > tgt; --step) {
        float bv = -INFINITY;
        int bi = CHAIN_NONE;
        for (int i = tid; i < T - 1; i += SYNC_BLOCK) {
            const float v = ms[i];
            if (v > bv) {
                bv = v;
                bi = i;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(bv, o, WAVE);
            const int j2 = __shfl_xor(bi, o, WAVE);
            if (v2 > bv || (v2 == bv && j2 < bi)) {
                bv = v2;
                bi = j2;
            }
        }
        if (lane == 0) {
            wbv[wid] = bv;
            wbi[wid] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            float v = wbv[0];
            int j = wbi[0];
            for (int w = 1; w < NW; ++w)
                if (wbv[w] > v || (wbv[w] == v && wbi[w] < j)) {
                    v = wbv[w];
                    j = wbi[w];
                }
            sel = (j == CHAIN_NONE) ? head : j;
        }
        __syncthreads();
        const int j = sel;
        const int p = pv[j], q = nx[j];
        if (p >= 0) {
            for (int n = wid; n < N; n += NW) {
                const float c = dprow_pair_cos<DT, VPL>(x + (size_t)p * frame_vecs + (size_t)n * nvec,
                                                        x + (size_t)q * frame_vecs + (size_t)n * nvec, nvec, lane);
                if (lane == 0) cn[n] = c;
            }
        } else {
            head = q;
        }
        __syncthreads();
        if (wid == 0) {
            float m = 0.f;
            if (p >= 0) {
                float s = 0.f;
                for (int k = lane; k < N; k += WAVE) s += cn[k];
                s = wave_sum(s);
                m = round_to(s / (float)N, round_mode);
            }
            if (lane == 0) {
                ms[j] = -INFINITY;
                pv[j] = -2;
                pv[q] = p;
                if (p >= 0) {
                    nx[p] = q;
                    ms[p] = m;
                }
            }
        }
        __syncthreads();
    }
    if (wid == 0) {   // ordered emit (one wave)
        const int per = (T + WAVE - 1) / WAVE;
        const int b = lane * per, e = min(T, b + per);
        int cnt = 0;
        for (int i = b; i < e; ++i) cnt += pv[i] != -2;
        int inc = cnt;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const int t2 = __shfl_up(inc, o, WAVE);
            if (lane >= o) inc += t2;
        }
        int at = inc - cnt;
        for (int i = b; i < e; ++i)
            if (pv[i] != -2) idx_out[at++] = i;
    }
}

template <int DT>
static int chain_launch(const void* x, const float* cos0, int T, int N, int C, int tgt, int sync, int round_mode,
                        int64_t* idx_out, hipStream_t st) {
    using vec_t = typename Elem<DT>::vec_t;
    constexpr int PV = Elem<DT>::PER_VEC;
    const bool vec_ok = (C % PV == 0) && ((uintptr_t)x % 16 == 0);
    const int nvec = C / PV;
    const int vpl = (nvec + WAVE - 1) / WAVE;
    const size_t lds = (size_t)T * 12 + (sync ? (size_t)N * 4 : 0);
    if (!vec_ok || vpl > 16 || lds > 150 * 1024) {
        set_error("rtk_mallm_hard_chain: needs 16-byte rows of at most %d channels and T <= ~12000 (loop the single step)",
                  16 * WAVE * PV);
        return RTK_EUNSUPPORTED;
    }
    const vec_t* xv = (const vec_t*)x;
#define RTK_CHAIN_CASE(V)                                                                                                \
    {                                                                                                                    \
        const void* fn = sync ? (const void*)mallm_hard_chain_sync_kernel<DT, V> : (const void*)mallm_hard_chain_kernel<DT, V>; \
        if (lds > 48 * 1024) { /* long frame lists: opt in to the large dynamic LDS window */                           \
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
            if (e != hipSuccess) return hip_fail(e, "rtk_mallm_hard_chain: hipFuncSetAttribute");                          \
        }                                                                                                                \
        if (sync)                                                                                                        \
            RTK_LAUNCH(KID_DPSEL, (mallm_hard_chain_sync_kernel<DT, V>), dim3(1), dim3(SYNC_BLOCK), lds, st, xv, cos0, T, N, \
                       nvec, tgt, round_mode, idx_out);                                                                  \
        else                                                                                                             \
            RTK_LAUNCH(KID_DPSEL, (mallm_hard_chain_kernel<DT, V>), dim3(N), dim3(WAVE), lds, st, xv, cos0, T, N, nvec,    \
                       tgt, idx_out);                                                                                    \
    }                                                                                                                    \
    break;
    switch (vpl) {
        case 1: RTK_CHAIN_CASE(1)
        case 2: RTK_CHAIN_CASE(2)
        case 3: RTK_CHAIN_CASE(3)
        case 4: RTK_CHAIN_CASE(4)
        case 5: RTK_CHAIN_CASE(5)
        case 6: RTK_CHAIN_CASE(6)
        case 7: case 8: RTK_CHAIN_CASE(8)
        case 9: case 10: RTK_CHAIN_CASE(10)
        case 11: case 12: RTK_CHAIN_CASE(12)
        case 13: case 14: RTK_CHAIN_CASE(14)
        default: RTK_CHAIN_CASE(16)
    }
#undef RTK_CHAIN_CASE
    RTK_LAUNCH_CHECK("mallm_hard_chain_kernel");
    return RTK_OK;
}

}  // namespace rtk

using namespace rtk;

extern "C" int rtk_mallm_hard_chain(const void* x, int T, int N, int C, int dtype, int tgt, int sync, float* cos_ws,
                                    int64_t* idx_out, rtk_stream_t stream) {
    RTK_CHECK_ARG(x && cos_ws && idx_out, "rtk_mallm_hard_chain: NULL pointer");
    RTK_CHECK_ARG(T >= 2 && N >= 1 && C >= 1, "rtk_mallm_hard_chain: bad shape T=%d N=%d C=%d", T, N, C);
    RTK_CHECK_ARG(tgt >= 1 && tgt <= T, "rtk_mallm_hard_chain: target length %d outside [1, %d]", tgt, T);
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_F16, "rtk_mallm_hard_chain: unsupported dtype %d", dtype);
    int rc = rtk_adjacent_cosine(x, T, N, C, dtype, cos_ws, stream);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int mode = dtype == RTK_BF16 ? 1 : (dtype == RTK_F16 ? 2 : 0);
    if (dtype == RTK_BF16) return chain_launch<RTK_BF16>(x, cos_ws, T, N, C, tgt, sync, mode, idx_out, st);
    if (dtype == RTK_F16) return chain_launch<RTK_F16>(x, cos_ws, T, N, C, tgt, sync, mode, idx_out, st);
    return chain_launch<RTK_F32>(x, cos_ws, T, N, C, tgt, sync, mode, idx_out, st);
}

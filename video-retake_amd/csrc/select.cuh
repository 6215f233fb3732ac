// select.cuh — workgroup-cooperative exact top-k selection with ordered compaction.
//
// Replaces `topk(k, sorted=False)[1].sort()[0]` (visual_compression.py:134-135,167-168 and
// longvideo_cache.py:276-277).  One workgroup owns one row.  Selection is exact on the 32-bit
// order-preserving key of the fp32 value (4 radix passes of 8 bits, LDS histogram with integer
// atomics => deterministic), ties at the k-th boundary are resolved lowest index first, and the
// winners are emitted in ascending index order by a ballot/prefix compaction — no sort needed.
#pragma once
#include "common.cuh"

namespace rtk {

struct SelectSmem {
    uint32_t hist[256];
    uint32_t wave_cnt[2][16];  // per-wave counts of the two compaction scans (<= 16 waves)
    uint32_t bcast[4];         // prefix, remaining k
};

// Finds thr = key of the k-th largest element and need_eq = how many elements equal to thr
// belong to the top-k.  key(i) must be pure (it is re-evaluated on every pass).
template <int BLOCK, typename KeyFn>
__device__ __forceinline__ void block_radix_threshold(KeyFn key, int n, int k, SelectSmem& sm,
                                                      uint32_t& thr, int& need_eq) {
    const int tid = threadIdx.x;
    uint32_t prefix = 0, mask = 0;
    int kk = k;
#pragma unroll 1
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = tid; b < 256; b += BLOCK) sm.hist[b] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += BLOCK) {
            const uint32_t ki = key(i);
            if ((ki & mask) == prefix) atomicAdd(&sm.hist[(ki >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < WAVE) {
            // lane l owns bins 4l..4l+3; walk from the top bin (255) downwards
            const int lane = tid;
            uint32_t h0 = sm.hist[4 * lane], h1 = sm.hist[4 * lane + 1], h2 = sm.hist[4 * lane + 2],
                     h3 = sm.hist[4 * lane + 3];
            const uint32_t mine = h0 + h1 + h2 + h3;
            // suffix sum over lanes: above = sum of bins in lanes > lane
            uint32_t incl = mine;
#pragma unroll
            for (int o = 1; o < WAVE; o <<= 1) {
                const uint32_t t = __shfl_down(incl, o, WAVE);
                if (lane + o < WAVE) incl += t;
            }
            const uint32_t above = incl - mine;
            if (above < (uint32_t)kk && (uint32_t)kk <= incl) {
                // the k-th largest falls in this lane's four bins
                uint32_t c = above;
                int b;
                if ((uint32_t)kk <= c + h3) { b = 3; }
                else { c += h3; if ((uint32_t)kk <= c + h2) { b = 2; }
                else { c += h2; if ((uint32_t)kk <= c + h1) { b = 1; }
                else { c += h1; b = 0; } } }
                sm.bcast[0] = prefix | ((uint32_t)(4 * lane + b) << shift);
                sm.bcast[1] = (uint32_t)kk - c;
            }
        }
        __syncthreads();
        prefix = sm.bcast[0];
        kk = (int)sm.bcast[1];
        mask |= 255u << shift;
    }
    thr = prefix;
    need_eq = kk;
}

// Emits the winners in ascending index order: emit(out_rank, i).  All threads must call it.
template <int BLOCK, typename KeyFn, typename EmitFn>
__device__ __forceinline__ void block_ordered_compact(KeyFn key, int n, uint32_t thr, int need_eq,
                                                      SelectSmem& sm, EmitFn emit) {
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1), wid = tid / WAVE;
    constexpr int NW = BLOCK / WAVE;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (WAVE - lane));
    int eq_base = 0, out_base = 0;
#pragma unroll 1
    for (int i0 = 0; i0 < n; i0 += BLOCK) {
        const int i = i0 + tid;
        const bool valid = i < n;
        const uint32_t ki = valid ? key(i) : 0u;
        const bool gt = valid && ki > thr;
        const bool eq = valid && ki == thr;
        const unsigned long long eqm = __ballot(eq);
        if (lane == 0) sm.wave_cnt[0][wid] = (uint32_t)__popcll(eqm);
        __syncthreads();
        int eq_before = eq_base, eq_total = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int c = (int)sm.wave_cnt[0][w];
            if (w < wid) eq_before += c;
            eq_total += c;
        }
        const int eq_rank = eq_before + __popcll(eqm & lt);
        const bool sel = gt || (eq && eq_rank < need_eq);
        const unsigned long long sm_ = __ballot(sel);
        if (lane == 0) sm.wave_cnt[1][wid] = (uint32_t)__popcll(sm_);
        __syncthreads();
        int before = out_base, total = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int c = (int)sm.wave_cnt[1][w];
            if (w < wid) before += c;
            total += c;
        }
        if (sel) emit(before + __popcll(sm_ & lt), i);
        eq_base += eq_total;
        out_base += total;
        __syncthreads();  // wave_cnt reused next iteration
    }
}

}  // namespace rtk

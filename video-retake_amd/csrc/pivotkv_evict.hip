// pivotkv_evict.hip — PivotKV selection and eviction scan on gfx950.
// Replaces longvideo_cache.py:272-318 (masked_fill_, topk+sort, the three gathers, the temporal-id
// rescale, the forward re-rotation and the two torch.cat cache rebuilds).
//
// Roofline: HBM-bound byte shuffling.  One launch with two roles: the append blocks copy every K and
// V row of the chunk to the pre-allocated cache tail with 16-byte coalesced accesses, the kept blocks
// gather the kept 1/ratio of the rows (through keep_idx) into the compacted staging rows —
// algorithmic bytes per (layer, chunk): 2*Hkv*L*D*s read + 2*Hkv*keep*D*s written (+ the tail
// append 2*Hkv*L*D*s, which the reference pays as an O(cache) torch.cat).
#include <atomic>

#include "common.cuh"
#include "select.cuh"

namespace rtk {

constexpr int PSEL_BLOCK = 1024;

__global__ __launch_bounds__(PSEL_BLOCK) void pivotkv_select_kernel(float* __restrict__ score,
                                                                    const uint8_t* __restrict__ mask, int L, int keep,
                                                                    const int64_t* __restrict__ pos, int P,
                                                                    int reforge, int64_t* __restrict__ keep_idx,
                                                                    int32_t* __restrict__ rank,
                                                                    int64_t* __restrict__ pos_out, int64_t pos_ld) {
    __shared__ SelectSmem sm;
    __shared__ long long red[PSEL_BLOCK / WAVE];
    const int tid = threadIdx.x;
    for (int i = tid; i < L; i += PSEL_BLOCK) {
        if (mask && mask[i]) score[i] = 1.0f;  // attn_weights.masked_fill_(mask, 1.)  (:274)
        rank[i] = -1;
    }
    __syncthreads();
    auto key = [&](int i) -> uint32_t { return f2key(score[i]); };
    uint32_t thr;
    int need_eq;
    block_radix_threshold<PSEL_BLOCK>(key, L, keep, sm, thr, need_eq);
    block_ordered_compact<PSEL_BLOCK>(key, L, thr, need_eq, sm, [&](int r, int i) {
        keep_idx[r] = i;  // topk(keep).sort()  (:276-277)
        rank[i] = r;
        if (pos)
            for (int p = 0; p < P; ++p) pos_out[(size_t)p * pos_ld + r] = pos[(size_t)p * L + i];  // :283-288
    });
    if (!(pos && reforge)) return;
    __syncthreads();
    // min_temp_id = compressed_position_ids[0].min()  (:293)
    long long mn = 0x7fffffffffffffffLL;
    for (int r = tid; r < keep; r += PSEL_BLOCK) mn = min(mn, (long long)pos_out[r]);
    for (int o = 32; o > 0; o >>= 1) {
        const long long t = __shfl_xor(mn, o, WAVE);
        mn = min(mn, t);
    }
    if ((tid & (WAVE - 1)) == 0) red[tid / WAVE] = mn;
    __syncthreads();
    mn = red[0];
    for (int w = 1; w < PSEL_BLOCK / WAVE; ++w) mn = min(mn, red[w]);
    // comp_ratio = keep_len / k_len (python float) ; int64 * float -> float32 multiply ; .long() truncates (:294-295)
    const float ratio = (float)((double)keep / (double)L);
    for (int r = tid; r < keep; r += PSEL_BLOCK) {
        const float f = (float)((long long)pos_out[r] - mn) * ratio;
        pos_out[r] = mn + (long long)f;
    }
}

constexpr uint32_t KEY_ONE = 0xBF800000u;   // f2key(1.0f)

// ------------------------------------------------------------------------------------------------
// One workgroup per unit, small code: the selection of a batched launch runs on ONE CU per unit, where what counts is
// the number of instructions issued AND fetched - a kernel is entered with a cold instruction cache, and the
// register-resident form of this kernel (keys of 8 consecutive tokens per thread, every loop unrolled: 14 KB of
// straight-line code executed once) took 31 us per 28-unit launch at L = 6272 against 24 us for this one, 4.5 KB
// (same-box A/B, profiles/r13_ab_select.txt).  The keys live in LDS
// (thread t owns tokens [t*per, (t+1)*per), row stride per|1 words: conflict-free), every per-token loop is a real
// loop, and all global traffic is coalesced:
//   0  scores (+ mask override, :272-274) -> keys; the tokens at exactly 1.0 (every key-patch token) are counted once
//   1  4 radix passes of 8 bits over the LDS keys: a wave whose digits all agree issues one atomic
//   2  one packed block scan (greater | equal << 16): output position of every thread's first kept token
//   3  kept indices -> LDS in ascending order
//   4  cooperative over the kept rows: temporal ids gathered, min_temp_id (:293)
//   5  cooperative, coalesced stores: keep_idx, gathered / rescaled ids (:283-295), rank
// Exact radix select, ties lowest index first: the results of the chip-wide rank / emit pair and of the generic kernel above.
// ------------------------------------------------------------------------------------------------
constexpr int SEL_LDS_MAX_PER = 16;   // tokens per thread: L <= 16384
__host__ __device__ inline size_t select_lds_bytes(int L, int keep) {
    const int per = (L + PSEL_BLOCK - 1) / PSEL_BLOCK;
    return ((size_t)(per | 1) * PSEL_BLOCK + (size_t)keep) * sizeof(uint32_t);
}

__device__ __forceinline__ void select_lds_body(float* __restrict__ score, const uint8_t* __restrict__ mask, int L, int keep,
                                                const int64_t* __restrict__ pos, int P, int reforge,
                                                int64_t* __restrict__ keep_idx, int32_t* __restrict__ rank,
                                                int64_t* __restrict__ pos_out, int64_t pos_ld) {
    extern __shared__ uint32_t sel_lds[];
    __shared__ SelectSmem sm;
    __shared__ uint32_t wtot[PSEL_BLOCK / WAVE];
    __shared__ long long red[PSEL_BLOCK / WAVE];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
    const int per = (L + PSEL_BLOCK - 1) / PSEL_BLOCK, ps = per | 1;
    uint32_t* keyL = sel_lds;                         // token i at (i / per) * ps + i % per
    uint32_t* keepL = sel_lds + (size_t)ps * PSEL_BLOCK;
    const int base = tid * per, kb = tid * ps;
    const int mine = max(0, min(per, L - base));      // tokens this thread owns
    // ---- 0: keys --------------------------------------------------------------------------------------------
    uint32_t ones = 0;
    // i / per as a multiply-high: exact for i < 2^14 and 2 <= per <= 16 (the error term i * (M - 2^20/per) / 2^20 < 1/per)
    const uint32_t magic = (((1u << 20) + (uint32_t)per - 1u) / (uint32_t)per) << 12;
    for (int i = tid; i < L; i += PSEL_BLOCK) {
        float sc = score[i];
        if (mask && mask[i]) {  // attn_weights.masked_fill_(mask, 1.)  (:274)
            sc = 1.0f;
            score[i] = 1.0f;
        }
        const uint32_t k = f2key(sc);
        const int t = per == 1 ? i : (int)__umulhi((uint32_t)i, magic);
        keyL[t * ps + (i - t * per)] = k;
        ones += (uint32_t)__popcll(__ballot(k == KEY_ONE));
        if (rank) rank[i] = -1;
    }
    if (lane == 0) wtot[wid] = ones;
    // ---- 1: exact k-th largest key --------------------------------------------------------------------------
    uint32_t prefix = 0, pmask = 0;
    int kk = keep;
#pragma unroll 1
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) sm.hist[tid] = 0;
        __syncthreads();
        if (tid == PSEL_BLOCK - 1 && (KEY_ONE & pmask) == prefix) {   // the tokens at 1.0, as one count
            uint32_t n1 = 0;
            for (int w = 0; w < PSEL_BLOCK / WAVE; ++w) n1 += wtot[w];
            if (n1) atomicAdd(&sm.hist[(KEY_ONE >> shift) & 255u], n1);
        }
#pragma unroll 1
        for (int e = 0; e < per; ++e) {
            const uint32_t k = keyL[kb + e];
            const bool m = e < mine && k != KEY_ONE && (k & pmask) == prefix;
            const uint32_t d = (k >> shift) & 255u;
            const unsigned long long bal = __ballot(m);
            if (bal != 0) {                                           // wave-uniform
                const int leader = __ffsll((long long)bal) - 1;
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, leader);
                if (__ballot(m && d == d0) == bal) {                  // every digit of the wave alike: one atomic
                    if (lane == leader) atomicAdd(&sm.hist[d0], (uint32_t)__popcll(bal));
                } else if (m) {
                    atomicAdd(&sm.hist[d], 1u);
                }
            }
        }
        __syncthreads();
        if (tid < WAVE) {
            const uint32_t h0 = sm.hist[4 * lane], h1 = sm.hist[4 * lane + 1], h2 = sm.hist[4 * lane + 2],
                           h3 = sm.hist[4 * lane + 3];
            const uint32_t own = h0 + h1 + h2 + h3;
            uint32_t incl = own;
#pragma unroll
            for (int o = 1; o < WAVE; o <<= 1) {
                const uint32_t t = __shfl_down(incl, o, WAVE);
                if (lane + o < WAVE) incl += t;
            }
            const uint32_t above = incl - own;
            if (above < (uint32_t)kk && (uint32_t)kk <= incl) {
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
        pmask |= 255u << shift;
    }
    const uint32_t thr = prefix;
    const int need_eq = kk;
    // ---- 2: where this thread's kept tokens go ----------------------------------------------------------------
    uint32_t own = 0;
#pragma unroll 1
    for (int e = 0; e < mine; ++e) {
        const uint32_t k = keyL[kb + e];
        own += (k > thr ? 1u : 0u) + (k == thr ? 0x10000u : 0u);
    }
    uint32_t inc = own;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, WAVE);
        if (lane >= o) inc += t;
    }
    if (lane == WAVE - 1) wtot[wid] = inc;
    __syncthreads();
    uint32_t before = inc - own;
    for (int w = 0; w < wid; ++w) before += wtot[w];
    const int eq_before = (int)(before >> 16);
    int eq_left = max(0, need_eq - eq_before);                   // ties: lowest index first
    int r = (int)(before & 0xffffu) + min(eq_before, need_eq);
    // ---- 3: kept indices, ascending ----------------------------------------------------------------------------
#pragma unroll 1
    for (int e = 0; e < mine; ++e) {
        const uint32_t k = keyL[kb + e];
        bool sel = k > thr;
        if (k == thr && eq_left > 0) {
            sel = true;
            --eq_left;
        }
        if (sel) keepL[r++] = (uint32_t)(base + e);
    }
    __syncthreads();
    // ---- 4: min_temp_id = compressed_position_ids[0].min()  (:293) ---------------------------------------------
    const bool rf = pos && reforge;
    long long mn = 0x7fffffffffffffffLL;
    if (rf) {
        for (int q = tid; q < keep; q += PSEL_BLOCK) mn = min(mn, (long long)pos[keepL[q]]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const long long t = __shfl_xor(mn, o, WAVE);
            mn = min(mn, t);
        }
        if (lane == 0) red[wid] = mn;
        __syncthreads();
        mn = red[0];
        for (int w = 1; w < PSEL_BLOCK / WAVE; ++w) mn = min(mn, red[w]);
    }
    // ---- 5: outputs ----------------------------------------------------------------------------------------------
    const float ratio = (float)((double)keep / (double)L);  // comp_ratio = keep_len / k_len (:294)
    for (int q = tid; q < keep; q += PSEL_BLOCK) {
        const int i = (int)keepL[q];
        keep_idx[q] = i;  // topk(keep).sort()  (:276-277)
        if (pos) {
            const long long t0 = pos[i];
            // row 0: gathered id, rescaled when reforging: int64 -> float32 multiply -> truncation (:293-295)
            pos_out[q] = rf ? mn + (long long)((float)(t0 - mn) * ratio) : t0;
            for (int p = 1; p < P; ++p) pos_out[(size_t)p * pos_ld + q] = pos[(size_t)p * L + i];   // :283-288
        }
        if (rank) rank[i] = q;
    }
}

// ------------------------------------------------------------------------------------------------
// Chip-wide selection (the default path): the same exact result as the one-workgroup kernels above, in two
// launches that use every CU instead of one.
//   rank kernel  every workgroup loads ALL L keys into LDS and ranks RANK_TOK tokens by counting:
//                rank_i = #{j : key_j > key_i  or  (key_j == key_i and j < i)}   (ties: lowest index first)
//                token i is kept  <=>  rank_i < keep.                 O(L^2) compares over ~L/32 workgroups
//   emit kernel  ascending compaction: position of a kept token = number of kept tokens before it
//                (= topk(keep).sort()), id gather and temporal rescale
// Integer compares only: bit-reproducible and independent of the launch geometry.
// ------------------------------------------------------------------------------------------------
constexpr int RANK_TOK = 64;           // tokens ranked per workgroup (one per lane of a wave)
constexpr int RANK_SEG = 16;           // waves per workgroup, each scanning one segment of the keys
constexpr int RANK_BLOCK = RANK_TOK * RANK_SEG;

struct SelUnits {
    rtk_select_unit u[RTK_SELECT_MAX_UNITS];
};
// Batched launches (one unit per layer of a chunk): one workgroup per unit, all units side by side.  With >= 8 units in
// flight that beats spreading every unit over the chip: the 28 selections of a chunk take one workgroup's latency
// instead of 28 x 98 ranking workgroups.
__global__ __launch_bounds__(PSEL_BLOCK) void pivotkv_select_lds_units_kernel(SelUnits units, int L, int keep, int P,
                                                                              int reforge, int64_t pos_ld) {
    const rtk_select_unit& u = units.u[blockIdx.x];
    select_lds_body(u.score, u.mask, L, keep, u.pos, P, reforge, u.keep_idx, u.rank, u.pos_out, pos_ld);
}

// scratch layout inside a unit's workspace: sel [L] bytes | per-rank-workgroup counts | per-rank-workgroup minima
__host__ __device__ inline size_t sel_ws_cnt_off(int L) { return ((size_t)L + 255) & ~(size_t)255; }
__host__ __device__ inline size_t sel_ws_tmin_off(int L) {
    const size_t nb = ((size_t)L + 63) / 64;
    return sel_ws_cnt_off(L) + ((nb * 4 + 255) & ~(size_t)255);
}

// score[j] = mean_g( (sum_split partial[g,split,j]) / G ) for every unit that still carries partials
// (longvideo_cache.py:269-270): the deferred form of score_finalize_kernel, same fixed summation order.
__global__ __launch_bounds__(256) void finalize_units_kernel(SelUnits units, int Hkv, int RS, int G, int L) {
    extern __shared__ float fin_gs[];  // [Hkv][64]
    const rtk_select_unit& un = units.u[blockIdx.y];
    if (!un.partial) return;
    const int jl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jl;
    const int jc = min(j, L - 1);
    for (int g = part; g < Hkv; g += 4) {
        const float* p = un.partial + (size_t)g * RS * L + jc;
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
        un.score[j] = tot / (float)Hkv;
    }
}

// the same for RTK_BF16_REFROUND / RTK_F16_REFROUND partials ([Hkv*G][RS][L], per head): the reference's 16-bit roundings of the per-head
// sums and the two means (longvideo_cache.py:268-270), finalize_ref_column() in common.cuh
template <bool F16>
__global__ __launch_bounds__(256) void finalize_units_ref_kernel(SelUnits units, int Hkv, int RS, int G, int L) {
    const rtk_select_unit& un = units.u[blockIdx.y];
    if (!un.partial) return;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < L) un.score[j] = finalize_ref_column<F16>(un.partial, Hkv, RS, G, L, j);
}

__global__ __launch_bounds__(RANK_BLOCK) void pivotkv_rank_kernel(SelUnits units, int L, int keep, int reforge) {
    const rtk_select_unit& un = units.u[blockIdx.y];
    float* __restrict__ score = un.score;
    const uint8_t* __restrict__ mask = un.mask;
    const int64_t* __restrict__ pos = un.pos;
    uint8_t* __restrict__ sel = (uint8_t*)un.workspace;
    int32_t* __restrict__ blk_cnt = (int32_t*)((char*)un.workspace + sel_ws_cnt_off(L));
    int64_t* __restrict__ blk_tmin = (int64_t*)((char*)un.workspace + sel_ws_tmin_off(L));
    const int vec_ok = (((uintptr_t)score & 15) == 0 && ((uintptr_t)mask & 3) == 0) ? 1 : 0;
    extern __shared__ __attribute__((aligned(16))) uint32_t rk_keys[];  // [n4 * 4] keys (zero padded) + [RANK_SEG][RANK_TOK] counts
    const int tid = threadIdx.x;
    const int i0 = blockIdx.x * RANK_TOK;
    const int n4 = ((L + RANK_TOK - 1) / RANK_TOK) * (RANK_TOK / 4);   // keys padded to whole workgroup ranges
    int* part = (int*)(rk_keys + (size_t)n4 * 4);
    // every key of the chunk -> LDS (no global store in this loop: the loads pipeline freely).  Masked
    // tokens take the key of 1.0 (attn_weights.masked_fill_(mask, 1.), :274); padding keys are 0, smaller
    // than every real key, and only ever compared with '>'.
    const uint32_t one_key = f2key(1.0f);
    for (int j4 = tid; j4 < n4; j4 += RANK_BLOCK) {
        const int j = 4 * j4;
        uint4 kk = {0u, 0u, 0u, 0u};
        if (vec_ok && j + 3 < L) {
            const float4 sc = *(const float4*)(score + j);
            const uint32_t m = mask ? *(const uint32_t*)(mask + j) : 0u;
            kk.x = (m & 0x000000ffu) ? one_key : f2key(sc.x);
            kk.y = (m & 0x0000ff00u) ? one_key : f2key(sc.y);
            kk.z = (m & 0x00ff0000u) ? one_key : f2key(sc.z);
            kk.w = (m & 0xff000000u) ? one_key : f2key(sc.w);
        } else {
            uint32_t e[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j + u < L) e[u] = (mask && mask[j + u]) ? one_key : f2key(score[j + u]);
            kk = uint4{e[0], e[1], e[2], e[3]};
        }
        ((uint4*)rk_keys)[j4] = kk;
    }
    __syncthreads();
    const int t = tid % RANK_TOK;
    const int q = __builtin_amdgcn_readfirstlane(tid / RANK_TOK);   // wave index: uniform, loop bounds are scalar
    const int i = i0 + t;
    const uint32_t ki = (i < L) ? rk_keys[i] : 0xffffffffu;
    const int per = (n4 + RANK_SEG - 1) / RANK_SEG;
    const int j4b = q * per, j4e = min(n4, j4b + per);
    const int own4b = i0 / 4, own4e = own4b + RANK_TOK / 4;
    int cnt = 0;
    const uint4* k4 = (const uint4*)rk_keys;
    // Keys before the workgroup's own token range count on '>=' (ties: lower index first) = '> ki - 1', keys
    // after it on '>'; the own range is handled exactly below.  No real key is 0, so ki - 1 cannot wrap.
    auto count_range = [&](int b, int e, uint32_t thr) {
#pragma unroll 4
        for (int j4 = b; j4 < e; ++j4) {
            const uint4 kk = k4[j4];        // same address in every lane: LDS broadcast
            cnt += (kk.x > thr) + (kk.y > thr) + (kk.z > thr) + (kk.w > thr);
        }
    };
    count_range(j4b, min(j4e, own4b), ki - 1u);
    count_range(max(j4b, own4e), j4e, ki);
    {   // own range [i0, i0 + RANK_TOK): wave q takes keys i0 + 4q .. i0 + 4q + 3 (RANK_SEG * 4 == RANK_TOK)
        static_assert(RANK_SEG * 4 == RANK_TOK, "own-range split");
        const int j = i0 + 4 * q;
        const uint4 kk = k4[j >> 2];        // padding keys (j >= L) are 0: never counted
        cnt += (kk.x > ki || (kk.x == ki && j < i)) + (kk.y > ki || (kk.y == ki && j + 1 < i)) +
               (kk.z > ki || (kk.z == ki && j + 2 < i)) + (kk.w > ki || (kk.w == ki && j + 3 < i));
    }
    part[q * RANK_TOK + t] = cnt;
    __syncthreads();
    if (tid < WAVE) {  // first wave: lane t owns token i0 + t
        bool kept = false;
        long long tm = 0x7fffffffffffffffLL;
        if (i < L) {
            int rank = 0;
#pragma unroll
            for (int sg = 0; sg < RANK_SEG; ++sg) rank += part[sg * RANK_TOK + tid];
            kept = rank < keep;
            sel[i] = kept;
            if (ki == one_key && mask && mask[i]) score[i] = 1.0f;  // masked_fill_ is in place (:274)
            if (kept && pos && reforge) tm = pos[i];
        }
        const unsigned long long b = __ballot(kept);
        for (int o = 32; o > 0; o >>= 1) {
            const long long t2 = __shfl_xor(tm, o, WAVE);
            tm = min(tm, t2);
        }
        if (tid == 0) {
            blk_cnt[blockIdx.x] = (int32_t)__popcll(b);
            blk_tmin[blockIdx.x] = tm;  // min_temp_id partial (:293)
        }
    }
}

// ordered emit: one workgroup per 256 tokens.  The number of kept tokens before its range and min_temp_id
// come from the rank kernel's per-workgroup records (RANK_TOK tokens each).
__global__ __launch_bounds__(256) void pivotkv_emit_kernel(SelUnits units, int L, int keep, int P, int reforge,
                                                           int64_t pos_ld) {
    const rtk_select_unit& un = units.u[blockIdx.y];
    const uint8_t* __restrict__ sel = (const uint8_t*)un.workspace;
    const int32_t* __restrict__ blk_cnt = (const int32_t*)((const char*)un.workspace + sel_ws_cnt_off(L));
    const int64_t* __restrict__ blk_tmin = (const int64_t*)((const char*)un.workspace + sel_ws_tmin_off(L));
    const int64_t* __restrict__ pos = un.pos;
    int64_t* __restrict__ keep_idx = un.keep_idx;
    int32_t* __restrict__ rank = un.rank;
    int64_t* __restrict__ pos_out = un.pos_out;
    __shared__ int wtot[4];
    __shared__ int wsum[4];
    __shared__ long long wmin[4];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
    const int b0 = blockIdx.x * 256;
    const bool rf = pos && reforge;
    const int nb = (L + RANK_TOK - 1) / RANK_TOK, nb_before = b0 / RANK_TOK;
    const int i = b0 + tid;
    const int mine = (i < L) ? (int)sel[i] : 0;
    const long long t0 = (mine && pos) ? (long long)pos[i] : 0;
    int before = 0;
    long long mn = 0x7fffffffffffffffLL;
    for (int b = tid; b < nb; b += 256) {
        before += (b < nb_before) ? blk_cnt[b] : 0;
        if (rf) mn = min(mn, (long long)blk_tmin[b]);
    }
    // wave scan of the own flags + wave sums of `before` / min of mn, then combine across the 4 waves
    int inc = mine;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const int t = __shfl_up(inc, o, WAVE);
        if (lane >= o) inc += t;
    }
    before = wave_sum_i(before);
    for (int o = 32; o > 0; o >>= 1) {
        const long long t = __shfl_xor(mn, o, WAVE);
        mn = min(mn, t);
    }
    if (lane == WAVE - 1) wtot[wid] = inc;
    if (lane == 0) { wsum[wid] = before; wmin[wid] = mn; }
    __syncthreads();
    int r = wsum[0] + wsum[1] + wsum[2] + wsum[3] + inc - mine;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wid) r += wtot[w];
    mn = min(min(wmin[0], wmin[1]), min(wmin[2], wmin[3]));
    if (i >= L) return;
    if (!mine) {
        if (rank) rank[i] = -1;
        return;
    }
    keep_idx[r] = i;  // topk(keep).sort()  (:276-277)
    if (rank) rank[i] = r;
    if (pos) {
        // row 0: gathered id, rescaled when reforging: int64 -> float32 multiply -> truncation (:293-295)
        const float ratio = (float)((double)keep / (double)L);  // comp_ratio = keep_len / k_len (:294)
        pos_out[r] = rf ? mn + (long long)((float)(t0 - mn) * ratio) : t0;
        for (int p = 1; p < P; ++p) pos_out[(size_t)p * pos_ld + r] = pos[(size_t)p * L + i];  // :283-288
    }
}

// ------------------------------------------------------------------------------------------------
// eviction scan.  LPR lanes cooperate on one (head, token) row: lane c owns the 16-byte chunk c of the
// first half of the row and its rotation partner in the second half.
// ------------------------------------------------------------------------------------------------
template <int DT> struct Row16;
template <> struct Row16<RTK_F32> {
    static constexpr int VE = 4;
    __device__ static void unpack(const u32x4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    __device__ static u32x4 pack(const float* f) {
        return u32x4{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
    }
    __device__ static float rnd(float x) { return x; }
};
template <> struct Row16<RTK_BF16> {
    static constexpr int VE = 8;
    __device__ static void unpack(const u32x4& v, float* f) {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
        f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
        f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    __device__ static u32x4 pack(const float* f) {
        return u32x4{(uint32_t)f2bf(f[0]) | ((uint32_t)f2bf(f[1]) << 16), (uint32_t)f2bf(f[2]) | ((uint32_t)f2bf(f[3]) << 16),
                     (uint32_t)f2bf(f[4]) | ((uint32_t)f2bf(f[5]) << 16), (uint32_t)f2bf(f[6]) | ((uint32_t)f2bf(f[7]) << 16)};
    }
    __device__ static float rnd(float x) { return rbf(x); }
};
template <> struct Row16<RTK_F16> {
    static constexpr int VE = 8;
    __device__ static void unpack(const u32x4& v, float* f) {
        f[0] = H16<RTK_F16>::lo(v.x); f[1] = H16<RTK_F16>::hi(v.x); f[2] = H16<RTK_F16>::lo(v.y); f[3] = H16<RTK_F16>::hi(v.y);
        f[4] = H16<RTK_F16>::lo(v.z); f[5] = H16<RTK_F16>::hi(v.z); f[6] = H16<RTK_F16>::lo(v.w); f[7] = H16<RTK_F16>::hi(v.w);
    }
    __device__ static u32x4 pack(const float* f) {
        return u32x4{H16<RTK_F16>::pack2(f[0], f[1]), H16<RTK_F16>::pack2(f[2], f[3]), H16<RTK_F16>::pack2(f[4], f[5]),
                     H16<RTK_F16>::pack2(f[6], f[7])};
    }
    __device__ static float rnd(float x) { return rhf(x); }
};

// Two roles in one launch, so neither waits on the other's dependent loads:
//   blocks [0, append_blocks)   append: every 16-byte chunk of the chunk's K and V rows -> cache tail
//   blocks [append_blocks, ...) kept:   row r of the compacted cache <- token keep_idx[r]
//                                       (V copy; K copy, or un-rotated K re-rotated at its new position)
template <int DT>
__global__ __launch_bounds__(256) void evict_scan_kernel(const char* __restrict__ k, int64_t k_sh, int64_t k_sl,
                                                         const char* __restrict__ v, int64_t v_sh, int64_t v_sl,
                                                         const char* __restrict__ k_unrot, int Hkv, int L, int D,
                                                         const int64_t* __restrict__ keep_idx, int keep,
                                                         const float* __restrict__ cos_new,
                                                         const float* __restrict__ sin_new, char* __restrict__ k_tail,
                                                         char* __restrict__ v_tail, int64_t tail_sh,
                                                         char* __restrict__ k_kept, char* __restrict__ v_kept,
                                                         int64_t kept_sh, int append_blocks) {
    using R = Row16<DT>;
    constexpr int VE = R::VE;
    constexpr int ES = 16 / VE;
    if ((int)blockIdx.x < append_blocks) {  // ---- append (DynamicCache.update, :238) ----
        const int cpr = D / VE;             // 16-byte chunks per row
        const size_t total = (size_t)Hkv * L * cpr;
        for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < total;
             id += (size_t)append_blocks * blockDim.x) {
            const int c = (int)(id % cpr);
            const size_t hl = id / cpr;
            const int l = (int)(hl % L), h = (int)(hl / L);
            const u32x4 kk = *(const u32x4*)(k + ((size_t)h * k_sh + (size_t)l * k_sl + (size_t)c * VE) * ES);
            const u32x4 vv = *(const u32x4*)(v + ((size_t)h * v_sh + (size_t)l * v_sl + (size_t)c * VE) * ES);
            const size_t dst = ((size_t)h * tail_sh + (size_t)l * D + (size_t)c * VE) * ES;
            *(u32x4*)(k_tail + dst) = kk;
            *(u32x4*)(v_tail + dst) = vv;
        }
        return;
    }
    // ---- kept rows ----
    const int h2 = D / 2;
    const int lpr = h2 / VE;  // lanes per row: lane c owns chunk c of the first half + its rotation partner
    const size_t total = (size_t)Hkv * keep * lpr;
    const size_t nthreads = (size_t)(gridDim.x - append_blocks) * blockDim.x;
    for (size_t id = (size_t)(blockIdx.x - append_blocks) * blockDim.x + threadIdx.x; id < total; id += nthreads) {
        const int c = (int)(id % lpr);
        const size_t hr = id / lpr;
        const int r = (int)(hr % keep), h = (int)(hr / keep);
        const int l = (int)keep_idx[r];
        const int d = c * VE;
        const char* vr = v + ((size_t)h * v_sh + (size_t)l * v_sl) * ES;
        const u32x4 v_lo = *(const u32x4*)(vr + (size_t)d * ES), v_hi = *(const u32x4*)(vr + (size_t)(d + h2) * ES);
        char* kk = k_kept + ((size_t)h * kept_sh + (size_t)r * D) * ES;
        char* vk = v_kept + ((size_t)h * kept_sh + (size_t)r * D) * ES;
        if (!cos_new) {  // torch.gather(key_states, 2, keep)  (:279)
            const char* kr = k + ((size_t)h * k_sh + (size_t)l * k_sl) * ES;
            const u32x4 k_lo = *(const u32x4*)(kr + (size_t)d * ES), k_hi = *(const u32x4*)(kr + (size_t)(d + h2) * ES);
            *(u32x4*)(kk + (size_t)d * ES) = k_lo;
            *(u32x4*)(kk + (size_t)(d + h2) * ES) = k_hi;
        } else {         // reforge: kept K = un-rotated row rotated forward at its new position (:297-306)
            const char* ur = k_unrot + ((size_t)h * L + l) * D * ES;
            const u32x4 u_lo = *(const u32x4*)(ur + (size_t)d * ES), u_hi = *(const u32x4*)(ur + (size_t)(d + h2) * ES);
            const float* cr = cos_new + (size_t)r * D;
            const float* sr = sin_new + (size_t)r * D;
            float c1[VE], s1[VE], c2[VE], s2[VE];
#pragma unroll
            for (int e = 0; e < VE; e += 4) {
                *(float4*)(c1 + e) = *(const float4*)(cr + d + e);
                *(float4*)(s1 + e) = *(const float4*)(sr + d + e);
                *(float4*)(c2 + e) = *(const float4*)(cr + d + h2 + e);
                *(float4*)(s2 + e) = *(const float4*)(sr + d + h2 + e);
            }
            float x1[VE], x2[VE], o1[VE], o2[VE];
            R::unpack(u_lo, x1);
            R::unpack(u_hi, x2);
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                // (k*cos) + (rotate_half(k)*sin), one rounding per torch op, no fma contraction
                o1[e] = R::rnd(__fadd_rn(R::rnd(__fmul_rn(x1[e], c1[e])), R::rnd(__fmul_rn(-x2[e], s1[e]))));
                o2[e] = R::rnd(__fadd_rn(R::rnd(__fmul_rn(x2[e], c2[e])), R::rnd(__fmul_rn(x1[e], s2[e]))));
            }
            *(u32x4*)(kk + (size_t)d * ES) = R::pack(o1);
            *(u32x4*)(kk + (size_t)(d + h2) * ES) = R::pack(o2);
        }
        *(u32x4*)(vk + (size_t)d * ES) = v_lo;  // torch.gather(value_states, 2, keep)  (:280)
        *(u32x4*)(vk + (size_t)(d + h2) * ES) = v_hi;
    }
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const char* __restrict__ src, int64_t src_sh_bytes,
                                                        char* __restrict__ dst, int64_t dst_sh_bytes, int H,
                                                        size_t row_block_bytes) {
    // each head's [rows,D] block is contiguous: copy it as 16-byte vectors
    const size_t vec_per_head = row_block_bytes / 16;
    const size_t total = (size_t)H * vec_per_head;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (size_t)gridDim.x * blockDim.x) {
        const size_t h = id / vec_per_head, i = id % vec_per_head;
        *(u32x4*)(dst + h * dst_sh_bytes + i * 16) = *(const u32x4*)(src + h * src_sh_bytes + i * 16);
    }
}

// commit: both staged blocks (K and V) in one launch; blockIdx.y selects the tensor
__global__ __launch_bounds__(256) void commit_rows_kernel(const char* __restrict__ ks, const char* __restrict__ vs,
                                                          int64_t src_sh_bytes, char* __restrict__ kd,
                                                          char* __restrict__ vd, int64_t dst_sh_bytes, int H,
                                                          size_t row_block_bytes) {
    const char* src = blockIdx.y ? vs : ks;
    char* dst = blockIdx.y ? vd : kd;
    const size_t vec_per_head = row_block_bytes / 16;
    const size_t total = (size_t)H * vec_per_head;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (size_t)gridDim.x * blockDim.x) {
        const size_t h = id / vec_per_head, i = id % vec_per_head;
        *(u32x4*)(dst + h * dst_sh_bytes + i * 16) = *(const u32x4*)(src + h * src_sh_bytes + i * 16);
    }
}


// ------------------------------------------------------------------------------------------------
// Chunk-batched cache maintenance.  `PivotKVCache.update` only has to hand the layer's attention the
// uncompressed [prefix | chunk] view (longvideo_cache.py:238) and to decide which rows survive; the
// gather / re-rotation / compaction of ALL layers of a chunk is flushed in two launches from
// `after_forward` (the hook the reference calls after every video chunk, qwen2_vl.py:715-716):
//   append_kernel          per update: K and V rows of the chunk -> cache tail (pure streaming copy)
//   evict_batched_kernel   per flush : every pending (layer, chunk) unit in one launch (blockIdx.y = unit)
//   commit_batched_kernel  per flush : staged rows -> head of the tail, every unit in one launch
// A single unit is 6.5 MB of traffic (a 10 us launch is ramp-dominated); 28 layers per launch keep every
// CU streaming.
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void append_kernel(const char* __restrict__ k, int64_t k_sh, int64_t k_sl,
                                                     const char* __restrict__ v, int64_t v_sh, int64_t v_sl, int Hkv,
                                                     int L, int D, char* __restrict__ k_tail,
                                                     char* __restrict__ v_tail, int64_t tail_sh) {
    using R = Row16<DT>;
    constexpr int VE = R::VE;
    constexpr int ES = 16 / VE;
    constexpr int U = 4;                // 16-byte chunks per thread per tensor, all loads before the stores
    const int cpr = D / VE;             // 16-byte chunks per row
    const size_t total = (size_t)Hkv * L * cpr;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; id < total; id += U * stride) {
        u32x4 kk[U], vv[U];
        size_t dst[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = id + u * stride;
            const size_t ic = i < total ? i : id;
            const int c = (int)(ic % cpr);
            const size_t hl = ic / cpr;
            const int l = (int)(hl % L), h = (int)(hl / L);
            kk[u] = *(const u32x4*)(k + ((size_t)h * k_sh + (size_t)l * k_sl + (size_t)c * VE) * ES);
            vv[u] = *(const u32x4*)(v + ((size_t)h * v_sh + (size_t)l * v_sl + (size_t)c * VE) * ES);
            dst[u] = ((size_t)h * tail_sh + (size_t)l * D + (size_t)c * VE) * ES;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (id + u * stride < total) {
                *(u32x4*)(k_tail + dst[u]) = kk[u];
                *(u32x4*)(v_tail + dst[u]) = vv[u];
            }
        }
    }
}

struct EvictUnits {
    rtk_evict_unit u[RTK_EVICT_MAX_UNITS];
};

// One thread owns, for ONE kept row r of a unit, the 16-byte chunk c of the first half of the row and its
// rotation partner in the second half, and walks the KV heads with it: the row's cos/sin (reforge) are
// loaded once and reused by every head, and all of a head group's loads are issued before its stores.
// NATIVE: the cos/sin of the kept rows' NEW ids are computed here (rope_table_kernel's arithmetic on pos_src, 8 or 16
// sincos_cr per thread, reused by all KV heads) instead of being read from fp32 tables another launch wrote:
// one launch and 2 x keep x D x 4 bytes of write + read per unit less.
template <int DT, int HU, bool NATIVE>
__global__ __launch_bounds__(256) void evict_batched_kernel(EvictUnits units, int Hkv, int D, int keep, int P,
                                                            const float* __restrict__ inv_freq, float scaling,
                                                            RowSel rs, int round_bf16, int low_only) {
    using R = Row16<DT>;
    constexpr int VE = R::VE;
    constexpr int ES = 16 / VE;
    const rtk_evict_unit& un = units.u[blockIdx.y];
    const int h2 = D / 2;
    const int lpr = h2 / VE;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    // position ids of the kept tokens -> the layer's position cache (longvideo_cache.py:308-309)
    if (un.pos_dst && id < P * keep) {
        const int p = id / keep, r = id - p * keep;
        un.pos_dst[(size_t)p * un.pos_dst_stride + r] = un.pos_src[(size_t)p * un.pos_src_stride + r];
    }
    const int r = id / lpr;
    if (r >= keep) return;
    const int d = (id - r * lpr) * VE;
    const int l = (int)un.keep_idx[r];
    const char* ks = (const char*)un.k_src;
    const char* vs = (const char*)un.v_src;
    char* kd = (char*)un.k_dst;
    char* vd = (char*)un.v_dst;
    // a NULL destination skips the tensor: keep-all chunks leave V (and an unchanged K) where the append put them
    const bool reforge = (NATIVE || un.cos_new != nullptr) && kd != nullptr;
    // low_only: rows copied verbatim (V; K without reforge) are staged only when their source lies inside the
    // destination range [0, keep) of the tail they will overwrite; the others are moved in place by place_batched_kernel
    const bool stage_row = !(low_only & 1) || l < keep;
    // low_only bit 1: K comes from a buffer of its own (the un-rotated copy of a deferred re-rotation), not from the
    // tail it is written to: every kept K row is copied
    const bool copy_k = kd != nullptr && !reforge && (stage_row || (low_only & 2)), copy_rows = vd != nullptr && stage_row;
    u32x4 k_lo[HU], k_hi[HU], v_lo[HU], v_hi[HU];
    auto load_batch = [&](int hb) {
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            const int h = min(hb + u, Hkv - 1);
            const char* kr = ks + ((size_t)h * un.k_src_stride_h + (size_t)l * D) * ES;
            const char* vr = vs + ((size_t)h * un.v_src_stride_h + (size_t)l * D) * ES;
            if (reforge || copy_k) {
                k_lo[u] = *(const u32x4*)(kr + (size_t)d * ES);
                k_hi[u] = *(const u32x4*)(kr + (size_t)(d + h2) * ES);
            }
            if (copy_rows) {
                v_lo[u] = *(const u32x4*)(vr + (size_t)d * ES);
                v_hi[u] = *(const u32x4*)(vr + (size_t)(d + h2) * ES);
            }
        }
    };
    if (!reforge && !copy_k && !copy_rows) return;
    load_batch(0);   // the rows are requested before the table arithmetic / table reads below
    float c1[VE], s1[VE], c2[VE], s2[VE];
    if (NATIVE && reforge) {
        float pid[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) pid[p] = (float)un.pos_src[(size_t)min(p, P - 1) * un.pos_src_stride + r];
        rope_chunk<VE>(inv_freq, rs, d, h2, pid, scaling, round_bf16, c1, s1, c2, s2);
    } else if (reforge) {
        const float* cr = un.cos_new + (size_t)r * D;
        const float* sr = un.sin_new + (size_t)r * D;
#pragma unroll
        for (int e = 0; e < VE; e += 4) {
            *(float4*)(c1 + e) = *(const float4*)(cr + d + e);
            *(float4*)(s1 + e) = *(const float4*)(sr + d + e);
            *(float4*)(c2 + e) = *(const float4*)(cr + d + h2 + e);
            *(float4*)(s2 + e) = *(const float4*)(sr + d + h2 + e);
        }
    }
    for (int hb = 0; hb < Hkv; hb += HU) {
        if (hb > 0) load_batch(hb);
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            const int h = hb + u;
            if (h >= Hkv) break;
            char* ko = kd + ((size_t)h * un.k_dst_stride_h + (size_t)r * D) * ES;
            char* vo = vd + ((size_t)h * un.v_dst_stride_h + (size_t)r * D) * ES;
            if (reforge) {  // kept K = un-rotated row rotated forward at its new position (:297-306)
                // (the 16-bit roundings through the packed converts: the integer sequence made this kernel VALU-bound)
                u32x4 olo, ohi;
                rotate_chunk_pair<DT>(k_lo[u], k_hi[u], c1, s1, c2, s2, olo, ohi);
                *(u32x4*)(ko + (size_t)d * ES) = olo;
                *(u32x4*)(ko + (size_t)(d + h2) * ES) = ohi;
            } else if (copy_k) {   // torch.gather(key_states, 2, keep)  (:279)
                *(u32x4*)(ko + (size_t)d * ES) = k_lo[u];
                *(u32x4*)(ko + (size_t)(d + h2) * ES) = k_hi[u];
            }
            if (copy_rows) {
                *(u32x4*)(vo + (size_t)d * ES) = v_lo[u];  // torch.gather(value_states, 2, keep)  (:280)
                *(u32x4*)(vo + (size_t)(d + h2) * ES) = v_hi[u];
            }
        }
    }
}

struct CopyUnits {
    rtk_copy_unit u[RTK_COPY_MAX_UNITS];
};

// dst[h][0:rows] = src[h][0:rows] for every unit (blockIdx.y); each head's row block is contiguous
__global__ __launch_bounds__(256) void commit_batched_kernel(CopyUnits units, int H, size_t row_block_bytes) {
    const rtk_copy_unit& un = units.u[blockIdx.y];
    const char* src = (const char*)un.src;
    char* dst = (char*)un.dst;
    const size_t vec_per_head = row_block_bytes / 16;
    const size_t total = (size_t)H * vec_per_head;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    constexpr int U = 4;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += U * stride) {
        u32x4 t[U];
        size_t o[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = id + u * stride;
            const size_t ic = i < total ? i : id;
            const size_t h = ic / vec_per_head, j = ic % vec_per_head;
            t[u] = *(const u32x4*)(src + h * un.src_stride_h_bytes + j * 16);
            o[u] = h * un.dst_stride_h_bytes + j * 16;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (id + u * stride < total) *(u32x4*)(dst + o[u]) = t[u];
    }
}

struct PlaceUnits {
    rtk_place_unit u[RTK_PLACE_MAX_UNITS];
};

// P13 (longvideo_cache.py:313-318) without a full staging copy: kept row r of a unit goes to tail[h][r].  Its source
// is the chunk row keep_idx[r] of the same tail; rows whose source lies below `keep` would race with the rows being
// written there, so an earlier launch (evict_batched_kernel, low_only) parked exactly those in `stage`; every other
// row is read where it sits - beyond the destination range, which nothing writes.  ~ratio of the rows take the hop.
__global__ __launch_bounds__(256) void place_batched_kernel(PlaceUnits units, int H, int keep, int vec_per_row) {
    const rtk_place_unit& un = units.u[blockIdx.y];
    const size_t total = (size_t)H * keep * vec_per_row;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    constexpr int U = 4;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += U * stride) {
        u32x4 t[U];
        size_t o[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = id + u * stride;
            const size_t ic = i < total ? i : id;
            const int j = (int)(ic % vec_per_row);
            const size_t hr = ic / vec_per_row;
            const int r = (int)(hr % keep), h = (int)(hr / keep);
            const int64_t l = un.keep_idx[r];
            const char* src = l < keep ? (const char*)un.stage + ((size_t)h * un.stage_stride_h_bytes + (size_t)r * vec_per_row * 16)
                                       : (const char*)un.tail + ((size_t)h * un.tail_stride_h_bytes + (size_t)l * vec_per_row * 16);
            t[u] = *(const u32x4*)(src + (size_t)j * 16);
            o[u] = (size_t)h * un.tail_stride_h_bytes + ((size_t)r * vec_per_row + j) * 16;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (id + u * stride < total) *(u32x4*)((char*)un.tail + o[u]) = t[u];
    }
}

}  // namespace rtk

using namespace rtk;

extern "C" int rtk_pivotkv_commit(const void* k_stage, const void* v_stage, int64_t stage_stride_h, void* k_dst,
                                  void* v_dst, int64_t dst_stride_h, int H, int rows, int D, int dtype,
                                  rtk_stream_t stream) {
    RTK_CHECK_ARG(k_stage && v_stage && k_dst && v_dst, "rtk_pivotkv_commit: NULL pointer");
    RTK_CHECK_ARG(H >= 1 && rows >= 0 && D >= 1, "rtk_pivotkv_commit: bad shape");
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_F16, "rtk_pivotkv_commit: unsupported dtype %d", dtype);
    if (rows == 0) return RTK_OK;
    const size_t es = dtype != RTK_F32 ? 2 : 4;
    const size_t blk = (size_t)rows * D * es;
    if (blk % 16 || (stage_stride_h * es) % 16 || (dst_stride_h * es) % 16 ||
        (((uintptr_t)k_stage | (uintptr_t)v_stage | (uintptr_t)k_dst | (uintptr_t)v_dst) & 15)) {
        set_error("rtk_pivotkv_commit: blocks must be 16-byte aligned");
        return RTK_EUNSUPPORTED;
    }
    const size_t total = (size_t)H * (blk / 16);
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 4096);
    RTK_LAUNCH(KID_COPY, commit_rows_kernel, dim3(grid, 2), dim3(256), 0, (hipStream_t)stream, (const char*)k_stage,
               (const char*)v_stage, (int64_t)(stage_stride_h * es), (char*)k_dst, (char*)v_dst,
               (int64_t)(dst_stride_h * es), H, blk);
    RTK_LAUNCH_CHECK("commit_rows_kernel");
    return RTK_OK;
}

extern "C" size_t rtk_pivotkv_select_workspace_bytes(int L) {
    if (L <= 0) return 0;
    const size_t nb = ((size_t)L + RANK_TOK - 1) / RANK_TOK;
    return sel_ws_tmin_off(L) + ((nb * 8 + 255) & ~(size_t)255);
}

static bool chipwide_ok(int L) {
    const size_t lds = ((size_t)((L + RANK_TOK - 1) / RANK_TOK) * RANK_TOK + RANK_BLOCK) * sizeof(uint32_t);
    return L >= 512 && lds <= 160 * 1024;
}

// finalize (units that carry partials) -> rank -> emit, every unit in the same three launches
constexpr int UNITS_ONE_WG = 8;   // batched launches with at least this many units select one workgroup per unit
// > 64 KiB of dynamic LDS: opt-in per device, remembered in one atomic bit per device
static void select_lds_opt_in() {
    static std::atomic<uint64_t> opted{0};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    const uint64_t bit = 1ull << (dev_id & 63);
    if (!(opted.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute((const void*)pivotkv_rank_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)pivotkv_select_lds_units_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        opted.fetch_or(bit, std::memory_order_relaxed);
    }
}

static int select_units(const rtk_select_unit* units, int n, int Hkv, int RS, int G, int L, int keep, int P, int reforge,
                        int64_t pos_out_stride, hipStream_t st, int refround = 0 /* 1: bf16 chain, 2: fp16 chain */) {
    select_lds_opt_in();
    const size_t lds = ((size_t)((L + RANK_TOK - 1) / RANK_TOK) * RANK_TOK + RANK_BLOCK) * sizeof(uint32_t);
    for (int b = 0; b < n; b += RTK_SELECT_MAX_UNITS) {
        const int m = std::min(RTK_SELECT_MAX_UNITS, n - b);
        SelUnits su;
        bool any_partial = false;
        for (int i = 0; i < RTK_SELECT_MAX_UNITS; ++i) {
            su.u[i] = units[b + std::min(i, m - 1)];
            any_partial = any_partial || (i < m && su.u[i].partial != nullptr);
        }
        if (any_partial) {
            if (refround == 2)
                RTK_LAUNCH(KID_FINALIZE, finalize_units_ref_kernel<true>, dim3((L + 255) / 256, m), dim3(256), 0, st, su, Hkv, RS, G, L);
            else if (refround)
                RTK_LAUNCH(KID_FINALIZE, finalize_units_ref_kernel<false>, dim3((L + 255) / 256, m), dim3(256), 0, st, su, Hkv, RS, G, L);
            else
                RTK_LAUNCH(KID_FINALIZE, finalize_units_kernel, dim3((L + 63) / 64, m), dim3(256),
                           (size_t)Hkv * 64 * sizeof(float), st, su, Hkv, RS, G, L);
            RTK_LAUNCH_CHECK("finalize_units_kernel");
        }
        const int per = (L + PSEL_BLOCK - 1) / PSEL_BLOCK;
        if (m >= UNITS_ONE_WG && per <= SEL_LDS_MAX_PER) {
            RTK_LAUNCH(KID_PSEL, pivotkv_select_lds_units_kernel, dim3(m), dim3(PSEL_BLOCK), select_lds_bytes(L, keep), st, su, L,
                       keep, P, reforge, pos_out_stride);
            RTK_LAUNCH_CHECK("pivotkv_select_lds_units_kernel");
            continue;
        }
        RTK_LAUNCH(KID_PSEL, pivotkv_rank_kernel, dim3((L + RANK_TOK - 1) / RANK_TOK, m), dim3(RANK_BLOCK), lds, st, su, L, keep,
                   reforge);
        RTK_LAUNCH_CHECK("pivotkv_rank_kernel");
        RTK_LAUNCH(KID_PEMIT, pivotkv_emit_kernel, dim3((L + 255) / 256, m), dim3(256), 0, st, su, L, keep, P, reforge,
                   pos_out_stride);
        RTK_LAUNCH_CHECK("pivotkv_emit_kernel");
    }
    return RTK_OK;
}

extern "C" int rtk_pivotkv_select_batched(const rtk_select_unit* units, int n_units, int Hkv, int RS, int G, int L,
                                          int keep, int P, int reforge, int64_t pos_out_stride, int score_dtype,
                                          rtk_stream_t stream) {
    RTK_CHECK_ARG(units && n_units >= 1, "rtk_pivotkv_select_batched: no units");
    RTK_CHECK_ARG(L >= 1 && keep >= 1 && keep <= L, "rtk_pivotkv_select_batched: keep=%d out of range for L=%d", keep, L);
    RTK_CHECK_ARG(P == 0 || P == 1 || P == 3, "rtk_pivotkv_select_batched: P must be 0, 1 or 3, got %d", P);
    RTK_CHECK_ARG(P == 0 || pos_out_stride >= keep, "rtk_pivotkv_select_batched: pos_out_stride < keep");
    bool partials = false;
    for (int i = 0; i < n_units; ++i) {
        const rtk_select_unit& u = units[i];
        RTK_CHECK_ARG(u.score && u.keep_idx && u.workspace, "rtk_pivotkv_select_batched: unit %d: NULL pointer", i);
        RTK_CHECK_ARG(((uintptr_t)u.workspace & 255) == 0, "rtk_pivotkv_select_batched: unit %d: workspace must be 256-byte aligned", i);
        RTK_CHECK_ARG((u.pos == nullptr) == (u.pos_out == nullptr), "rtk_pivotkv_select_batched: unit %d: pos and pos_out go together", i);
        RTK_CHECK_ARG((u.pos != nullptr) == (P > 0), "rtk_pivotkv_select_batched: unit %d: pos must be given iff P > 0", i);
        partials = partials || u.partial;
    }
    RTK_CHECK_ARG(!partials || (Hkv >= 1 && RS >= 1 && G >= 1), "rtk_pivotkv_select_batched: partials need Hkv, RS, G");
    if (!chipwide_ok(L)) {
        set_error("rtk_pivotkv_select_batched: L=%d is outside the chip-wide selection path (use rtk_pivotkv_select)", L);
        return RTK_EUNSUPPORTED;
    }
    return select_units(units, n_units, Hkv, RS, G, L, keep, P, reforge, pos_out_stride, (hipStream_t)stream,
                        (score_dtype & ~RTK_SCORE_MANY_UNITS) == RTK_BF16_REFROUND ? 1
                        : ((score_dtype & ~RTK_SCORE_MANY_UNITS) == RTK_F16_REFROUND ? 2 : 0));
}

extern "C" int rtk_pivotkv_select(float* score, const uint8_t* mask, int L, int keep, const int64_t* pos, int P,
                                  int reforge, int64_t* keep_idx, int32_t* rank, int64_t* pos_out,
                                  int64_t pos_out_stride, void* workspace, size_t workspace_bytes,
                                  rtk_stream_t stream) {
    RTK_CHECK_ARG(score && keep_idx, "rtk_pivotkv_select: NULL pointer");
    RTK_CHECK_ARG(L >= 1 && keep >= 1 && keep <= L, "rtk_pivotkv_select: keep=%d out of range for L=%d", keep, L);
    RTK_CHECK_ARG((pos == nullptr) == (pos_out == nullptr), "rtk_pivotkv_select: pos and pos_out go together");
    RTK_CHECK_ARG(!pos || P == 1 || P == 3, "rtk_pivotkv_select: P must be 1 or 3, got %d", P);
    RTK_CHECK_ARG(!pos || pos_out_stride >= keep, "rtk_pivotkv_select: pos_out_stride %lld < keep %d",
                  (long long)pos_out_stride, keep);
    hipStream_t st = (hipStream_t)stream;
    if (workspace && workspace_bytes >= rtk_pivotkv_select_workspace_bytes(L) && ((uintptr_t)workspace & 255) == 0 &&
        chipwide_ok(L)) {
        // chip-wide path: rank by counting (every CU), then ordered emit
        rtk_select_unit u;
        u.partial = nullptr;
        u.score = score;
        u.mask = mask;
        u.pos = pos;
        u.keep_idx = keep_idx;
        u.rank = rank;
        u.pos_out = pos_out;
        u.workspace = workspace;
        return select_units(&u, 1, 0, 0, 0, L, keep, pos ? P : 0, reforge, pos_out_stride, st);
    }
    RTK_CHECK_ARG(rank, "rtk_pivotkv_select: the one-workgroup path needs the rank buffer");
    const int per = (L + PSEL_BLOCK - 1) / PSEL_BLOCK;
    if (per <= SEL_LDS_MAX_PER) {
        rtk_select_unit u;
        u.partial = nullptr;
        u.score = score;
        u.mask = mask;
        u.pos = pos;
        u.keep_idx = keep_idx;
        u.rank = rank;
        u.pos_out = pos_out;
        u.workspace = nullptr;
        SelUnits su;
        for (int i = 0; i < RTK_SELECT_MAX_UNITS; ++i) su.u[i] = u;
        select_lds_opt_in();
        RTK_LAUNCH(KID_PSEL, pivotkv_select_lds_units_kernel, dim3(1), dim3(PSEL_BLOCK), select_lds_bytes(L, keep), st, su, L, keep,
                   pos ? P : 0, reforge, pos_out_stride);
    } else   // longer rows: keys re-read from memory on every pass
        RTK_LAUNCH(KID_PSEL, pivotkv_select_kernel, dim3(1), dim3(PSEL_BLOCK), 0, st, score, mask, L, keep, pos, P,
                   reforge, keep_idx, rank, pos_out, pos_out_stride);
    RTK_LAUNCH_CHECK("pivotkv_select_kernel");
    return RTK_OK;
}

extern "C" int rtk_pivotkv_evict(const void* k, int64_t k_stride_h, int64_t k_stride_l, const void* v,
                                 int64_t v_stride_h, int64_t v_stride_l, const void* k_unrot, int Hkv, int L, int D,
                                 int dtype, const int64_t* keep_idx, int keep, const float* cos_new,
                                 const float* sin_new, void* k_tail, void* v_tail, int64_t tail_stride_h, void* k_kept,
                                 void* v_kept, int64_t kept_stride_h, rtk_stream_t stream) {
    RTK_CHECK_ARG(k && v && keep_idx && k_kept && v_kept, "rtk_pivotkv_evict: NULL pointer");
    RTK_CHECK_ARG(Hkv >= 1 && L >= 1 && keep >= 1 && keep <= L, "rtk_pivotkv_evict: bad shape");
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_F16, "rtk_pivotkv_evict: unsupported dtype %d", dtype);
    RTK_CHECK_ARG((cos_new == nullptr) == (sin_new == nullptr), "rtk_pivotkv_evict: cos_new and sin_new go together");
    RTK_CHECK_ARG(!cos_new || k_unrot, "rtk_pivotkv_evict: reforge needs k_unrot");
    RTK_CHECK_ARG((k_tail == nullptr) == (v_tail == nullptr), "rtk_pivotkv_evict: k_tail and v_tail go together");
    const int ve = dtype != RTK_F32 ? 8 : 4;
    const int es = dtype != RTK_F32 ? 2 : 4;
    if (D % (2 * ve) != 0) {
        set_error("rtk_pivotkv_evict: head_dim %d must be a multiple of %d for this dtype", D, 2 * ve);
        return RTK_EUNSUPPORTED;
    }
    const bool aligned = (k_stride_h * es) % 16 == 0 && (k_stride_l * es) % 16 == 0 && (v_stride_h * es) % 16 == 0 &&
                         (v_stride_l * es) % 16 == 0 && (tail_stride_h * es) % 16 == 0 && (kept_stride_h * es) % 16 == 0 &&
                         (((uintptr_t)k | (uintptr_t)v | (uintptr_t)k_unrot | (uintptr_t)k_tail | (uintptr_t)v_tail |
                           (uintptr_t)k_kept | (uintptr_t)v_kept | (uintptr_t)cos_new | (uintptr_t)sin_new) & 15) == 0;
    if (!aligned) {
        set_error("rtk_pivotkv_evict: pointers and strides must be 16-byte aligned");
        return RTK_EUNSUPPORTED;
    }
    const size_t append_chunks = k_tail ? (size_t)Hkv * L * (D / ve) : 0;
    const size_t kept_threads = (size_t)Hkv * keep * (D / 2 / ve);
    const unsigned append_blocks = (unsigned)std::min<size_t>((append_chunks + 255) / 256, 8192);
    const unsigned kept_blocks = (unsigned)std::min<size_t>((kept_threads + 255) / 256, 8192);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RTK_BF16)
        RTK_LAUNCH(KID_EVICT, evict_scan_kernel<RTK_BF16>, dim3(append_blocks + kept_blocks), dim3(256), 0, st,
                   (const char*)k, k_stride_h, k_stride_l, (const char*)v, v_stride_h, v_stride_l, (const char*)k_unrot,
                   Hkv, L, D, keep_idx, keep, cos_new, sin_new, (char*)k_tail, (char*)v_tail, tail_stride_h,
                   (char*)k_kept, (char*)v_kept, kept_stride_h, (int)append_blocks);
    else if (dtype == RTK_F16)
        RTK_LAUNCH(KID_EVICT, evict_scan_kernel<RTK_F16>, dim3(append_blocks + kept_blocks), dim3(256), 0, st,
                   (const char*)k, k_stride_h, k_stride_l, (const char*)v, v_stride_h, v_stride_l, (const char*)k_unrot,
                   Hkv, L, D, keep_idx, keep, cos_new, sin_new, (char*)k_tail, (char*)v_tail, tail_stride_h,
                   (char*)k_kept, (char*)v_kept, kept_stride_h, (int)append_blocks);
    else
        RTK_LAUNCH(KID_EVICT, evict_scan_kernel<RTK_F32>, dim3(append_blocks + kept_blocks), dim3(256), 0, st,
                   (const char*)k, k_stride_h, k_stride_l, (const char*)v, v_stride_h, v_stride_l, (const char*)k_unrot,
                   Hkv, L, D, keep_idx, keep, cos_new, sin_new, (char*)k_tail, (char*)v_tail, tail_stride_h,
                   (char*)k_kept, (char*)v_kept, kept_stride_h, (int)append_blocks);
    RTK_LAUNCH_CHECK("evict_scan_kernel");
    return RTK_OK;
}

extern "C" int rtk_copy_rows(const void* src, int64_t src_stride_h, void* dst, int64_t dst_stride_h, int H, int rows,
                             int D, int dtype, rtk_stream_t stream) {
    RTK_CHECK_ARG(src && dst, "rtk_copy_rows: NULL pointer");
    RTK_CHECK_ARG(H >= 1 && rows >= 0 && D >= 1, "rtk_copy_rows: bad shape");
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_F16, "rtk_copy_rows: unsupported dtype %d", dtype);
    if (rows == 0) return RTK_OK;
    const size_t es = dtype != RTK_F32 ? 2 : 4;
    const size_t blk = (size_t)rows * D * es;
    if (blk % 16 || (src_stride_h * es) % 16 || (dst_stride_h * es) % 16 || (((uintptr_t)src | (uintptr_t)dst) & 15)) {
        set_error("rtk_copy_rows: blocks must be 16-byte aligned");
        return RTK_EUNSUPPORTED;
    }
    const size_t total = (size_t)H * (blk / 16);
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 8192);
    RTK_LAUNCH(KID_COPY, copy_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const char*)src,
                       (int64_t)(src_stride_h * es), (char*)dst, (int64_t)(dst_stride_h * es), H, blk);
    RTK_LAUNCH_CHECK("copy_rows_kernel");
    return RTK_OK;
}

extern "C" int rtk_pivotkv_append(const void* k, int64_t k_stride_h, int64_t k_stride_l, const void* v,
                                  int64_t v_stride_h, int64_t v_stride_l, int Hkv, int L, int D, int dtype,
                                  void* k_tail, void* v_tail, int64_t tail_stride_h, rtk_stream_t stream) {
    RTK_CHECK_ARG(k && v && k_tail && v_tail, "rtk_pivotkv_append: NULL pointer");
    RTK_CHECK_ARG(Hkv >= 1 && L >= 1 && D >= 1, "rtk_pivotkv_append: bad shape");
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_F16, "rtk_pivotkv_append: unsupported dtype %d", dtype);
    const int ve = dtype != RTK_F32 ? 8 : 4;
    const int es = dtype != RTK_F32 ? 2 : 4;
    const bool aligned = D % ve == 0 && (k_stride_h * es) % 16 == 0 && (k_stride_l * es) % 16 == 0 &&
                         (v_stride_h * es) % 16 == 0 && (v_stride_l * es) % 16 == 0 && (tail_stride_h * es) % 16 == 0 &&
                         (((uintptr_t)k | (uintptr_t)v | (uintptr_t)k_tail | (uintptr_t)v_tail) & 15) == 0;
    if (!aligned) {
        set_error("rtk_pivotkv_append: pointers, strides and head_dim rows must be 16-byte aligned");
        return RTK_EUNSUPPORTED;
    }
    const size_t chunks = (size_t)Hkv * L * (D / ve);
    const unsigned grid = (unsigned)std::min<size_t>((chunks + 4 * 256 - 1) / (4 * 256), 4096);
    hipStream_t st = (hipStream_t)stream;
    if (dtype != RTK_F32)   // a pure copy: the 2-byte instantiation serves both 16-bit formats
        RTK_LAUNCH(KID_APPEND, append_kernel<RTK_BF16>, dim3(grid), dim3(256), 0, st, (const char*)k, k_stride_h,
                   k_stride_l, (const char*)v, v_stride_h, v_stride_l, Hkv, L, D, (char*)k_tail, (char*)v_tail,
                   tail_stride_h);
    else
        RTK_LAUNCH(KID_APPEND, append_kernel<RTK_F32>, dim3(grid), dim3(256), 0, st, (const char*)k, k_stride_h,
                   k_stride_l, (const char*)v, v_stride_h, v_stride_l, Hkv, L, D, (char*)k_tail, (char*)v_tail,
                   tail_stride_h);
    RTK_LAUNCH_CHECK("append_kernel");
    return RTK_OK;
}

static int evict_batched_impl(const rtk_evict_unit* units, int n_units, int Hkv, int D, int keep, int P, int dtype,
                              const float* inv_freq, float scaling, const RowSel* rsel, int round_bf16, int low_only,
                              rtk_stream_t stream) {
    const bool native = inv_freq != nullptr;
    RowSel rs;
    if (rsel) rs = *rsel;
    else for (int d = 0; d < 256; ++d) rs.row[d] = 0;
    RTK_CHECK_ARG(units && n_units >= 1, "rtk_pivotkv_evict_batched: no units");
    RTK_CHECK_ARG(Hkv >= 1 && keep >= 1 && D >= 2, "rtk_pivotkv_evict_batched: bad shape");
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_F16, "rtk_pivotkv_evict_batched: unsupported dtype %d", dtype);
    RTK_CHECK_ARG(P == 0 || P == 1 || P == 3, "rtk_pivotkv_evict_batched: P must be 0, 1 or 3, got %d", P);
    const int ve = dtype != RTK_F32 ? 8 : 4;
    const int es = dtype != RTK_F32 ? 2 : 4;
    if (D % (2 * ve) != 0) {
        set_error("rtk_pivotkv_evict_batched: head_dim %d must be a multiple of %d for this dtype", D, 2 * ve);
        return RTK_EUNSUPPORTED;
    }
    for (int i = 0; i < n_units; ++i) {
        const rtk_evict_unit& u = units[i];
        RTK_CHECK_ARG(u.k_src && u.v_src && u.keep_idx, "rtk_pivotkv_evict_batched: unit %d: NULL pointer", i);
        RTK_CHECK_ARG((u.cos_new == nullptr) == (u.sin_new == nullptr), "rtk_pivotkv_evict_batched: unit %d: cos_new and sin_new go together", i);
        RTK_CHECK_ARG((u.pos_dst == nullptr) || (u.pos_src != nullptr && P > 0), "rtk_pivotkv_evict_batched: unit %d: pos_dst needs pos_src and P", i);
        RTK_CHECK_ARG(!native || (u.pos_src != nullptr && P > 0 && u.cos_new == nullptr),
                      "rtk_pivotkv_evict_batched_rope: unit %d: needs pos_src (the new ids), P > 0 and no tables", i);
        const bool aligned = (u.k_src_stride_h * es) % 16 == 0 && (u.v_src_stride_h * es) % 16 == 0 &&
                             (u.k_dst_stride_h * es) % 16 == 0 && (u.v_dst_stride_h * es) % 16 == 0 &&
                             (((uintptr_t)u.k_src | (uintptr_t)u.v_src | (uintptr_t)u.k_dst | (uintptr_t)u.v_dst |
                               (uintptr_t)u.cos_new | (uintptr_t)u.sin_new) & 15) == 0;
        if (!aligned) {
            set_error("rtk_pivotkv_evict_batched: unit %d: pointers and strides must be 16-byte aligned", i);
            return RTK_EUNSUPPORTED;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const int threads = std::max(keep * (D / 2 / ve), P * keep);
    const unsigned gx = (unsigned)((threads + 255) / 256);
    for (int b = 0; b < n_units; b += RTK_EVICT_MAX_UNITS) {
        const int n = std::min(RTK_EVICT_MAX_UNITS, n_units - b);
        EvictUnits eu;
        for (int i = 0; i < n; ++i) eu.u[i] = units[b + i];
        for (int i = n; i < RTK_EVICT_MAX_UNITS; ++i) eu.u[i] = units[b];
#define RTK_EVB(DTV, HUV, NAT)                                                                                    \
    RTK_LAUNCH(KID_EVICTB, (evict_batched_kernel<DTV, HUV, NAT>), dim3(gx, n), dim3(256), 0, st, eu, Hkv, D, keep, P, \
               inv_freq, scaling, rs, round_bf16, low_only)
        if (dtype == RTK_BF16) {
            if (native) RTK_EVB(RTK_BF16, 4, true);
            else RTK_EVB(RTK_BF16, 4, false);
        } else if (dtype == RTK_F16) {
            if (native) RTK_EVB(RTK_F16, 4, true);
            else RTK_EVB(RTK_F16, 4, false);
        } else {
            if (native) RTK_EVB(RTK_F32, 2, true);
            else RTK_EVB(RTK_F32, 2, false);
        }
#undef RTK_EVB
        RTK_LAUNCH_CHECK("evict_batched_kernel");
    }
    return RTK_OK;
}

extern "C" int rtk_pivotkv_evict_batched(const rtk_evict_unit* units, int n_units, int Hkv, int D, int keep, int P,
                                         int dtype, int stage_low_only, rtk_stream_t stream) {
    return evict_batched_impl(units, n_units, Hkv, D, keep, P, dtype, nullptr, 0.f, nullptr, 0, stage_low_only, stream);
}

extern "C" int rtk_pivotkv_evict_batched_rope(const rtk_evict_unit* units, int n_units, int Hkv, int D, int keep, int P,
                                              int dtype, const float* inv_freq, float attention_scaling,
                                              const int* sections_host, int nsec, int round_bf16, int stage_low_only,
                                              rtk_stream_t stream) {
    RTK_CHECK_ARG(inv_freq, "rtk_pivotkv_evict_batched_rope: inv_freq is NULL");
    RTK_CHECK_ARG(P == 1 || P == 3, "rtk_pivotkv_evict_batched_rope: P must be 1 or 3, got %d", P);
    RowSel rs;
    const int rc = make_rowsel(rs, P, D, sections_host, nsec, "rtk_pivotkv_evict_batched_rope");
    if (rc != RTK_OK) return rc;
    return evict_batched_impl(units, n_units, Hkv, D, keep, P, dtype, inv_freq, attention_scaling, &rs, round_bf16,
                              stage_low_only, stream);
}

extern "C" int rtk_pivotkv_place_batched(const rtk_place_unit* units, int n_units, int H, int keep, int D, int dtype,
                                         rtk_stream_t stream) {
    RTK_CHECK_ARG(units && n_units >= 1, "rtk_pivotkv_place_batched: no units");
    RTK_CHECK_ARG(H >= 1 && keep >= 1 && D >= 1, "rtk_pivotkv_place_batched: bad shape");
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_F16, "rtk_pivotkv_place_batched: unsupported dtype %d", dtype);
    const size_t row = (size_t)D * (dtype != RTK_F32 ? 2 : 4);
    for (int i = 0; i < n_units; ++i) {
        const rtk_place_unit& u = units[i];
        RTK_CHECK_ARG(u.stage && u.tail && u.keep_idx, "rtk_pivotkv_place_batched: unit %d: NULL pointer", i);
        if (row % 16 || u.stage_stride_h_bytes % 16 || u.tail_stride_h_bytes % 16 || (((uintptr_t)u.stage | (uintptr_t)u.tail) & 15)) {
            set_error("rtk_pivotkv_place_batched: unit %d: rows must be 16-byte aligned", i);
            return RTK_EUNSUPPORTED;
        }
    }
    const size_t total = (size_t)H * keep * (row / 16);
    const unsigned gx = (unsigned)std::min<size_t>((total + 4 * 256 - 1) / (4 * 256), 1024);
    hipStream_t st = (hipStream_t)stream;
    for (int b = 0; b < n_units; b += RTK_PLACE_MAX_UNITS) {
        const int n = std::min(RTK_PLACE_MAX_UNITS, n_units - b);
        PlaceUnits pu;
        for (int i = 0; i < n; ++i) pu.u[i] = units[b + i];
        for (int i = n; i < RTK_PLACE_MAX_UNITS; ++i) pu.u[i] = units[b];
        RTK_LAUNCH(KID_COMMITB, place_batched_kernel, dim3(gx, n), dim3(256), 0, st, pu, H, keep, (int)(row / 16));
        RTK_LAUNCH_CHECK("place_batched_kernel");
    }
    return RTK_OK;
}

extern "C" int rtk_pivotkv_commit_batched(const rtk_copy_unit* units, int n_units, int H, int rows, int D, int dtype,
                                          rtk_stream_t stream) {
    RTK_CHECK_ARG(units && n_units >= 1, "rtk_pivotkv_commit_batched: no units");
    RTK_CHECK_ARG(H >= 1 && rows >= 0 && D >= 1, "rtk_pivotkv_commit_batched: bad shape");
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_F16, "rtk_pivotkv_commit_batched: unsupported dtype %d", dtype);
    if (rows == 0) return RTK_OK;
    const size_t es = dtype != RTK_F32 ? 2 : 4;
    const size_t blk = (size_t)rows * D * es;
    for (int i = 0; i < n_units; ++i) {
        const rtk_copy_unit& u = units[i];
        RTK_CHECK_ARG(u.src && u.dst, "rtk_pivotkv_commit_batched: unit %d: NULL pointer", i);
        if (blk % 16 || u.src_stride_h_bytes % 16 || u.dst_stride_h_bytes % 16 || (((uintptr_t)u.src | (uintptr_t)u.dst) & 15)) {
            set_error("rtk_pivotkv_commit_batched: unit %d: blocks must be 16-byte aligned", i);
            return RTK_EUNSUPPORTED;
        }
    }
    const size_t total = (size_t)H * (blk / 16);
    const unsigned gx = (unsigned)std::min<size_t>((total + 4 * 256 - 1) / (4 * 256), 1024);
    hipStream_t st = (hipStream_t)stream;
    for (int b = 0; b < n_units; b += RTK_COPY_MAX_UNITS) {
        const int n = std::min(RTK_COPY_MAX_UNITS, n_units - b);
        CopyUnits cu;
        for (int i = 0; i < n; ++i) cu.u[i] = units[b + i];
        for (int i = n; i < RTK_COPY_MAX_UNITS; ++i) cu.u[i] = units[b];
        RTK_LAUNCH(KID_COMMITB, commit_batched_kernel, dim3(gx, n), dim3(256), 0, st, cu, H, blk);
        RTK_LAUNCH_CHECK("commit_batched_kernel");
    }
    return RTK_OK;
}

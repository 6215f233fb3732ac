// score_refround.cuh — PivotKV scoring in the REFERENCE's 16-bit semantics (dtype codes RTK_BF16_REFROUND and, template
// flag F16, RTK_F16_REFROUND: the same chain with every "-> bf16" below read as "-> fp16"), included by
// pivotkv_score.hip.  longvideo_cache.py:264-270 run on a bf16 model rounds
//     matmul(q, k^T)                 -> bf16      (fp32 accumulation, one rounding)
//     / math.sqrt(D)                 -> bf16
//     softmax(dim=-1, dtype=float32) -> fp32 softmax of those bf16 logits
//     .to(bf16)                      -> probabilities rounded to bf16
//     [0].sum(1)                     -> per HEAD column sums (fp32 accumulation) rounded to bf16
//     .reshape(Hkv, G, L).mean(1)    -> bf16
//     .mean(0)                       -> bf16
// The production kernels (exact bf16 products, fp32 everywhere else) are more accurate than that; this opt-in form
// reproduces the reference's heavily quantised scores (~100-250 distinct values per chunk) so that a user who needs
// the reference's bf16 behaviour bit for bit can have it.  Same two-pass decomposition and LDS-DMA staging as
// score_pass1_dma_kernel / score_pass2_dma_kernel (two 32-row register blocks per wave); pass 1 leaves the base-2
// log-sum-exp lse2_i of each row's bf16 logits, pass 2 forms p = bf16(exp2(l * log2(e) - lse2_i)) - the reference's
// fp32 softmax of the bf16 logits up to the fp32 rounding of the exponent - and sums the bf16 probabilities per head.
// What cannot be reproduced bit for bit is the fp32 summation order inside ATen's bf16 gemm and sum kernels (not part
// of their contract): measured against the reference on CPU, <= 1-2 scores per 6272 differ, by one bf16 ulp
// (tests/test_hip_parity.py::test_pivotkv_reference_rounding_matches_reference_bf16).
#pragma once

namespace rtk {

constexpr float LOG2E_F = 1.4426950408889634f;

// two fp32 matmul results -> the reference's bf16 logits: bf16(bf16(acc) / sqrt(D)).  DIV 1: the division is a
// multiplication by fl32(1/sqrt(D)), used only after the host has verified over all 65536 bf16 inputs that it rounds
// identically (bf16_rcp_is_exact); DIV 2: IEEE division.
template <int DIV, bool F16>
__device__ __forceinline__ void ref_logits2(float a0, float a1, float sqrt_d, float rcp_sd, float& l0, float& l1) {
    using Hh = H16<F16 ? RTK_F16 : RTK_BF16>;   // (fp16: an overflowing product becomes inf, as the reference's fp16 matmul does)
    const uint32_t p = Hh::pack2(a0, a1);
    uint32_t o;
    if constexpr (DIV == 1) o = Hh::pack2(Hh::lo(p) * rcp_sd, Hh::hi(p) * rcp_sd);
    else o = Hh::pack2(__fdiv_rn(Hh::lo(p), sqrt_d), __fdiv_rn(Hh::hi(p), sqrt_d));
    l0 = Hh::lo(o);
    l1 = Hh::hi(o);
}

// Row statistic of pass 1 in the reference's semantics, as ONE number per row: lse2_i = log2 sum_j exp(l_ij) over the
// bf16 logits l (natural exp, base-2 logarithm) - the reference's fp32 softmax is p_ij = exp(l_ij - m_i) / S_i =
// exp2(l_ij * log2(e) - lse2_i) whatever offset m_i it subtracts; what differs between the two forms is the fp32
// rounding of the exponent's argument (1e-7 relative on p, far below the bf16 rounding p gets next).
//   RowStatRefRaw  plain sums of exp2(l * log2 e), checked once at the end of the row: a sum outside fp32's comfortable
//                  range is published as NaN and the fix-up launch recomputes that row tile with RowStatRef
//                  (same scheme as the exact modes' RowStatRX: no max, no compare, no branch per block)
//   RowStatRef     online max / sum of exp(l - max): the robust form
struct RowStatRefRaw {
    float sum;
    __device__ __forceinline__ void init() { sum = 0.f; }
    template <int DIV, bool RAGGED, bool F16>
    __device__ __forceinline__ void update(const f32x16& a, int j0, int j_end, int hf, float sqrt_d, float rcp_sd) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            float l0, l1;
            ref_logits2<DIV, F16>(a[r], a[r + 1], sqrt_d, rcp_sd, l0, l1);
            float e0 = __builtin_amdgcn_exp2f(l0 * LOG2E_F), e1 = __builtin_amdgcn_exp2f(l1 * LOG2E_F);
            if (RAGGED) {
                if (j0 + acc_row(r, hf) >= j_end) e0 = 0.f;
                if (j0 + acc_row(r + 1, hf) >= j_end) e1 = 0.f;
            }
            sum += e0;
            sum += e1;
        }
    }
    __device__ __forceinline__ float finish() const {
        const float s = sum + __shfl_xor(sum, 32, WAVE);
        // inf / NaN are sticky in a sum of non-negative terms; a tiny sum means the terms that matter were flushed
        return (s < 3.0e38f && s > 8.7e-19f) ? __builtin_amdgcn_logf(s) : __builtin_nanf("");
    }
};

struct RowStatRef {  // online max / sum of exp(l - max) of one query row over the keys this lane sees (natural exp)
    float m, sum;
    __device__ __forceinline__ void init() { m = -INFINITY; sum = 0.f; }
    template <int DIV, bool RAGGED, bool F16>
    __device__ __forceinline__ void update(const f32x16& a, int j0, int j_end, int hf, float sqrt_d, float rcp_sd) {
        float l[16];
#pragma unroll
        for (int r = 0; r < 16; r += 2) ref_logits2<DIV, F16>(a[r], a[r + 1], sqrt_d, rcp_sd, l[r], l[r + 1]);
        if (RAGGED) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (j0 + acc_row(r, hf) >= j_end) l[r] = -INFINITY;
        }
        float mn = m;
#pragma unroll
        for (int r = 0; r < 16; ++r) mn = fmaxf(mn, l[r]);
        if (RAGGED && mn == -INFINITY) return;
        float add = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) add += __builtin_amdgcn_exp2f((l[r] - mn) * LOG2E_F);
        sum = sum * __builtin_amdgcn_exp2f((m - mn) * LOG2E_F) + add;
        m = mn;
    }
    __device__ __forceinline__ float finish() const {
        const float m2 = __shfl_xor(m, 32, WAVE), s2 = __shfl_xor(sum, 32, WAVE);
        const float mm = fmaxf(m, m2);
        if (mm == -INFINITY) return -INFINITY;
        const float s = sum * __builtin_amdgcn_exp2f((m - mm) * LOG2E_F) + s2 * __builtin_amdgcn_exp2f((m2 - mm) * LOG2E_F);
        return mm * LOG2E_F + __builtin_amdgcn_logf(s);
    }
};

// pass 1 of one workgroup: NB x 32 query rows of head h per wave starting at i_base + wid * 32 * NB, key split ks ->
// lse_part[ks][h][i] (base-2 log-sum-exp of the row's bf16 logits over the split).  ROBUST: RowStatRef, else RowStatRefRaw.
// Same LDS-DMA staging as score_pass1_dma_body; every A fragment read from LDS feeds NB MFMAs.
template <int DIV, bool ROBUST, int NB, bool F16>
__device__ __forceinline__ void score_pass1_ref_body(const char* __restrict__ q, const char* __restrict__ k, int Hq, int Hkv,
                                                     int L, int keys_per_split, float* __restrict__ lse_part, int i_base,
                                                     int h, int ks, float sqrt_d, float rcp_sd) {
    constexpr int DT = RTK_BF16;
    using M = MM<DT>;
    using T = Tile<DT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), hf = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int G = Hq / Hkv;
    const int g = h / G;
    const int i0 = i_base + wid * (32 * NB);
    const bool live = __builtin_amdgcn_readfirstlane(i0) < L;   // a wave past L keeps its DMA pieces and barriers only
    const int jb = ks * keys_per_split, je = min(L, jb + keys_per_split);
    const int nkeys = je - jb;
    const int nfull = nkeys / TILE_ROWS;
    const int ntiles = (nkeys + TILE_ROWS - 1) / TILE_ROWS;
    u32x4 qf[NB][M::NREG];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) load_reg_frag<DT>(q + (size_t)h * L * HD * M::ESIZE, i0 + 32 * nb, L, lane, qf[nb]);
    int frag_off[M::NREG];
    {
        const int row = lane & 31;
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) frag_off[r] = row * T::ROWB + ((M::chunk_of(r, hf) ^ (row & 15)) * 16);
    }
    using Stat = std::conditional_t<ROBUST, RowStatRef, RowStatRefRaw>;
    Stat rs[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) rs[nb].init();
    const int drow = 4 * wid + (lane >> 4);
    const int dvoff = drow * T::ROWB + (((lane & 15) ^ (drow & 15)) * 16);
    const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(k + (size_t)g * L * HD * M::ESIZE), 0, L * HD * M::ESIZE, 0x00020000);
    auto issue = [&](int t, int b) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                krsrc, (void __attribute__((address_space(3)))*)(smem + b * T::BYTES + (4 * u + wid) * 1024), 16, dvoff,
                (jb + t * TILE_ROWS + 16 * u) * T::ROWB, 0, 0);
    };
    issue(0, 0);
    __syncthreads();
    for (int jt = 0; jt < ntiles; ++jt) {
        const int buf = jt & 1;
        const char* cur = smem + buf * T::BYTES;
        if (jt + 1 < ntiles) issue(jt + 1, buf ^ 1);
        if (live) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                f32x16 acc[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x16{0};
#pragma unroll
                for (int r = 0; r < M::NREG; ++r) {
                    const u32x4 a = *(const u32x4*)(cur + blk * 32 * T::ROWB + frag_off[r]);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) mma16<F16>(acc[nb], a, qf[nb][r], acc[nb]);
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if (jt < nfull) rs[nb].template update<DIV, false, F16>(acc[nb], 0, 0, hf, sqrt_d, rcp_sd);
                    else rs[nb].template update<DIV, true, F16>(acc[nb], jt * TILE_ROWS + 32 * blk, nkeys, hf, sqrt_d, rcp_sd);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float out = rs[nb].finish();
        const int i = i0 + 32 * nb + (lane & 31);
        if (hf == 0 && i < L) lse_part[((size_t)ks * Hq + h) * L + i] = out;
    }
}

constexpr int REF_NB = RTK_REF_P1_NB;    // 32-row register blocks per wave, pass 1 of the reference-rounding kernels
constexpr int REF_NB2 = RTK_REF_P2_NB;   // 32-key register blocks per wave, pass 2

// blockIdx.x -> (row tile bx, head h, key split ks), blockIdx.y = unit of a batched launch
template <int DIV, bool F16>
__global__ __launch_bounds__(SC_BLOCK, (REF_NB == 1 ? 4 : 3)) void score_pass1_ref_kernel(const char* __restrict__ q, const char* __restrict__ k,
                                                                      int Hq, int Hkv, int L, int keys_per_split,
                                                                      int row_tiles, int xcd_remap,
                                                                      float* __restrict__ lse_part, size_t q_unit_bytes,
                                                                      size_t k_unit_bytes, size_t lse_unit_floats,
                                                                      float sqrt_d, float rcp_sd) {
    q += blockIdx.y * q_unit_bytes;
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
    const int i_base = bx * (REG_ROWS * REF_NB);
    if (REF_NB == 2 && L - i_base <= REG_ROWS)   // a last tile that is at most half full: 32 rows per wave
        score_pass1_ref_body<DIV, false, 1, F16>(q, k, Hq, Hkv, L, keys_per_split, lse_part, i_base, h, ks, sqrt_d, rcp_sd);
    else
        score_pass1_ref_body<DIV, false, REF_NB, F16>(q, k, Hq, Hkv, L, keys_per_split, lse_part, i_base, h, ks, sqrt_d, rcp_sd);
}

// fix-up launch of the reference-rounding pass 1 (see score_pass1_fixup_kernel): the row tiles whose plain sums left
// fp32's range (NaN) are recomputed with the online-max form; normally none.  Tiles are numbered
// ((ks * Hq + h) * row_tiles + bx).
template <int DIV, bool F16>
__global__ __launch_bounds__(SC_BLOCK, 2) void score_pass1_ref_fixup_kernel(
    const char* __restrict__ q, const char* __restrict__ k, int Hq, int Hkv, int L, int keys_per_split, int row_tiles,
    int n_tiles, float* __restrict__ lse_part, size_t q_unit_bytes, size_t k_unit_bytes, size_t lse_unit_floats,
    float sqrt_d, float rcp_sd) {
    q += blockIdx.y * q_unit_bytes;
    k += blockIdx.y * k_unit_bytes;
    lse_part += blockIdx.y * lse_unit_floats;
    const int t0 = blockIdx.x * FIX_TILES;
    __shared__ unsigned nan_tiles;
    if (threadIdx.x == 0) nan_tiles = 0;
    const unsigned mine = scan_nan_tiles<REG_ROWS * REF_NB>(lse_part, t0, n_tiles, row_tiles, L);
    __syncthreads();
    if (mine) atomicOr(&nan_tiles, mine);
    __syncthreads();
    unsigned todo = nan_tiles;
    while (todo) {
        const int u = __builtin_ctz(todo);
        todo &= todo - 1;
        const int t = t0 + u;
        const int bx = t % row_tiles, kh = t / row_tiles;
        const int h = kh % Hq, ks = kh / Hq;
        const int i_base = bx * (REG_ROWS * REF_NB);
        if (REF_NB == 2 && L - i_base <= REG_ROWS)
            score_pass1_ref_body<DIV, true, 1, F16>(q, k, Hq, Hkv, L, keys_per_split, lse_part, i_base, h, ks, sqrt_d, rcp_sd);
        else
            score_pass1_ref_body<DIV, true, REF_NB, F16>(q, k, Hq, Hkv, L, keys_per_split, lse_part, i_base, h, ks, sqrt_d, rcp_sd);
        __syncthreads();
    }
}

// pass 2 of one workgroup: NB x 32 keys per wave starting at position j_base + wid * 32 * NB of the live-key list, the
// query rows of split rs of the G heads of KV group g:
//   partial[h][rs][j] = sum_{i in split} bf16(exp2(l_ij * log2(e) - lse2_i))           per HEAD
// The probabilities are rounded to bf16 in pairs (v_cvt_pk_bf16_f32) and summed straight from the packed pair
// (v_dot2_f32_bf16 against (1, 1): fp32 accumulation of exact bf16 values, no unpacking).
template <int DIV, int NB, bool F16>
__device__ __forceinline__ void score_pass2_ref_body(const char* __restrict__ q, const char* __restrict__ k,
                                                     const float* __restrict__ lse, int Hq, int Hkv, int L,
                                                     int rows_per_split, int RS, float* __restrict__ partial, int j_base,
                                                     int g, int rs, const int* __restrict__ kidx, int Lk, float sqrt_d,
                                                     float rcp_sd) {
    constexpr int DT = RTK_BF16;
    using M = MM<DT>;
    using T = Tile<DT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lse_s = (float*)(smem + 2 * T::BYTES);        // [2][TILE_ROWS]
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
    int jcol[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int jp = j0 + 32 * nb + (lane & 31);                  // position in the live-key list
        jcol[nb] = jp < Lk ? (kidx ? kidx[jp] : jp) : -1;            // its token index (k~ row, output column)
        const u32x4* p = (const u32x4*)(kg + (size_t)max(jcol[nb], 0) * HD * M::ESIZE);
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) kf[nb][r] = jcol[nb] >= 0 ? p[M::chunk_of(r, hf)] : u32x4{0, 0, 0, 0};
    }
    const int drow = 4 * wid + (lane >> 4);
    const int dvoff = drow * T::ROWB + (((lane & 15) ^ (drow & 15)) * 16);
    const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)q, 0, Hq * L * HD * M::ESIZE, 0x00020000);
    const int last_row = Hq * L - 1;
    int nt = 0, nrow0 = (g * G) * L + ib;
    float pl = 0.f;
    // DMA of the cursor tile into buffer b + this thread's row normaliser; rows past the split end get lse2 = +inf
    // (exp2(-inf) = 0)
    auto issue = [&](int b) {
        const int row_base = nrow0 + nt * TILE_ROWS;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                qrsrc, (void __attribute__((address_space(3)))*)(smem + b * T::BYTES + (4 * u + wid) * 1024), 16, dvoff,
                (row_base + 16 * u) * T::ROWB, 0, 0);
        const int r = nt * TILE_ROWS + (tid & (TILE_ROWS - 1));
        pl = (r < nrows) ? lse[min(nrow0 + r, last_row)] : INFINITY;
        const bool wrap = (nt + 1 == tiles_per_head);
        nt = wrap ? 0 : nt + 1;
        nrow0 += wrap ? L : 0;
    };
    issue(0);
    if (tid < TILE_ROWS) lse_s[tid] = pl;
    __syncthreads();
    float col[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) col[nb] = 0.f;
    int tcount = 0, hh = 0;
    for (int it = 0; it < ntiles; ++it) {
        const int buf = it & 1;
        const char* cur = smem + buf * T::BYTES;
        if (it + 1 < ntiles) issue(buf ^ 1);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f32x16 acc[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x16{0};
#pragma unroll
            for (int r = 0; r < M::NREG; ++r) {   // one fragment at a time: each feeds NB MFMAs and is dead afterwards
                const u32x4 a = *(const u32x4*)(cur + blk * 32 * T::ROWB + frag_off[r]);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) mma16<F16>(acc[nb], a, kf[nb][r], acc[nb]);
            }
            float ls[16];
            load_ls(ls, lse_s + buf * TILE_ROWS, blk, hf);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    float l0, l1;
                    ref_logits2<DIV, F16>(acc[nb][r], acc[nb][r + 1], sqrt_d, rcp_sd, l0, l1);
                    const float p0 = __builtin_amdgcn_exp2f(fmaf(l0, LOG2E_F, -ls[r]));
                    const float p1 = __builtin_amdgcn_exp2f(fmaf(l1, LOG2E_F, -ls[r + 1]));
                    using Hh = H16<F16 ? RTK_F16 : RTK_BF16>;
                    col[nb] = Hh::dot2(Hh::pack2(p0, p1), Hh::ONE2, col[nb]);   // .to(bf16 / fp16), .sum
                }
            }
        }
        if (it + 1 < ntiles && tid < TILE_ROWS) lse_s[(buf ^ 1) * TILE_ROWS + tid] = pl;
        if (++tcount == tiles_per_head) {   // this head's rows are done: its column sums leave on their own
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float c = col[nb] + __shfl_xor(col[nb], 32, WAVE);
                if (hf == 0 && jcol[nb] >= 0) partial[((size_t)(g * G + hh) * RS + rs) * L + jcol[nb]] = c;
                col[nb] = 0.f;
            }
            tcount = 0;
            ++hh;
        }
        __syncthreads();
    }
}

template <int DIV, bool F16>
__global__ __launch_bounds__(SC_BLOCK, (REF_NB2 == 1 ? 4 : 3)) void score_pass2_ref_kernel(
    const char* __restrict__ q, const char* __restrict__ k, const float* __restrict__ lse, int Hq, int Hkv, int L,
    int rows_per_split, int col_tiles, int RS, int xcd_remap, float* __restrict__ partial, size_t q_unit_bytes,
    size_t k_unit_bytes, size_t lse_unit_floats, size_t part_unit_floats, float sqrt_d, float rcp_sd,
    const int* __restrict__ key_index = nullptr) {
    // live keys of the unit (key_compact_kernel, see score_pass2_dma_kernel)
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
    q += blockIdx.y * q_unit_bytes;
    k += blockIdx.y * k_unit_bytes;
    lse += blockIdx.y * lse_unit_floats;
    partial += blockIdx.y * part_unit_floats;
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
    const int j_base = bx * (REG_ROWS * REF_NB2);
    if (j_base >= Lk) return;   // uniform per workgroup
    if (REF_NB2 == 2 && Lk - j_base <= REG_ROWS)
        score_pass2_ref_body<DIV, 1, F16>(q, k, lse, Hq, Hkv, L, rows_per_split, RS, partial, j_base, g, rs, kidx, Lk, sqrt_d, rcp_sd);
    else
        score_pass2_ref_body<DIV, REF_NB2, F16>(q, k, lse, Hq, Hkv, L, rows_per_split, RS, partial, j_base, g, rs, kidx, Lk, sqrt_d,
                                           rcp_sd);
}

template <bool F16>
__global__ __launch_bounds__(256) void score_finalize_ref_kernel(const float* __restrict__ partial, int Hkv, int RS, int G,
                                                                 int L, float* __restrict__ score) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < L) score[j] = finalize_ref_column<F16>(partial, Hkv, RS, G, L, j);
}

}  // namespace rtk

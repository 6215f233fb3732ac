// score_refround.cuh — PivotKV scoring in the REFERENCE's bf16 semantics (dtype code RTK_BF16_REFROUND), included by
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
// score_pass1_dma_kernel / score_pass2_dma_kernel; the softmax is kept in the reference's form: row max m_i and row
// sum S_i of exp(l - m_i) from pass 1, p = bf16(exp(l - m_i) * (1 / S_i)) in pass 2, column sums per head.
// What cannot be reproduced bit for bit is the fp32 summation order inside ATen's bf16 gemm and sum kernels (not part
// of their contract): measured against the reference on CPU, <= 1-2 scores per 6272 differ, by one bf16 ulp
// (tests/test_hip_parity.py::test_pivotkv_reference_rounding_matches_reference_bf16).
#pragma once

namespace rtk {

constexpr float LOG2E_F = 1.4426950408889634f;

// two fp32 matmul results -> the reference's bf16 logits: bf16(bf16(acc) / sqrt(D)).  DIV 1: the division is a
// multiplication by fl32(1/sqrt(D)), used only after the host has verified over all 65536 bf16 inputs that it rounds
// identically (bf16_rcp_is_exact); DIV 2: IEEE division.
template <int DIV>
__device__ __forceinline__ void ref_logits2(float a0, float a1, float sqrt_d, float rcp_sd, float& l0, float& l1) {
    const uint32_t p = pack2_bf16(a0, a1);
    uint32_t o;
    if constexpr (DIV == 1) o = pack2_bf16(bf_lo(p) * rcp_sd, bf_hi(p) * rcp_sd);
    else o = pack2_bf16(__fdiv_rn(bf_lo(p), sqrt_d), __fdiv_rn(bf_hi(p), sqrt_d));
    l0 = bf_lo(o);
    l1 = bf_hi(o);
}

struct RowStatRef {  // online max / sum of exp(l - max) of one query row over the keys this lane sees (natural exp)
    float m, sum;
    __device__ __forceinline__ void init() { m = -INFINITY; sum = 0.f; }
    template <int DIV, bool RAGGED>
    __device__ __forceinline__ void update(const f32x16& a, int j0, int j_end, int hf, float sqrt_d, float rcp_sd) {
        float l[16];
#pragma unroll
        for (int r = 0; r < 16; r += 2) ref_logits2<DIV>(a[r], a[r + 1], sqrt_d, rcp_sd, l[r], l[r + 1]);
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
    __device__ __forceinline__ void finish(float& m_out, float& s_out) const {
        const float m2 = __shfl_xor(m, 32, WAVE), s2 = __shfl_xor(sum, 32, WAVE);
        const float mm = fmaxf(m, m2);
        m_out = mm;
        s_out = (mm == -INFINITY) ? 0.f
                                  : sum * __builtin_amdgcn_exp2f((m - mm) * LOG2E_F) + s2 * __builtin_amdgcn_exp2f((m2 - mm) * LOG2E_F);
    }
};

// pass 1: stat[0][ks][h][i] = max_j l_ij, stat[1][ks][h][i] = sum_j exp(l_ij - max) over key split ks
template <int DIV>
__global__ __launch_bounds__(SC_BLOCK, 3) void score_pass1_ref_kernel(const char* __restrict__ q, const char* __restrict__ k,
                                                                      int Hq, int Hkv, int L, int keys_per_split,
                                                                      int row_tiles, int xcd_remap, int KS,
                                                                      float* __restrict__ stat, size_t q_unit_bytes,
                                                                      size_t k_unit_bytes, size_t stat_unit_floats,
                                                                      float sqrt_d, float rcp_sd) {
    constexpr int DT = RTK_BF16;
    using M = MM<DT>;
    using T = Tile<DT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    q += blockIdx.y * q_unit_bytes;
    k += blockIdx.y * k_unit_bytes;
    stat += blockIdx.y * stat_unit_floats;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), hf = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid / WAVE);
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
    const int g = h / G;
    const int i0 = bx * REG_ROWS + wid * 32;
    const int jb = ks * keys_per_split, je = min(L, jb + keys_per_split);
    const int nkeys = je - jb;
    const int nfull = nkeys / TILE_ROWS;
    const int ntiles = (nkeys + TILE_ROWS - 1) / TILE_ROWS;
    u32x4 qf[M::NREG];
    load_reg_frag<DT>(q + (size_t)h * L * HD * M::ESIZE, i0, L, lane, qf);
    int frag_off[M::NREG];
    {
        const int row = lane & 31;
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) frag_off[r] = row * T::ROWB + ((M::chunk_of(r, hf) ^ (row & 15)) * 16);
    }
    RowStatRef rs;
    rs.init();
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
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f32x16 acc = f32x16{0};
#pragma unroll
            for (int r = 0; r < M::NREG; ++r) M::mma(acc, *(const u32x4*)(cur + blk * 32 * T::ROWB + frag_off[r]), qf[r]);
            if (jt < nfull) rs.update<DIV, false>(acc, 0, 0, hf, sqrt_d, rcp_sd);
            else rs.update<DIV, true>(acc, jt * TILE_ROWS + 32 * blk, nkeys, hf, sqrt_d, rcp_sd);
        }
        __syncthreads();
    }
    float mo, so;
    rs.finish(mo, so);
    const int i = i0 + (lane & 31);
    if (hf == 0 && i < L) {
        const size_t n = (size_t)Hq * L;
        stat[(size_t)ks * n + (size_t)h * L + i] = mo;
        stat[((size_t)KS + ks) * n + (size_t)h * L + i] = so;
    }
}

// row statistics over all key splits: stat[0][0][.] = m, stat[1][0][.] = 1 / S   (S in fp32, like the reference's
// softmax kernel; its exact summation order is ATen's business)
__global__ __launch_bounds__(256) void stat_combine_ref_kernel(float* __restrict__ stat, size_t n, int KS, int KS_alloc,
                                                               size_t unit_floats) {
    stat += blockIdx.y * unit_floats;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    float mv[8], sv[8];
    float mx = -INFINITY;
    for (int s = 0; s < KS; ++s) {
        mv[s] = stat[(size_t)s * n + idx];
        sv[s] = stat[((size_t)KS_alloc + s) * n + idx];
        mx = fmaxf(mx, mv[s]);
    }
    float tot = 0.f;
    for (int s = 0; s < KS; ++s) tot += sv[s] * __builtin_amdgcn_exp2f((mv[s] - mx) * LOG2E_F);
    stat[idx] = mx;
    stat[(size_t)KS_alloc * n + idx] = __fdiv_rn(1.0f, tot);
}

// pass 2: partial[h][rs][j] = sum_{i in row split rs} bf16(exp(l_ij - m_i) * (1 / S_i))   per HEAD
template <int DIV>
__global__ __launch_bounds__(SC_BLOCK, 3) void score_pass2_ref_kernel(
    const char* __restrict__ q, const char* __restrict__ k, const float* __restrict__ stat, int Hq, int Hkv, int L,
    int rows_per_split, int col_tiles, int RS, int xcd_remap, int KS_alloc, float* __restrict__ partial,
    size_t q_unit_bytes, size_t k_unit_bytes, size_t stat_unit_floats, size_t part_unit_floats, float sqrt_d,
    float rcp_sd, const int* __restrict__ key_index = nullptr) {
    constexpr int DT = RTK_BF16;
    using M = MM<DT>;
    using T = Tile<DT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* m_s = (float*)(smem + 2 * T::BYTES);        // [2][TILE_ROWS]
    // live keys of the unit (key_compact_kernel, see score_pass2_dma_kernel): positions of the compacted list name the
    // k~ row a lane loads and the column it writes; key tiles past the list have nothing to do
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
    float* r_s = m_s + 2 * TILE_ROWS;                   // [2][TILE_ROWS]
    q += blockIdx.y * q_unit_bytes;
    k += blockIdx.y * k_unit_bytes;
    stat += blockIdx.y * stat_unit_floats;
    partial += blockIdx.y * part_unit_floats;
    const float* mrow = stat;
    const float* rrow = stat + (size_t)KS_alloc * Hq * L;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), hf = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int G = Hq / Hkv;
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
    const int j0 = bx * REG_ROWS + wid * 32;
    if (bx * REG_ROWS >= Lk) return;   // uniform per workgroup
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
    u32x4 kf[M::NREG];
    const int jp = j0 + (lane & 31);                       // position in the live-key list
    const int jcol = jp < Lk ? (kidx ? kidx[jp] : jp) : -1; // its token index (k~ row, output column)
    {
        const u32x4* p = (const u32x4*)(kg + (size_t)max(jcol, 0) * HD * M::ESIZE);
#pragma unroll
        for (int r = 0; r < M::NREG; ++r) kf[r] = jcol >= 0 ? p[M::chunk_of(r, hf)] : u32x4{0, 0, 0, 0};
    }
    const int drow = 4 * wid + (lane >> 4);
    const int dvoff = drow * T::ROWB + (((lane & 15) ^ (drow & 15)) * 16);
    const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)q, 0, Hq * L * HD * M::ESIZE, 0x00020000);
    const int last_row = Hq * L - 1;
    int nt = 0, nrow0 = (g * G) * L + ib;
    float pm = 0.f, pr = 0.f;
    // DMA of the cursor tile into buffer b + this thread's row statistics; rows past the split end get m = +inf
    // (exp(l - inf) = 0) and 1/S = 0
    auto issue = [&](int b) {
        const int row_base = nrow0 + nt * TILE_ROWS;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                qrsrc, (void __attribute__((address_space(3)))*)(smem + b * T::BYTES + (4 * u + wid) * 1024), 16, dvoff,
                (row_base + 16 * u) * T::ROWB, 0, 0);
        const int r = nt * TILE_ROWS + (tid & (TILE_ROWS - 1));
        const int gi = min(nrow0 + r, last_row);
        pm = (r < nrows) ? mrow[gi] : INFINITY;
        pr = (r < nrows) ? rrow[gi] : 0.f;
        const bool wrap = (nt + 1 == tiles_per_head);
        nt = wrap ? 0 : nt + 1;
        nrow0 += wrap ? L : 0;
    };
    issue(0);
    if (tid < TILE_ROWS) { m_s[tid] = pm; r_s[tid] = pr; }
    __syncthreads();
    float col = 0.f;
    int tcount = 0, hh = 0;
    for (int it = 0; it < ntiles; ++it) {
        const int buf = it & 1;
        const char* cur = smem + buf * T::BYTES;
        if (it + 1 < ntiles) issue(buf ^ 1);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f32x16 acc = f32x16{0};
#pragma unroll
            for (int r = 0; r < M::NREG; ++r) M::mma(acc, *(const u32x4*)(cur + blk * 32 * T::ROWB + frag_off[r]), kf[r]);
            float ms[16], rr[16];
            load_ls(ms, m_s + buf * TILE_ROWS, blk, hf);
            load_ls(rr, r_s + buf * TILE_ROWS, blk, hf);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float l0, l1;
                ref_logits2<DIV>(acc[r], acc[r + 1], sqrt_d, rcp_sd, l0, l1);
                const float p0 = __builtin_amdgcn_exp2f((l0 - ms[r]) * LOG2E_F) * rr[r];
                const float p1 = __builtin_amdgcn_exp2f((l1 - ms[r + 1]) * LOG2E_F) * rr[r + 1];
                const uint32_t pb = pack2_bf16(p0, p1);       // attn_weights.to(bf16)
                col += bf_lo(pb);
                col += bf_hi(pb);
            }
        }
        if (it + 1 < ntiles && tid < TILE_ROWS) { m_s[(buf ^ 1) * TILE_ROWS + tid] = pm; r_s[(buf ^ 1) * TILE_ROWS + tid] = pr; }
        if (++tcount == tiles_per_head) {   // this head's rows are done: its column sums leave on their own
            const float c = col + __shfl_xor(col, 32, WAVE);
            if (hf == 0 && jcol >= 0) partial[((size_t)(g * G + hh) * RS + rs) * L + jcol] = c;
            col = 0.f;
            tcount = 0;
            ++hh;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void score_finalize_ref_kernel(const float* __restrict__ partial, int Hkv, int RS, int G,
                                                                 int L, float* __restrict__ score) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < L) score[j] = finalize_ref_column(partial, Hkv, RS, G, L, j);
}

}  // namespace rtk

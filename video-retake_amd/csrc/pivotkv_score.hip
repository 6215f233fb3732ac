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
#include "common.cuh"

namespace rtk {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int HD = 128;       // head_dim of the MFMA path
constexpr int TILE_ROWS = 64; // rows of the streamed LDS tile
constexpr int REG_ROWS = 128; // rows held in registers per workgroup (32 per wave)
constexpr int SC_BLOCK = 256;

// ------------------------------------------------------------------------------------------------
// un-rotate + pack:  x [H,L,D] strided -> out [H,L,D] contiguous (same dtype)
//   cos == NULL: plain copy;  else ((x*cos) - (rotate_half(x)*sin)) / a^2  with one rounding per
//   torch op (bf16: every intermediate is a bf16 tensor; fp32: no fma contraction).
// ------------------------------------------------------------------------------------------------
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
        if (DT == RTK_BF16) {
            x1 = bf2f(((const uint16_t*)xv)[src + d]);
            x2 = bf2f(((const uint16_t*)xv)[src + d + h2]);
        } else {
            x1 = ((const float*)xv)[src + d];
            x2 = ((const float*)xv)[src + d + h2];
        }
        float o1 = x1, o2 = x2;
        if (cosv) {
            const float c1 = cosv[(size_t)l * D + d], s1 = sinv[(size_t)l * D + d];
            const float c2 = cosv[(size_t)l * D + d + h2], s2 = sinv[(size_t)l * D + d + h2];
            // rotate_half(x)[d] = -x2, rotate_half(x)[d+h2] = x1   (longvideo_cache.py:28-32)
            if (DT == RTK_BF16) {
                o1 = rbf(rbf(rbf(x1 * c1) - rbf(-x2 * s1)) / a2);
                o2 = rbf(rbf(rbf(x2 * c2) - rbf(x1 * s2)) / a2);
            } else {
                o1 = __fdiv_rn(__fsub_rn(__fmul_rn(x1, c1), __fmul_rn(-x2, s1)), a2);
                o2 = __fdiv_rn(__fsub_rn(__fmul_rn(x2, c2), __fmul_rn(x1, s2)), a2);
            }
        }
        if (DT == RTK_BF16) {
            ((uint16_t*)outv)[dst + d] = f2bf(o1);
            ((uint16_t*)outv)[dst + d + h2] = f2bf(o2);
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
    static constexpr int STAGE = (TILE_ROWS * CHUNKS) / SC_BLOCK;  // 4 chunks per thread per tile
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
    static constexpr int STAGE = (TILE_ROWS * CHUNKS) / SC_BLOCK;  // 8
    __device__ static __forceinline__ int chunk_of(int r, int hf) { return 16 * hf + r; }
    __device__ static __forceinline__ void mma(f32x16& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

// accumulator register r of lane (half hf) holds output row  m = (r&3) + 8*(r>>2) + 4*hf
__device__ __forceinline__ int acc_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

// Register fragment: row (lane & 31) of a 32-row block starting at `row0` of a contiguous [rows,128] matrix.
template <int DT>
__device__ __forceinline__ void load_reg_frag(const char* __restrict__ base, int row0, int nrows, int lane,
                                              u32x4* rf) {
    using M = MM<DT>;
    const int row = row0 + (lane & 31), hf = lane >> 5;
    const bool ok = row < nrows;
    const u32x4* p = (const u32x4*)(base + (size_t)row * HD * M::ESIZE);
#pragma unroll
    for (int r = 0; r < M::NREG; ++r) rf[r] = ok ? p[M::chunk_of(r, hf)] : u32x4{0, 0, 0, 0};
}

// Streamed tile: global -> registers (issue early) ...
template <int DT>
__device__ __forceinline__ void stage_load(const char* __restrict__ base, int row0, int row_end, int tid, u32x4* st) {
    using M = MM<DT>;
#pragma unroll
    for (int u = 0; u < M::STAGE; ++u) {
        const int c = tid + SC_BLOCK * u;
        const int row = c / M::CHUNKS, ch = c % M::CHUNKS;
        const int grow = row0 + row;
        st[u] = (grow < row_end) ? ((const u32x4*)(base + (size_t)grow * HD * M::ESIZE))[ch] : u32x4{0, 0, 0, 0};
    }
}
// ... registers -> LDS (write late), 16-byte chunks XOR-swizzled by (row & 15): the 16 lanes of every
// ds_read_b128 lane group address 16 distinct rows (mod 16) => 16 distinct 16-byte bank slots.
template <int DT>
__device__ __forceinline__ void stage_store(char* lds, int tid, const u32x4* st) {
    using M = MM<DT>;
#pragma unroll
    for (int u = 0; u < M::STAGE; ++u) {
        const int c = tid + SC_BLOCK * u;
        const int row = c / M::CHUNKS, ch = c % M::CHUNKS;
        *(u32x4*)(lds + (size_t)row * (M::CHUNKS * 16) + (size_t)((ch ^ (row & 15)) * 16)) = st[u];
    }
}

// acc += A(32 LDS rows starting at blk_row) x B(register fragment)
template <int DT>
__device__ __forceinline__ void block_mma(f32x16& acc, const char* lds, int blk_row, int lane, const u32x4* rf) {
    using M = MM<DT>;
    const int row = blk_row + (lane & 31), hf = lane >> 5;
    const char* rp = lds + (size_t)row * (M::CHUNKS * 16);
    const int sw = row & 15;
#pragma unroll
    for (int r = 0; r < M::NREG; ++r) {
        const u32x4 a = *(const u32x4*)(rp + ((M::chunk_of(r, hf) ^ sw) * 16));
        M::mma(acc, a, rf[r]);
    }
}

// ------------------------------------------------------------------------------------------------
// pass 1: lse[h,i] = log sum_j exp(q_hi . k_gj / sqrt(D))      (natural log for fp32, log2 for bf16)
// grid (ceil(L/128), Hq), 256 threads; wave w keeps query rows i0 + 32w + (lane&31) in registers.
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(SC_BLOCK) void score_pass1_kernel(const char* __restrict__ q, const char* __restrict__ k,
                                                               int Hq, int Hkv, int L, float* __restrict__ lse) {
    using M = MM<DT>;
    constexpr int TILE_BYTES = TILE_ROWS * M::CHUNKS * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE, hf = lane >> 5;
    const int h = blockIdx.y, g = h / (Hq / Hkv);
    const int i0 = blockIdx.x * REG_ROWS + wid * 32;
    const char* qh = q + (size_t)h * L * HD * M::ESIZE;
    const char* kg = k + (size_t)g * L * HD * M::ESIZE;

    u32x4 qf[M::NREG];
    load_reg_frag<DT>(qh, i0, L, lane, qf);

    u32x4 st[M::STAGE];
    const int ntiles = (L + TILE_ROWS - 1) / TILE_ROWS;
    stage_load<DT>(kg, 0, L, tid, st);
    stage_store<DT>(smem, tid, st);
    __syncthreads();

    // bf16: base-2 domain with the 1/sqrt(D) folded into the scale; fp32: the reference's own
    // operation order (logits / sqrt(D), natural exp).
    const float sqrt_d = sqrtf((float)HD);
    const float c2 = 1.4426950408889634f / sqrt_d;
    float m = -INFINITY, sum = 0.f;
    for (int jt = 0; jt < ntiles; ++jt) {
        const char* cur = smem + (size_t)(jt & 1) * TILE_BYTES;
        if (jt + 1 < ntiles) stage_load<DT>(kg, (jt + 1) * TILE_ROWS, L, tid, st);
        f32x16 acc0 = {0}, acc1 = {0};
        block_mma<DT>(acc0, cur, 0, lane, qf);
        block_mma<DT>(acc1, cur, 32, lane, qf);
        float v[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[r] = (DT == RTK_BF16) ? acc0[r] * c2 : __fdiv_rn(acc0[r], sqrt_d);
            v[16 + r] = (DT == RTK_BF16) ? acc1[r] * c2 : __fdiv_rn(acc1[r], sqrt_d);
        }
        const int j0 = jt * TILE_ROWS;
        if (j0 + TILE_ROWS > L) {  // ragged last tile: keys >= L do not exist
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (j0 + acc_row(r, hf) >= L) v[r] = -INFINITY;
                if (j0 + 32 + acc_row(r, hf) >= L) v[16 + r] = -INFINITY;
            }
        }
        float mx = v[0];
#pragma unroll
        for (int r = 1; r < 32; ++r) mx = fmaxf(mx, v[r]);
        const float mn = fmaxf(m, mx);
        if (mn > -INFINITY) {
            float add = 0.f;
            if (DT == RTK_BF16) {
#pragma unroll
                for (int r = 0; r < 32; ++r) add += __builtin_amdgcn_exp2f(v[r] - mn);
                sum = sum * __builtin_amdgcn_exp2f(m - mn) + add;
            } else {
#pragma unroll
                for (int r = 0; r < 32; ++r) add += expf(v[r] - mn);
                sum = sum * expf(m - mn) + add;
            }
            m = mn;
        }
        if (jt + 1 < ntiles) stage_store<DT>(smem + (size_t)((jt + 1) & 1) * TILE_BYTES, tid, st);
        __syncthreads();
    }
    // the two halves of the wave saw disjoint key subsets of the same query row
    const float m2 = __shfl_xor(m, 32, WAVE), s2 = __shfl_xor(sum, 32, WAVE);
    const float mm = fmaxf(m, m2);
    float out;
    if (DT == RTK_BF16) {
        const float tot = sum * __builtin_amdgcn_exp2f(m - mm) + s2 * __builtin_amdgcn_exp2f(m2 - mm);
        out = mm + __builtin_amdgcn_logf(tot);  // v_log_f32 = log2
    } else {
        const float tot = sum * expf(m - mm) + s2 * expf(m2 - mm);
        out = mm + logf(tot);
    }
    const int i = i0 + (lane & 31);
    if (hf == 0 && i < L) lse[(size_t)h * L + i] = out;
}

// ------------------------------------------------------------------------------------------------
// pass 2: partial[g,split,j] = sum_{h in g} sum_{i in split} exp(s_hij - lse[h,i])
// grid (ceil(L/128), Hkv, RS); wave w keeps keys j0 + 32w + (lane&31) in registers.
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(SC_BLOCK) void score_pass2_kernel(const char* __restrict__ q, const char* __restrict__ k,
                                                               const float* __restrict__ lse, int Hq, int Hkv, int L,
                                                               int rows_per_split, float* __restrict__ partial) {
    using M = MM<DT>;
    constexpr int TILE_BYTES = TILE_ROWS * M::CHUNKS * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lse_s = (float*)(smem + 2 * TILE_BYTES);  // [2][TILE_ROWS]
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE, hf = lane >> 5;
    const int g = blockIdx.y, G = Hq / Hkv, rs = blockIdx.z, RS = gridDim.z;
    const int j0 = blockIdx.x * REG_ROWS + wid * 32;
    const char* kg = k + (size_t)g * L * HD * M::ESIZE;
    const int ib = rs * rows_per_split, ie = min(L, ib + rows_per_split);
    const int tiles_per_head = (ie > ib) ? (ie - ib + TILE_ROWS - 1) / TILE_ROWS : 0;
    const int ntiles = tiles_per_head * G;

    u32x4 kf[M::NREG];
    load_reg_frag<DT>(kg, j0, L, lane, kf);

    auto tile_src = [&](int it, const char*& qh, const float*& lh, int& row0) {
        const int hh = it / tiles_per_head, tt = it % tiles_per_head;
        const int h = g * G + hh;
        qh = q + (size_t)h * L * HD * M::ESIZE;
        lh = lse + (size_t)h * L;
        row0 = ib + tt * TILE_ROWS;
    };

    u32x4 st[M::STAGE];
    float lst = 0.f;
    const float sqrt_d = sqrtf((float)HD);
    const float c2 = 1.4426950408889634f / sqrt_d;
    float col = 0.f;
    if (ntiles > 0) {
        const char* qh; const float* lh; int row0;
        tile_src(0, qh, lh, row0);
        stage_load<DT>(qh, row0, ie, tid, st);
        if (tid < TILE_ROWS) lse_s[tid] = (row0 + tid < ie) ? lh[row0 + tid] : INFINITY;
        stage_store<DT>(smem, tid, st);
        __syncthreads();
        for (int it = 0; it < ntiles; ++it) {
            const char* cur = smem + (size_t)(it & 1) * TILE_BYTES;
            const float* lcur = lse_s + (it & 1) * TILE_ROWS;
            if (it + 1 < ntiles) {
                tile_src(it + 1, qh, lh, row0);
                stage_load<DT>(qh, row0, ie, tid, st);
                if (tid < TILE_ROWS) lst = (row0 + tid < ie) ? lh[row0 + tid] : INFINITY;
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                f32x16 acc = {0};
                block_mma<DT>(acc, cur, blk * 32, lane, kf);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 l4 = *(const float4*)(lcur + blk * 32 + 8 * r4 + 4 * hf);
                    const float ls[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float a = acc[4 * r4 + rr];
                        if (DT == RTK_BF16) col += __builtin_amdgcn_exp2f(fmaf(a, c2, -ls[rr]));
                        else col += expf(__fdiv_rn(a, sqrt_d) - ls[rr]);
                    }
                }
            }
            if (it + 1 < ntiles) {
                stage_store<DT>(smem + (size_t)((it + 1) & 1) * TILE_BYTES, tid, st);
                if (tid < TILE_ROWS) lse_s[((it + 1) & 1) * TILE_ROWS + tid] = lst;
            }
            __syncthreads();
        }
    }
    col += __shfl_xor(col, 32, WAVE);
    const int j = j0 + (lane & 31);
    if (hf == 0 && j < L) partial[((size_t)g * RS + rs) * L + j] = col;
}

// ------------------------------------------------------------------------------------------------
// generic fallback (any head_dim; small problems): plain fp32 VALU, same two passes.
// ------------------------------------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ float ldx(const void* p, size_t i) {
    return DT == RTK_BF16 ? bf2f(((const uint16_t*)p)[i]) : ((const float*)p)[i];
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
        for (int i = 0; i < L; ++i) {
            __syncthreads();
            for (int d = tid; d < D; d += blockDim.x) qs[d] = ldx<DT>(q, ((size_t)h * L + i) * D + d);
            __syncthreads();
            if (j < L) {
                float s = 0.f;
                for (int d = 0; d < D; ++d) s = fmaf(qs[d], ldx<DT>(k, ((size_t)g * L + j) * D + d), s);
                col += expf(__fdiv_rn(s, sqrt_d) - lse[(size_t)h * L + i]);
            }
        }
    }
    if (j < L) partial[(size_t)g * L + j] = col;
}

// finalize: score[j] = mean_g( (sum_split partial[g,split,j]) / G )      (longvideo_cache.py:269-270)
__global__ __launch_bounds__(256) void score_finalize_kernel(const float* __restrict__ partial, int Hkv, int RS, int G,
                                                             int L, float* __restrict__ score) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= L) return;
    float tot = 0.f;
    for (int g = 0; g < Hkv; ++g) {
        float gs = 0.f;
        for (int r = 0; r < RS; ++r) gs += partial[((size_t)g * RS + r) * L + j];
        tot += gs / (float)G;
    }
    score[j] = tot / (float)Hkv;
}

static int pick_row_splits(int L, int Hkv) {
    const int jt = (L + REG_ROWS - 1) / REG_ROWS;
    int rs = (1024 + jt * Hkv - 1) / (jt * Hkv);
    const int max_rs = (L + TILE_ROWS - 1) / TILE_ROWS;
    rs = std::max(1, std::min(std::min(rs, 16), max_rs));
    return rs;
}

struct ScoreWs {
    size_t q_off, k_off, lse_off, part_off, total;
    int RS;
};
static ScoreWs score_ws(int Hq, int Hkv, int L, int D, int dtype) {
    const size_t es = dtype == RTK_BF16 ? 2 : 4;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    ScoreWs w;
    w.RS = (D == HD) ? pick_row_splits(L, Hkv) : 1;
    w.q_off = 0;
    w.k_off = al((size_t)Hq * L * D * es);
    w.lse_off = w.k_off + al((size_t)Hkv * L * D * es);
    w.part_off = w.lse_off + al((size_t)Hq * L * 4);
    w.total = w.part_off + al((size_t)Hkv * w.RS * L * 4);
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
                      void* k_unrot, char* ws, const ScoreWs& w, hipStream_t st) {
    char* qt = ws + w.q_off;
    char* kt = k_unrot ? (char*)k_unrot : ws + w.k_off;
    float* lse = (float*)(ws + w.lse_off);
    float* part = (float*)(ws + w.part_off);
    const float a2 = (float)((double)a * (double)a);  // python float ** 2, then an fp32 tensor / scalar
    {
        const size_t nq = (size_t)Hq * L * (D / 2), nk = (size_t)Hkv * L * (D / 2);
        RTK_LAUNCH(KID_UNROT, unrotate_pack_kernel<DT>, dim3((unsigned)std::min<size_t>((nq + 255) / 256, 8192)), dim3(256),
                   0, st, q, qsh, qsl, Hq, L, D, cosv, sinv, a2, (void*)qt);
        RTK_LAUNCH(KID_UNROT, unrotate_pack_kernel<DT>, dim3((unsigned)std::min<size_t>((nk + 255) / 256, 8192)), dim3(256),
                   0, st, k, ksh, ksl, Hkv, L, D, cosv, sinv, a2, (void*)kt);
        RTK_LAUNCH_CHECK("unrotate_pack_kernel");
    }
    const int G = Hq / Hkv;
    if (D == HD) {
        using M = MM<DT>;
        constexpr int TILE_BYTES = TILE_ROWS * M::CHUNKS * 16;
        const int jt = (L + REG_ROWS - 1) / REG_ROWS;
        RTK_LAUNCH(KID_PASS1, score_pass1_kernel<DT>, dim3(jt, Hq), dim3(SC_BLOCK), 2 * TILE_BYTES, st, (const char*)qt,
                           (const char*)kt, Hq, Hkv, L, lse);
        RTK_LAUNCH_CHECK("score_pass1_kernel");
        int rows_per_split = (L + w.RS - 1) / w.RS;
        rows_per_split = ((rows_per_split + TILE_ROWS - 1) / TILE_ROWS) * TILE_ROWS;
        RTK_LAUNCH(KID_PASS2, score_pass2_kernel<DT>, dim3(jt, Hkv, w.RS), dim3(SC_BLOCK),
                           2 * TILE_BYTES + 2 * TILE_ROWS * sizeof(float), st, (const char*)qt, (const char*)kt, lse, Hq,
                           Hkv, L, rows_per_split, part);
        RTK_LAUNCH_CHECK("score_pass2_kernel");
    } else {
        RTK_LAUNCH(KID_PASS1, score_pass1_generic<DT>, dim3(L, Hq), dim3(256), D * sizeof(float), st, (const void*)qt,
                           (const void*)kt, Hq, Hkv, L, D, lse);
        RTK_LAUNCH(KID_PASS2, score_pass2_generic<DT>, dim3((L + 255) / 256, Hkv), dim3(256), D * sizeof(float), st,
                           (const void*)qt, (const void*)kt, lse, Hq, Hkv, L, D, part);
        RTK_LAUNCH_CHECK("score_generic");
    }
    RTK_LAUNCH(KID_FINALIZE, score_finalize_kernel, dim3((L + 255) / 256), dim3(256), 0, st, part, Hkv, w.RS, G, L, score);
    RTK_LAUNCH_CHECK("score_finalize_kernel");
    return RTK_OK;
}

extern "C" int rtk_pivotkv_score(const void* q, int64_t q_stride_h, int64_t q_stride_l, const void* k,
                                 int64_t k_stride_h, int64_t k_stride_l, int Hq, int Hkv, int L, int D, int dtype,
                                 const float* cosv, const float* sinv, float attention_scaling, float* score,
                                 void* k_unrot, void* workspace, size_t workspace_bytes, rtk_stream_t stream) {
    RTK_CHECK_ARG(q && k && score && workspace, "rtk_pivotkv_score: NULL pointer");
    RTK_CHECK_ARG(Hq >= 1 && Hkv >= 1 && Hq % Hkv == 0, "rtk_pivotkv_score: Hq=%d must be a multiple of Hkv=%d", Hq, Hkv);
    RTK_CHECK_ARG(L >= 1 && D >= 2 && D % 2 == 0, "rtk_pivotkv_score: bad shape L=%d D=%d", L, D);
    RTK_CHECK_ARG((cosv == nullptr) == (sinv == nullptr), "rtk_pivotkv_score: cos and sin must both be given or both NULL");
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16, "rtk_pivotkv_score: unsupported dtype %d", dtype);
    RTK_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "rtk_pivotkv_score: workspace must be 256-byte aligned");
    const ScoreWs w = score_ws(Hq, Hkv, L, D, dtype);
    if (workspace_bytes < w.total) {
        set_error("rtk_pivotkv_score: workspace %zu < required %zu bytes", workspace_bytes, w.total);
        return RTK_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RTK_BF16)
        return score_impl<RTK_BF16>(q, q_stride_h, q_stride_l, k, k_stride_h, k_stride_l, Hq, Hkv, L, D, cosv, sinv,
                                    attention_scaling, score, k_unrot, (char*)workspace, w, st);
    return score_impl<RTK_F32>(q, q_stride_h, q_stride_l, k, k_stride_h, k_stride_l, Hq, Hkv, L, D, cosv, sinv,
                               attention_scaling, score, k_unrot, (char*)workspace, w, st);
}

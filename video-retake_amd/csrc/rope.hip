// rope.hip — RoPE cos/sin tables for PivotKV's un-rotate / re-rotate steps.
// Replaces longvideo_cache.py:68-74 (M-RoPE section merge) and optionally the rotary_emb_fn calls
// at :249 and :298 for the standard inv_freq * position / attention_scaling rotary modules.
#include "common.cuh"

namespace rtk {

template <int DT>
__global__ __launch_bounds__(256) void rope_merge_kernel(const void* __restrict__ cin, const void* __restrict__ sin_,
                                                         int L, int D, RowSel rs, float* __restrict__ cos_out,
                                                         float* __restrict__ sin_out) {
    const size_t n = (size_t)L * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const size_t src = (size_t)rs.row[d] * n + i;
        if constexpr (DT != RTK_F32) {
            cos_out[i] = H16<DT>::ld(cin, src);
            sin_out[i] = H16<DT>::ld(sin_, src);
        } else {
            cos_out[i] = ((const float*)cin)[src];
            sin_out[i] = ((const float*)sin_)[src];
        }
    }
}

__global__ __launch_bounds__(256) void rope_table_kernel(const int64_t* __restrict__ pos, int64_t pos_ld, int L, int D,
                                                         const float* __restrict__ inv_freq, float scaling,
                                                         RowSel rs, int round_bf16, float* __restrict__ cos_out,
                                                         float* __restrict__ sin_out) {
    const size_t n = (size_t)L * D;
    const int h2 = D / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const size_t l = i / D;
        const float p = (float)pos[(size_t)rs.row[d] * pos_ld + l];
        const float ang = p * inv_freq[d < h2 ? d : d - h2];
        float s, c;
        sincos_cr(ang, s, c);
        c *= scaling;
        s *= scaling;
        c = round_to(c, round_bf16);   // 0 = fp32 tables, 1 = bf16, 2 = fp16
        s = round_to(s, round_bf16);
        cos_out[i] = c;
        sin_out[i] = s;
    }
}

// In-place extra rotation of already-rotated keys by `delta` temporal steps: k <- R(delta) k on the
// channels fed by position row 0 (all channels for 1-D RoPE).  R(a+b) = R(a) R(b), so keys rotated at
// provisional ids become keys rotated at ids + delta (8-GPU sharding: each rank compresses its chunks
// before the global temporal offsets are known).  delta is read from device memory.
template <int DT>
__global__ __launch_bounds__(256) void rope_shift_kernel(void* __restrict__ kv, int64_t stride_h, int H, int n, int D,
                                                         const int64_t* __restrict__ delta,
                                                         const float* __restrict__ inv_freq, RowSel rs) {
    const int h2 = D / 2;
    const size_t total = (size_t)H * n * h2;
    const float dl = (float)delta[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % h2);
        if (rs.row[d] != 0) continue;
        const size_t hr = i / h2;
        const int r = (int)(hr % n), h = (int)(hr / n);
        const size_t off = (size_t)h * stride_h + (size_t)r * D;
        float s, c;
        sincos_cr(dl * inv_freq[d], s, c);
        float x1, x2;
        if constexpr (DT != RTK_F32) {
            x1 = H16<DT>::ld(kv, off + d);
            x2 = H16<DT>::ld(kv, off + d + h2);
        } else {
            x1 = ((float*)kv)[off + d];
            x2 = ((float*)kv)[off + d + h2];
        }
        const float o1 = x1 * c - x2 * s, o2 = x2 * c + x1 * s;
        if constexpr (DT != RTK_F32) {
            H16<DT>::st(kv, off + d, o1);
            H16<DT>::st(kv, off + d + h2, o2);
        } else {
            ((float*)kv)[off + d] = o1;
            ((float*)kv)[off + d + h2] = o2;
        }
    }
}


// The same for a whole assembled cache in ONE launch (chunk sharding): k is [layers, H, world * seg, D] and
// the segment of rank r in layer l is rotated by table[r * layers + l] temporal steps - one launch instead of
// layers x world launches of rope_shift_kernel (224 at 8 ranks, each a few microseconds of work), same arithmetic.
template <int DT>
__global__ __launch_bounds__(256) void rope_shift_segments_kernel(void* __restrict__ kv, int64_t stride_layer, int64_t stride_h,
                                                                  int layers, int H, int world, int seg, int D,
                                                                  const int64_t* __restrict__ table,
                                                                  const float* __restrict__ inv_freq, RowSel rs) {
    const int h2 = D / 2;
    const size_t rows = (size_t)world * seg;
    const size_t total = (size_t)layers * H * rows * h2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % h2);
        if (rs.row[d] != 0) continue;
        size_t t = i / h2;
        const size_t row = t % rows;
        t /= rows;
        const int h = (int)(t % H), l = (int)(t / H);
        const int r = (int)(row / seg);
        const float dl = (float)table[(size_t)r * layers + l];
        const size_t off = (size_t)l * stride_layer + (size_t)h * stride_h + row * D;
        float s, c;
        sincos_cr(dl * inv_freq[d], s, c);
        float x1, x2;
        if constexpr (DT != RTK_F32) {
            x1 = H16<DT>::ld(kv, off + d);
            x2 = H16<DT>::ld(kv, off + d + h2);
        } else {
            x1 = ((float*)kv)[off + d];
            x2 = ((float*)kv)[off + d + h2];
        }
        const float o1 = x1 * c - x2 * s, o2 = x2 * c + x1 * s;
        if constexpr (DT != RTK_F32) {
            H16<DT>::st(kv, off + d, o1);
            H16<DT>::st(kv, off + d + h2, o2);
        } else {
            ((float*)kv)[off + d] = o1;
            ((float*)kv)[off + d + h2] = o2;
        }
    }
}

// Forward rotation IN PLACE of un-rotated key rows at given ids: k[l][h][r] <- (k*cos) + (rotate_half(k)*sin) with the
// cos / sin of ids[l][:, r] (longvideo_cache.py:297-306 for rows whose re-rotation was deferred: the chunk-sharded
// prefill keeps the kept keys un-rotated until the blocks' temporal offsets are known and rotates ONCE at the final
// ids).  Same arithmetic as the eviction kernel's re-rotation: rope_chunk tables, one rounding per torch op.
// One thread = one row x one 16-byte chunk pair, walking the heads.
template <int DT>
__global__ __launch_bounds__(256) void rope_rotate_rows_kernel(char* __restrict__ k, int64_t stride_layer, int64_t stride_h,
                                                               int H, int rows, int D, const int64_t* __restrict__ pos,
                                                               int64_t pos_stride_layer, int64_t pos_stride_p, int P,
                                                               const float* __restrict__ inv_freq, float scaling, RowSel rs,
                                                               int round_mode) {
    using V = Vec16<DT>;
    constexpr int VE = V::VE;
    constexpr int ES = 16 / VE;
    const int h2 = D / 2, lpr = h2 / VE;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = id / lpr;
    if (r >= rows) return;
    const int d = (id - r * lpr) * VE;
    const int layer = blockIdx.y;
    const int64_t* pl = pos + (size_t)layer * pos_stride_layer;
    float pid[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) pid[p] = (float)pl[(size_t)min(p, P - 1) * pos_stride_p + r];
    float c1[VE], s1[VE], c2[VE], s2[VE];
    rope_chunk<VE>(inv_freq, rs, d, h2, pid, scaling, round_mode, c1, s1, c2, s2);
    char* base = k + ((size_t)layer * stride_layer + (size_t)r * D) * ES;
    for (int h = 0; h < H; ++h) {
        char* row = base + (size_t)h * stride_h * ES;
        const u32x4 lo = *(const u32x4*)(row + (size_t)d * ES), hi = *(const u32x4*)(row + (size_t)(d + h2) * ES);
        u32x4 olo, ohi;
        rotate_chunk_pair<DT>(lo, hi, c1, s1, c2, s2, olo, ohi);
        *(u32x4*)(row + (size_t)d * ES) = olo;
        *(u32x4*)(row + (size_t)(d + h2) * ES) = ohi;
    }
}

// Temporal-id continuity fix of the attention patch (qwen2_vl.py:68-73), on the device: the whole row is
// shifted so that its first id follows the last id stored for the layer.  One workgroup: the first id is
// read by everyone before anyone writes.
__global__ __launch_bounds__(1024) void position_shift_kernel(int64_t* __restrict__ t, int n,
                                                             const int64_t* __restrict__ prev) {
    constexpr int E = 8;  // ids per thread per sweep, all loaded before the first id is overwritten
    const long long p = prev ? (long long)prev[0] : -1ll;
    const long long delta = p + 1 - (long long)t[0];
    for (int base = 0; base < n; base += E * 1024) {
        long long v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = base + e * 1024 + (int)threadIdx.x;
            v[e] = i < n ? (long long)t[i] : 0;
        }
        if (base == 0) __syncthreads();  // everybody has read t[0]
        if (delta != 0) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = base + e * 1024 + (int)threadIdx.x;
                if (i < n) t[i] = v[e] + delta;
            }
        }
    }
}

}  // namespace rtk

using namespace rtk;

extern "C" int rtk_position_shift(int64_t* temporal_ids, int n, const int64_t* prev_dev, rtk_stream_t stream) {
    RTK_CHECK_ARG(temporal_ids && n >= 1, "rtk_position_shift: NULL pointer or empty row");
    RTK_LAUNCH(KID_SHIFT, position_shift_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, temporal_ids, n, prev_dev);
    RTK_LAUNCH_CHECK("position_shift_kernel");
    return RTK_OK;
}

extern "C" int rtk_rope_rotate_rows(void* k, int64_t stride_layer, int64_t stride_h, int layers, int H, int rows, int D,
                                    int dtype, const int64_t* pos, int64_t pos_stride_layer, int64_t pos_stride_p, int P,
                                    const float* inv_freq, float attention_scaling, const int* sections_host, int nsec,
                                    int round_mode, rtk_stream_t stream) {
    RTK_CHECK_ARG(k && pos && inv_freq, "rtk_rope_rotate_rows: NULL pointer");
    RTK_CHECK_ARG(layers >= 1 && H >= 1 && rows >= 0 && D >= 2, "rtk_rope_rotate_rows: bad shape");
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_F16, "rtk_rope_rotate_rows: unsupported dtype %d", dtype);
    RTK_CHECK_ARG(P == 1 || P == 3, "rtk_rope_rotate_rows: P must be 1 or 3, got %d", P);
    RTK_CHECK_ARG(pos_stride_p >= rows, "rtk_rope_rotate_rows: pos_stride_p < rows");
    if (rows == 0) return RTK_OK;
    const int ve = dtype != RTK_F32 ? 8 : 4, es = dtype != RTK_F32 ? 2 : 4;
    if (D % (2 * ve) != 0 || D > 256 || (stride_h * es) % 16 || (stride_layer * es) % 16 || ((uintptr_t)k & 15)) {
        set_error("rtk_rope_rotate_rows: needs 16-byte aligned rows and head_dim a multiple of %d (<= 256)", 2 * ve);
        return RTK_EUNSUPPORTED;
    }
    RowSel rs;
    int rc = make_rowsel(rs, P, D, sections_host, nsec, "rtk_rope_rotate_rows");
    if (rc) return rc;
    const int threads = rows * (D / 2 / ve);
    const dim3 grid((threads + 255) / 256, layers);
#define RTK_RRR(DTV)                                                                                                  \
    RTK_LAUNCH(KID_ROPE, rope_rotate_rows_kernel<DTV>, grid, dim3(256), 0, (hipStream_t)stream, (char*)k, stride_layer, \
               stride_h, H, rows, D, pos, pos_stride_layer, pos_stride_p, P, inv_freq, attention_scaling, rs, round_mode)
    if (dtype == RTK_BF16) RTK_RRR(RTK_BF16);
    else if (dtype == RTK_F16) RTK_RRR(RTK_F16);
    else RTK_RRR(RTK_F32);
#undef RTK_RRR
    RTK_LAUNCH_CHECK("rope_rotate_rows_kernel");
    return RTK_OK;
}

extern "C" int rtk_rope_shift(void* k, int64_t stride_h, int H, int n, int D, int dtype, const int64_t* delta_dev,
                              const float* inv_freq, int P, const int* sections_host, int nsec, rtk_stream_t stream) {
    RTK_CHECK_ARG(k && delta_dev && inv_freq, "rtk_rope_shift: NULL pointer");
    RTK_CHECK_ARG(H >= 1 && n >= 0 && D >= 2, "rtk_rope_shift: bad shape");
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_F16, "rtk_rope_shift: unsupported dtype %d", dtype);
    if (n == 0) return RTK_OK;
    RowSel rs;
    int rc = make_rowsel(rs, P, D, sections_host, nsec, "rtk_rope_shift");
    if (rc) return rc;
    const size_t total = (size_t)H * n * (D / 2);
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 8192);
    if (dtype == RTK_BF16)
        RTK_LAUNCH(KID_ROPE, rope_shift_kernel<RTK_BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, k, stride_h, H, n,
                   D, delta_dev, inv_freq, rs);
    else if (dtype == RTK_F16)
        RTK_LAUNCH(KID_ROPE, rope_shift_kernel<RTK_F16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, k, stride_h, H, n,
                   D, delta_dev, inv_freq, rs);
    else
        RTK_LAUNCH(KID_ROPE, rope_shift_kernel<RTK_F32>, dim3(grid), dim3(256), 0, (hipStream_t)stream, k, stride_h, H, n,
                   D, delta_dev, inv_freq, rs);
    RTK_LAUNCH_CHECK("rope_shift_kernel");
    return RTK_OK;
}

extern "C" int rtk_rope_shift_segments(void* k, int64_t stride_layer, int64_t stride_h, int layers, int H, int world,
                                       int seg, int D, int dtype, const int64_t* table_dev, const float* inv_freq, int P,
                                       const int* sections_host, int nsec, rtk_stream_t stream) {
    RTK_CHECK_ARG(k && table_dev && inv_freq, "rtk_rope_shift_segments: NULL pointer");
    RTK_CHECK_ARG(layers >= 1 && H >= 1 && world >= 1 && seg >= 0 && D >= 2, "rtk_rope_shift_segments: bad shape");
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_F16, "rtk_rope_shift_segments: unsupported dtype %d", dtype);
    if (seg == 0) return RTK_OK;
    RowSel rs;
    int rc = make_rowsel(rs, P, D, sections_host, nsec, "rtk_rope_shift_segments");
    if (rc) return rc;
    const size_t total = (size_t)layers * H * world * seg * (D / 2);
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 65536);
#define RTK_RSS(DTV)                                                                                                    \
    RTK_LAUNCH(KID_ROPE, rope_shift_segments_kernel<DTV>, dim3(grid), dim3(256), 0, (hipStream_t)stream, k, stride_layer, \
               stride_h, layers, H, world, seg, D, table_dev, inv_freq, rs)
    if (dtype == RTK_BF16) RTK_RSS(RTK_BF16);
    else if (dtype == RTK_F16) RTK_RSS(RTK_F16);
    else RTK_RSS(RTK_F32);
#undef RTK_RSS
    RTK_LAUNCH_CHECK("rope_shift_segments_kernel");
    return RTK_OK;
}

extern "C" int rtk_rope_merge(const void* cos_in, const void* sin_in, int P, int L, int D, int dtype,
                              const int* sections_host, int nsec, float* cos_out, float* sin_out,
                              rtk_stream_t stream) {
    RTK_CHECK_ARG(cos_in && sin_in && cos_out && sin_out, "rtk_rope_merge: NULL pointer");
    RTK_CHECK_ARG(L >= 1 && D >= 2, "rtk_rope_merge: bad shape L=%d D=%d", L, D);
    RTK_CHECK_ARG(dtype == RTK_F32 || dtype == RTK_BF16 || dtype == RTK_F16, "rtk_rope_merge: unsupported dtype %d", dtype);
    RowSel rs;
    int rc = make_rowsel(rs, P, D, sections_host, nsec, "rtk_rope_merge");
    if (rc) return rc;
    const unsigned grid = (unsigned)std::min<size_t>(((size_t)L * D + 255) / 256, 4096);
    if (dtype == RTK_BF16)
        RTK_LAUNCH(KID_ROPE, rope_merge_kernel<RTK_BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, cos_in, sin_in, L,
                           D, rs, cos_out, sin_out);
    else if (dtype == RTK_F16)
        RTK_LAUNCH(KID_ROPE, rope_merge_kernel<RTK_F16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, cos_in, sin_in, L,
                           D, rs, cos_out, sin_out);
    else
        RTK_LAUNCH(KID_ROPE, rope_merge_kernel<RTK_F32>, dim3(grid), dim3(256), 0, (hipStream_t)stream, cos_in, sin_in, L,
                           D, rs, cos_out, sin_out);
    RTK_LAUNCH_CHECK("rope_merge_kernel");
    return RTK_OK;
}

extern "C" int rtk_rope_table(const int64_t* pos, int64_t pos_stride, int P, int L, const float* inv_freq, int D,
                              float attention_scaling,
                              const int* sections_host, int nsec, int round_bf16, float* cos_out, float* sin_out,
                              rtk_stream_t stream) {
    RTK_CHECK_ARG(pos && inv_freq && cos_out && sin_out, "rtk_rope_table: NULL pointer");
    RTK_CHECK_ARG(L >= 1 && D >= 2, "rtk_rope_table: bad shape L=%d D=%d", L, D);
    RTK_CHECK_ARG(pos_stride >= L, "rtk_rope_table: pos_stride %lld < L %d", (long long)pos_stride, L);
    RowSel rs;
    int rc = make_rowsel(rs, P, D, sections_host, nsec, "rtk_rope_table");
    if (rc) return rc;
    const unsigned grid = (unsigned)std::min<size_t>(((size_t)L * D + 255) / 256, 4096);
    RTK_LAUNCH(KID_ROPE, rope_table_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pos, pos_stride, L, D,
               inv_freq, attention_scaling, rs, round_bf16, cos_out, sin_out);
    RTK_LAUNCH_CHECK("rope_table_kernel");
    return RTK_OK;
}

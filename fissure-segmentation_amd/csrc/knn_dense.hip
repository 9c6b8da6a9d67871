// Dense kNN graph build for (B,C,N) clouds -- include/fsg_hip.h: fsg_knn_dense_f32.
// Replaces utils/general_utils.py:43-53,315-327 and models/dgcnn_opensrc.py:34-40 of the reference.
//
// v0 ("rows in LDS"): a 256-thread workgroup owns QB consecutive query points of one cloud.
//   phase A  every thread owns candidates j = tid, tid+256, ...: it streams the candidate's channels
//            from HBM/L2 (coalesced: N is the contiguous dim of (B,C,N)) and accumulates the QB dot
//            products against the query values broadcast from LDS; the finished distance row
//            d(q, 0..N) lives in LDS only -- the (B,N,N) matrix never exists in HBM.
//   phase B  one wave per query extracts the k smallest (distance, index) pairs by k rounds of a
//            wave-wide lexicographic arg-min over the LDS row.
// Arithmetic contract (bit-exact with oracle/fsg_oracle.c): channel-ordered fmaf chains from +0,
// d = (xx_i - 2*dot) + xx_j.
#include <float.h>

#include <stdlib.h>

#include "fsg_common.h"

namespace {

constexpr int BLOCK = 256;

template <int QB>
__global__ __launch_bounds__(BLOCK) void knn_dense_rows_kernel(
    const float *__restrict__ x, int N, int Npad, long sb, long sc, int c_knn, int k, int flags,
    int32_t *__restrict__ idx_out, float *__restrict__ dist_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *qv = smem;                  // [c_knn][QB] query channel values
    float *xxq = qv + c_knn * QB;      // [QB]
    float *rows = smem + (((c_knn * QB + QB) + 3) & ~3);  // [QB][Npad], 16-byte aligned

    const int b = blockIdx.y;
    const int q0 = blockIdx.x * QB;
    const int tid = threadIdx.x;
    const float *xb = x + (long)b * sb;

    for (int t = tid; t < c_knn * QB; t += BLOCK) {
        const int c = t / QB, q = t - c * QB;
        qv[t] = (q0 + q < N) ? xb[c * sc + q0 + q] : 0.f;
    }
    __syncthreads();
    if (tid < QB) {
        float a = 0.f;
        for (int c = 0; c < c_knn; ++c) a = __builtin_fmaf(qv[c * QB + tid], qv[c * QB + tid], a);
        xxq[tid] = a;
    }
    __syncthreads();

    // ---- phase A: distance rows into LDS
    for (int j = tid; j < Npad; j += BLOCK) {
        if (j < N) {
            float acc[QB];
#pragma unroll
            for (int q = 0; q < QB; ++q) acc[q] = 0.f;
            float xxj = 0.f;
            for (int c = 0; c < c_knn; ++c) {
                const float v = xb[c * sc + j];
                xxj = __builtin_fmaf(v, v, xxj);
#pragma unroll
                for (int q = 0; q < QB; ++q) acc[q] = __builtin_fmaf(qv[c * QB + q], v, acc[q]);
            }
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const float t = xxq[q] - 2.0f * acc[q];
                float d = t + xxj;
                if ((flags & FSG_KNN_FIX_DIAG) && j == q0 + q) d = 0.f;
                rows[q * Npad + j] = d;
            }
        } else {
#pragma unroll
            for (int q = 0; q < QB; ++q) rows[q * Npad + j] = INFINITY;
        }
    }
    __syncthreads();

    // ---- phase B: k rounds of wave arg-min per query
    const int lane = tid & 63, wave = tid >> 6;
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    const int kk = k + drop;
    for (int q = wave; q < QB; q += BLOCK / 64) {
        const int qg = q0 + q;
        if (qg >= N) break;
        float *row = rows + q * Npad;
        for (int r = 0; r < kk; ++r) {
            float bd = INFINITY;
            int bj = 0x7fffffff;
            for (int s = 0; s < Npad; s += 256) {
                const float4 v = *reinterpret_cast<const float4 *>(row + s + 4 * lane);
                const int j = s + 4 * lane;
                if (v.x < bd) { bd = v.x; bj = j; }
                if (v.y < bd) { bd = v.y; bj = j + 1; }
                if (v.z < bd) { bd = v.z; bj = j + 2; }
                if (v.w < bd) { bd = v.w; bj = j + 3; }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float od = __shfl_xor(bd, off);
                const int oj = __shfl_xor(bj, off);
                if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
            }
            if (lane == 0) {
                if (bj < N) row[bj] = INFINITY;  // consumed
                if (r >= drop) {
                    const long o = ((long)b * N + qg) * k + (r - drop);
                    idx_out[o] = bj < N ? bj : 0;
                    if (dist_out) dist_out[o] = bd;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

}  // namespace

int fsg_knn_rows_mfma_launch(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn, int k,
                             int flags, int32_t *idx_out, float *dist_out, float *xx_scratch, hipStream_t st);
size_t fsg_knn_split_workspace_bytes(int B, int N, int c_knn);
int fsg_knn_split_launch(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn, int k, int flags,
                         int32_t *idx_out, float *dist_out, void *ws, size_t ws_bytes, hipStream_t st);

extern "C" size_t fsg_knn_dense_workspace_bytes(int B, int N, int c_knn) {
    if (B <= 0 || N <= 0 || c_knn <= 0) return 0;
    const size_t xxb = sizeof(float) * (size_t)B * N, sp = fsg_knn_split_workspace_bytes(B, N, c_knn);
    return sp > xxb ? sp : xxb;
}

// Same contract as fsg_knn_dense_f32 with a caller-owned workspace of fsg_knn_dense_workspace_bytes(B, N, c_knn) bytes:
// inside its envelope the coarse-sweep + exact-refine kernel (knn_split.hip) builds the graph, everything else (and
// flag 2097152, A/B timing and cross-checks) goes to fsg_knn_dense_f32 with the workspace as its squared-norm scratch.
extern "C" int fsg_knn_dense_ws_f32(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn, int k,
                                    int flags, int32_t *idx_out, float *dist_out, void *workspace,
                                    size_t workspace_bytes, fsg_stream_t stream) {
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    FSG_REQUIRE(x && idx_out, "fsg_knn_dense_ws_f32: NULL pointer");
    FSG_REQUIRE(B >= 0 && N > 0 && c_knn > 0, "fsg_knn_dense_ws_f32: bad shape B=%d N=%d c_knn=%d", B, N, c_knn);
    FSG_REQUIRE(k >= 1 && k + drop <= N && k + drop <= FSG_KNN_MAX_K,
                "fsg_knn_dense_ws_f32: need 1 <= k and k+drop <= min(N, %d); got k=%d N=%d", FSG_KNN_MAX_K, k, N);
    FSG_REQUIRE(workspace == nullptr || workspace_bytes >= sizeof(float) * (size_t)B * N,
                "fsg_knn_dense_ws_f32: workspace of %zu bytes is smaller than the (B,N) squared norms", workspace_bytes);
    if (B == 0) return FSG_OK;
    if (workspace && !(flags & (2097152 | FSG_KNN_FORCE_ROWS | FSG_KNN_FORCE_MFMA | 4096 | 16384 | 131072 | 2048 | 8192))) {
        const int rc = fsg_knn_split_launch(x, B, N, stride_b, stride_c, c_knn, k, flags, idx_out, dist_out, workspace,
                                            workspace_bytes, (hipStream_t)stream);
        if (rc != FSG_ERR_UNSUPPORTED) return rc;
    }
    return fsg_knn_dense_f32(x, B, N, stride_b, stride_c, c_knn, k, flags & ~2097152, idx_out, dist_out,
                             static_cast<float *>(workspace), stream);
}

int fsg_knn_split_launch_ex(const float *x, const float *prepared_xt, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn,
                            int k, int flags, int32_t *idx_out, float *dist_out, void *ws, size_t ws_bytes, hipStream_t st);
int fsg_knn_split_launch_pq(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn, int k, int flags,
                            int32_t *idx_out, float *dist_out, void *ws, size_t ws_bytes, hipStream_t st, const float *pq_w,
                            int pq_rows, float *pq_out, bool *fused);
int fsg_knn_pq_rows_launch(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn, const float *pq_w,
                           int pq_rows, float *pq_out, hipStream_t st);

// fsg_knn_dense_ws_f32 over points of up to four channels PLUS their per-point product with a small weight:
// pq_out (B, N, rows_pq) = x^T w_pq^T, w_pq (rows_pq, c_knn) row-major -- the "one plain GEMM" of the FIRST EdgeConv's contract
// (models/dgcnn.py:212-243: its first 1x1 conv decomposed per point, fsg_edge_weights_many_f32 makes the weight), K = c_knn <= 4.
// Inside the coarse-sweep kernel's no-prep path (N = 2048, 16-byte addressable rows) the rows come out of the graph build's first
// launch, which holds the workgroup's points anyway; otherwise a small launch of its own follows the build.  256 % rows_pq == 0.
extern "C" int fsg_knn_dense_ws_pq_f32(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn, int k,
                                       int flags, int32_t *idx_out, float *dist_out, void *workspace, size_t workspace_bytes,
                                       const float *w_pq, int rows_pq, float *pq_out, fsg_stream_t stream) {
    FSG_REQUIRE(x && idx_out && w_pq && pq_out, "fsg_knn_dense_ws_pq_f32: NULL pointer");
    FSG_REQUIRE(c_knn >= 1 && c_knn <= 4 && rows_pq >= 1 && 256 % rows_pq == 0,
                "fsg_knn_dense_ws_pq_f32: needs 1 <= c_knn <= 4 and rows_pq dividing 256; got c_knn=%d rows_pq=%d", c_knn, rows_pq);
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    FSG_REQUIRE(B >= 0 && N > 0 && k >= 1 && k + drop <= N && k + drop <= FSG_KNN_MAX_K, "fsg_knn_dense_ws_pq_f32: bad shape B=%d N=%d k=%d",
                B, N, k);
    if (B == 0) return FSG_OK;
    hipStream_t st = (hipStream_t)stream;
    if (workspace && !(flags & (2097152 | FSG_KNN_FORCE_ROWS | FSG_KNN_FORCE_MFMA | 4096 | 16384 | 131072 | 2048 | 8192))) {
        bool fused = false;
        const int rc = fsg_knn_split_launch_pq(x, B, N, stride_b, stride_c, c_knn, k, flags, idx_out, dist_out, workspace,
                                               workspace_bytes, st, w_pq, rows_pq, pq_out, &fused);
        if (rc == FSG_OK) return fused ? FSG_OK : fsg_knn_pq_rows_launch(x, B, N, stride_b, stride_c, c_knn, w_pq, rows_pq, pq_out, st);
        if (rc != FSG_ERR_UNSUPPORTED) return rc;
    }
    const int rc = fsg_knn_dense_ws_f32(x, B, N, stride_b, stride_c, c_knn, k, flags, idx_out, dist_out, workspace, workspace_bytes,
                                        stream);
    if (rc != FSG_OK) return rc;
    return fsg_knn_pq_rows_launch(x, B, N, stride_b, stride_c, c_knn, w_pq, rows_pq, pq_out, st);
}

// The graph build of fsg_knn_dense_ws_f32 when the producer of the points has already PREPARED it (fsg_edgeconv_apply_f32 with a
// knn_workspace): x_pm = the point-major (B, N, c_knn) copy of the points, workspace = the one handed to the producer.  Same
// result bits as fsg_knn_dense_f32 on the same points.  Only the shapes the coarse-sweep kernel takes in its fp16 form:
// c_knn in {16, 32, 64}, N % 64 == 0, 1024 <= N <= 8192, k + drop <= 64 (FSG_ERR_UNSUPPORTED otherwise: use fsg_knn_dense_ws_f32).
extern "C" int fsg_knn_dense_prepared_f32(const float *x_pm, int B, int N, int c_knn, int k, int flags, int32_t *idx_out,
                                          float *dist_out, void *workspace, size_t workspace_bytes, fsg_stream_t stream) {
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    FSG_REQUIRE(x_pm && idx_out && workspace, "fsg_knn_dense_prepared_f32: NULL pointer");
    FSG_REQUIRE(B > 0 && N > 0 && c_knn > 0 && k >= 1 && k + drop <= N && k + drop <= FSG_KNN_MAX_K,
                "fsg_knn_dense_prepared_f32: bad shape B=%d N=%d c_knn=%d k=%d", B, N, c_knn, k);
    const int rc = fsg_knn_split_launch_ex(nullptr, x_pm, B, N, 0, 0, c_knn, k, flags, idx_out, dist_out, workspace, workspace_bytes,
                                           (hipStream_t)stream);
    if (rc == FSG_ERR_UNSUPPORTED) fsg_set_error("fsg_knn_dense_prepared_f32: shape outside the prepared path's envelope");
    return rc;
}

extern "C" int fsg_knn_dense_f32(const float *x, int B, int N, int64_t stride_b, int64_t stride_c,
                                 int c_knn, int k, int flags, int32_t *idx_out, float *dist_out,
                                 float *xx_scratch, fsg_stream_t stream) {
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    FSG_REQUIRE(x && idx_out, "fsg_knn_dense_f32: NULL pointer");
    FSG_REQUIRE(B >= 0 && N > 0 && c_knn > 0, "fsg_knn_dense_f32: bad shape B=%d N=%d c_knn=%d", B, N, c_knn);
    FSG_REQUIRE(k >= 1 && k + drop <= N && k + drop <= FSG_KNN_MAX_K,
                "fsg_knn_dense_f32: need 1 <= k and k+drop <= min(N, %d); got k=%d N=%d", FSG_KNN_MAX_K, k, N);
    FSG_REQUIRE(N <= 32768, "fsg_knn_dense_f32: N=%d > 32768 unsupported", N);
    if (B == 0) return FSG_OK;
    hipStream_t st = (hipStream_t)stream;
    // The superseded design kept as a cross-check (first matrix-core kernel, flag 8; 4096 / 16384 were the removed pipeline / filter) is
    // test / benchmark infrastructure and live in libfsg_hip_experiments.so (csrc/knn_experiments.hip), not in the product.
    FSG_REQUIRE(!(flags & (FSG_KNN_FORCE_MFMA | 4096 | 16384)),
                "fsg_knn_dense_f32: flags %d select an experimental kernel: call fsg_knn_experiment_f32 of "
                "libfsg_hip_experiments.so", flags);
    if (!(flags & FSG_KNN_FORCE_ROWS)) {  // production path
        const int rc = fsg_knn_rows_mfma_launch(x, B, N, stride_b, stride_c, c_knn, k, flags, idx_out, dist_out, xx_scratch, st);
        if (rc != FSG_ERR_UNSUPPORTED) return rc;
    }
    const int Npad = (N + 255) & ~255;
    auto launch = [&](auto kern, int QB) -> int {
        const size_t lds = sizeof(float) * ((((size_t)c_knn * QB + QB + 3) & ~(size_t)3) + (size_t)QB * Npad);
        if (lds > 160 * 1024) {
            fsg_set_error("fsg_knn_dense_f32: LDS need %zu B > 160 KiB (N=%d c_knn=%d)", lds, N, c_knn);
            return FSG_ERR_UNSUPPORTED;
        }
        static FsgLdsGrant grant;  // per instantiation (the lambda is instantiated per kernel)
        if (!grant.raise((const void *)kern, lds)) {
            fsg_set_error("fsg_knn_dense_f32: cannot raise dynamic LDS to %zu", lds);
            return FSG_ERR_HIP;
        }
        dim3 grid(fsg_cdiv(N, QB), B);
        hipLaunchKernelGGL(kern, grid, dim3(BLOCK), lds, st, x, N, Npad, (long)stride_b, (long)stride_c, c_knn, k,
                           flags, idx_out, dist_out);
        FSG_CHECK_LAUNCH("fsg_knn_dense_f32");
        return FSG_OK;
    };
    if (N <= 4096) return launch(knn_dense_rows_kernel<8>, 8);
    if (N <= 8192) return launch(knn_dense_rows_kernel<4>, 4);
    if (N <= 16384) return launch(knn_dense_rows_kernel<2>, 2);
    return launch(knn_dense_rows_kernel<1>, 1);
}

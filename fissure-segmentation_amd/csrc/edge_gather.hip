// Edge features [x_j - x_i ; x_i] and their transpose -- include/fsg_hip.h: fsg_edge_gather_*_f32.
// Replaces models/dgcnn.py:31-36 and models/dgcnn_opensrc.py:43-66 of the reference.
//
// HBM-bound: per point and layer the forward writes 2*C*k*4 bytes and reads 4*C + 4*k; lanes walk the
// flattened (i, s) edge axis, which is the contiguous axis of both idx (B,N,k) and edge (B,2C,N,k), so
// every index load and every edge store is a full 256-byte wave access; the x_j gathers hit L2 (one
// cloud's (C,N) slab is <= 2 MB).
#include "fsg_common.h"

namespace {

constexpr int BLOCK = 256;
constexpr int CCHUNK = 8;  // channels per workgroup: amortises the idx load, keeps the grid large

// T = float or __bf16 (storage type of x / edge; the subtraction is done in fp32 and rounded once)
template <typename T>
__global__ __launch_bounds__(BLOCK) void edge_fwd_kernel(const T *__restrict__ x, const int32_t *__restrict__ idx,
                                                          T *__restrict__ edge, int C, int N, int k) {
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * CCHUNK;
    const long NK = (long)N * k;
    const long e = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= NK) return;
    const int i = (int)(e / k);
    const int j = idx[(long)b * NK + e];
    const T *xb = x + (long)b * C * N;
    T *rel = edge + (long)b * 2 * C * NK + e;
    const int cend = min(c0 + CCHUNK, C);
    for (int c = c0; c < cend; ++c) {
        const float xi = (float)xb[(long)c * N + i];
        const float xj = (float)xb[(long)c * N + j];
        rel[(long)c * NK] = (T)(xj - xi);
        rel[(long)(C + c) * NK] = (T)xi;
    }
}

// grad_x[b,c,i] += sum_s (g_ctr - g_rel)[b,c,i,s] ; grad_x[b,c,idx[b,i,s]] += g_rel[b,c,i,s]
template <typename T>
__global__ __launch_bounds__(BLOCK) void edge_bwd_kernel(const T *__restrict__ g, const int32_t *__restrict__ idx,
                                                          float *__restrict__ gx, int C, int N, int k) {
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * CCHUNK;
    const long NK = (long)N * k;
    const long e = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= NK) return;
    const int i = (int)(e / k);
    const int j = idx[(long)b * NK + e];
    float *gxb = gx + (long)b * C * N;
    const T *gr = g + (long)b * 2 * C * NK + e;
    const int cend = min(c0 + CCHUNK, C);
    for (int c = c0; c < cend; ++c) {
        const float r = (float)gr[(long)c * NK];
        const float ct = (float)gr[(long)(C + c) * NK];
        atomicAdd(gxb + (long)c * N + j, r);
        atomicAdd(gxb + (long)c * N + i, ct - r);
    }
}

}  // namespace

extern "C" int fsg_edge_gather_fwd_f32(const float *x, const int32_t *idx, float *edge, int B, int C, int N, int k,
                                       fsg_stream_t stream) {
    FSG_REQUIRE(x && idx && edge, "fsg_edge_gather_fwd_f32: NULL pointer");
    FSG_REQUIRE(B >= 0 && C > 0 && N > 0 && k > 0, "fsg_edge_gather_fwd_f32: bad shape");
    FSG_REQUIRE(B <= 65535 && fsg_cdiv(C, CCHUNK) <= 65535, "fsg_edge_gather_fwd_f32: grid too large");
    if (B == 0) return FSG_OK;
    dim3 grid(fsg_cdiv((long)N * k, BLOCK), fsg_cdiv(C, CCHUNK), B);
    hipLaunchKernelGGL(edge_fwd_kernel<float>, grid, dim3(BLOCK), 0, (hipStream_t)stream, x, idx, edge, C, N, k);
    FSG_CHECK_LAUNCH("fsg_edge_gather_fwd_f32");
    return FSG_OK;
}

extern "C" int fsg_edge_gather_bwd_f32(const float *grad_edge, const int32_t *idx, float *grad_x, int B, int C, int N,
                                       int k, fsg_stream_t stream) {
    FSG_REQUIRE(grad_edge && idx && grad_x, "fsg_edge_gather_bwd_f32: NULL pointer");
    FSG_REQUIRE(B >= 0 && C > 0 && N > 0 && k > 0, "fsg_edge_gather_bwd_f32: bad shape");
    FSG_REQUIRE(B <= 65535 && fsg_cdiv(C, CCHUNK) <= 65535, "fsg_edge_gather_bwd_f32: grid too large");
    if (B == 0) return FSG_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(grad_x, 0, sizeof(float) * (size_t)B * C * N, st) != hipSuccess) {
        fsg_set_error("fsg_edge_gather_bwd_f32: memset failed");
        return FSG_ERR_HIP;
    }
    dim3 grid(fsg_cdiv((long)N * k, BLOCK), fsg_cdiv(C, CCHUNK), B);
    hipLaunchKernelGGL(edge_bwd_kernel<float>, grid, dim3(BLOCK), 0, st, grad_edge, idx, grad_x, C, N, k);
    FSG_CHECK_LAUNCH("fsg_edge_gather_bwd_f32");
    return FSG_OK;
}

// bf16 storage (fsg_edge_gather_{fwd,bwd}_bf16; 8d: s = 2 bytes per feature element): x and edge are bf16, the
// difference is formed in fp32 and rounded once; the backward reads a bf16 grad_edge and accumulates grad_x in FP32
// (hardware fp32 atomics; there is no bf16 atomic add) -- the caller rounds it to the leaf's type.
extern "C" int fsg_edge_gather_fwd_bf16(const void *x, const int32_t *idx, void *edge, int B, int C, int N, int k,
                                        fsg_stream_t stream) {
    FSG_REQUIRE(x && idx && edge, "fsg_edge_gather_fwd_bf16: NULL pointer");
    FSG_REQUIRE(B >= 0 && C > 0 && N > 0 && k > 0, "fsg_edge_gather_fwd_bf16: bad shape");
    FSG_REQUIRE(B <= 65535 && fsg_cdiv(C, CCHUNK) <= 65535, "fsg_edge_gather_fwd_bf16: grid too large");
    if (B == 0) return FSG_OK;
    dim3 grid(fsg_cdiv((long)N * k, BLOCK), fsg_cdiv(C, CCHUNK), B);
    hipLaunchKernelGGL(edge_fwd_kernel<__bf16>, grid, dim3(BLOCK), 0, (hipStream_t)stream, (const __bf16 *)x, idx,
                       (__bf16 *)edge, C, N, k);
    FSG_CHECK_LAUNCH("fsg_edge_gather_fwd_bf16");
    return FSG_OK;
}

extern "C" int fsg_edge_gather_bwd_bf16(const void *grad_edge, const int32_t *idx, float *grad_x, int B, int C, int N,
                                        int k, fsg_stream_t stream) {
    FSG_REQUIRE(grad_edge && idx && grad_x, "fsg_edge_gather_bwd_bf16: NULL pointer");
    FSG_REQUIRE(B >= 0 && C > 0 && N > 0 && k > 0, "fsg_edge_gather_bwd_bf16: bad shape");
    FSG_REQUIRE(B <= 65535 && fsg_cdiv(C, CCHUNK) <= 65535, "fsg_edge_gather_bwd_bf16: grid too large");
    if (B == 0) return FSG_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(grad_x, 0, sizeof(float) * (size_t)B * C * N, st) != hipSuccess) {
        fsg_set_error("fsg_edge_gather_bwd_bf16: memset failed");
        return FSG_ERR_HIP;
    }
    dim3 grid(fsg_cdiv((long)N * k, BLOCK), fsg_cdiv(C, CCHUNK), B);
    hipLaunchKernelGGL(edge_bwd_kernel<__bf16>, grid, dim3(BLOCK), 0, st, (const __bf16 *)grad_edge, idx, grad_x, C, N, k);
    FSG_CHECK_LAUNCH("fsg_edge_gather_bwd_bf16");
    return FSG_OK;
}

// The reference's create_neighbor_features (models/dgcnn.py:15-36) as ONE entry point: dynamic graph over the first c_knn
// channels (self included, models/dgcnn.py:26-27) + the (B,2C,N,k) edge tensor.  SURVEY 8(b): fsg_knn_gather_fused_*.
// (The training path never materialises the edge tensor -- fsg_edgeconv{1,2}_* -- this is the reference-semantic op for
// callers that want exactly what the reference function returns; the graph is returned too.)
extern "C" int fsg_knn_gather_fused_f32(const float *x, int B, int C, int N, int k, int c_knn, int32_t *idx_out, float *edge,
                                        float *xx_scratch, fsg_stream_t stream) {
    FSG_REQUIRE(x && idx_out && edge && xx_scratch, "fsg_knn_gather_fused_f32: NULL pointer");
    FSG_REQUIRE(c_knn > 0 && c_knn <= C, "fsg_knn_gather_fused_f32: c_knn=%d outside 1..C=%d", c_knn, C);
    int rc = fsg_knn_dense_f32(x, B, N, (int64_t)C * N, (int64_t)N, c_knn, k, FSG_KNN_FIX_DIAG, idx_out, nullptr, xx_scratch,
                               stream);
    if (rc != FSG_OK) return rc;
    return fsg_edge_gather_fwd_f32(x, idx_out, edge, B, C, N, k, stream);
}

// The same call with the workspace of fsg_knn_dense_workspace_bytes(B, N, c_knn): the graph comes from fsg_knn_dense_ws_f32
// (coarse-sweep + exact-refine kernel inside its envelope: same indices).
extern "C" int fsg_knn_gather_fused_ws_f32(const float *x, int B, int C, int N, int k, int c_knn, int32_t *idx_out,
                                           float *edge, void *workspace, size_t workspace_bytes, fsg_stream_t stream) {
    FSG_REQUIRE(x && idx_out && edge && workspace, "fsg_knn_gather_fused_ws_f32: NULL pointer");
    FSG_REQUIRE(c_knn > 0 && c_knn <= C, "fsg_knn_gather_fused_ws_f32: c_knn=%d outside 1..C=%d", c_knn, C);
    int rc = fsg_knn_dense_ws_f32(x, B, N, (int64_t)C * N, (int64_t)N, c_knn, k, FSG_KNN_FIX_DIAG, idx_out, nullptr, workspace,
                                  workspace_bytes, stream);
    if (rc != FSG_OK) return rc;
    return fsg_edge_gather_fwd_f32(x, idx_out, edge, B, C, N, k, stream);
}

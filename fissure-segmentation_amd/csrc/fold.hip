// First layer of a folding MLP -- include/fsg_hip.h: fsg_fold_layer1_f32.
//
// models/folding_net.py:205-221 (FoldingDecoder.forward) feeds Conv1d(E + cp, W, 1) with cat([code repeated over the m grid
// points, the cp = 2|3 grid / intermediate coordinates]) and applies ReLU.  The code part of that 1x1 conv is constant per
// cloud (one small GEMM, done by the caller); what is left per point is a cp-term dot product -- as a GEMM + broadcast add +
// ReLU that is a K = 2|3 product WRITING the (B*m, W) tensor, a pass reading and writing it, and another one (335 MB of
// traffic at B*m = 32768, W = 512).  Here: out[b,i,:] = relu(per_cloud[b,:] + sum_j pts[b,i,j] * w[:,j]), written once
// (fma chain over j = 0..cp-1 starting from the per-cloud value; a padded coordinate multiplies a zero weight).
#include "fsg_common.h"

namespace {

constexpr int FOLD_PTS = 32;   // points per workgroup

// thread = one group of four output channels (its 4 x cp weights and its per-cloud values stay in registers), the workgroup
// walks FOLD_PTS points whose coordinates are wave-uniform (scalar loads): per 16-byte store no vector load at all.  (One
// thread per (point, channel group) re-read 12 weights and 3 coordinates for every store: 115 us instead of ~20.)
__global__ __launch_bounds__(256) void fold_layer1_kernel(const float *__restrict__ pts, int cp, const float *__restrict__ w,
                                                          long ldw, const float *__restrict__ per_cloud, int m, int Cout,
                                                          int relu, float *__restrict__ out) {
    const int b = blockIdx.y;
    const int c = (blockIdx.z * 256 + threadIdx.x) * 4;
    if (c >= Cout) return;
    float wv[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 3; ++j) wv[u][j] = j < cp ? w[(long)(c + u) * ldw + j] : 0.f;
    const float4 pc = *reinterpret_cast<const float4 *>(per_cloud + (long)b * Cout + c);
    const int i0 = blockIdx.x * FOLD_PTS, i1 = min(m, i0 + FOLD_PTS);
    for (int i = i0; i < i1; ++i) {
        const float *p = pts + ((long)b * m + i) * cp;   // uniform address: scalar loads
        const float p0 = p[0], p1 = cp > 1 ? p[1] : 0.f, p2 = cp > 2 ? p[2] : 0.f;
        float o[4] = {pc.x, pc.y, pc.z, pc.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            o[u] = __builtin_fmaf(p0, wv[u][0], o[u]);
            o[u] = __builtin_fmaf(p1, wv[u][1], o[u]);
            o[u] = __builtin_fmaf(p2, wv[u][2], o[u]);
            if (relu) o[u] = fmaxf(o[u], 0.f);
        }
        *reinterpret_cast<float4 *>(out + ((long)b * m + i) * Cout + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace

extern "C" int fsg_fold_layer1_f32(const float *pts, int cp, const float *w, int64_t ldw, const float *per_cloud, int B, int m,
                                   int Cout, int relu, float *out, fsg_stream_t stream) {
    FSG_REQUIRE(B >= 0 && m >= 0 && cp >= 1 && cp <= 3 && Cout >= 4 && Cout % 4 == 0 && ldw >= cp,
                "fsg_fold_layer1_f32: bad shape B=%d m=%d cp=%d Cout=%d ldw=%ld", B, m, cp, Cout, (long)ldw);
    if (B == 0 || m == 0) return FSG_OK;
    FSG_REQUIRE(pts && w && per_cloud && out, "fsg_fold_layer1_f32: NULL pointer");
    FSG_REQUIRE(B <= 65535 && (((uintptr_t)per_cloud | (uintptr_t)out) & 15) == 0,
                "fsg_fold_layer1_f32: per_cloud / out must be 16-byte aligned, B <= 65535");
    hipLaunchKernelGGL(fold_layer1_kernel, dim3(fsg_cdiv(m, FOLD_PTS), B, fsg_cdiv(Cout >> 2, 256)), dim3(256), 0,
                       (hipStream_t)stream, pts, cp, w, (long)ldw, per_cloud, m, Cout, relu, out);
    FSG_CHECK_LAUNCH("fsg_fold_layer1_f32");
    return FSG_OK;
}

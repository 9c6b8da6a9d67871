// Test-time ensembling over random point subsets -- include/fsg_hip.h: fsg_ensemble_accumulate_f32.
//
// models/point_seg_net.py:21-48 (`predict_full_pointcloud`, the one thing the reference itself times, train.py:383-392)
// runs 50 forwards on random `sample_points`-subsets of a cloud, one after the other, each followed by
//     softmax_accumulation[..., perm] += softmax(net(pc[..., perm]))
// In eval mode the subsets are independent clouds, so the host side (models/point_seg_net.py here) runs them as ONE
// batch; this file is the accumulation of all R runs in one pass:
//   ens_invert_kernel     inv[r][p] = slot of point p in run r, or -1   (a run may name a point twice in the fill-up phase:
//                         the highest slot wins -- the reference's indexed `+=` also adds ONE of the duplicates)
//   ens_accumulate_kernel one thread per (cloud, point): its runs in run order r = 0..R-1 (the order of the reference's
//                         loop, so the fp32 sums associate identically), softmax over the classes in registers.
// No atomics on floats: reproducible.
#include "fsg_common.h"

namespace {

constexpr int ENS_MAX_CLS = 32;

__global__ __launch_bounds__(256) void ens_invert_kernel(const int64_t *__restrict__ pts, int R, int S, long P,
                                                         int32_t *__restrict__ inv) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)R * S) return;
    const int r = (int)(t / S), s = (int)(t - (long)r * S);
    const int64_t p = pts[t];
    if (p >= 0 && p < P) atomicMax(&inv[(long)r * P + p], s);
}

__global__ __launch_bounds__(256) void ens_accumulate_kernel(const float *__restrict__ logits, const int32_t *__restrict__ inv,
                                                             int R, int B, int cls, int S, long P, float *__restrict__ acc) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= P) return;
    float a[ENS_MAX_CLS];
    bool touched = false;
    for (int r = 0; r < R; ++r) {
        const int s = inv[(long)r * P + p];
        if (s < 0) continue;
        const float *l = logits + (((long)r * B + b) * cls) * S + s;
        if (!touched) {
#pragma unroll
            for (int c = 0; c < ENS_MAX_CLS; ++c)
                if (c < cls) a[c] = acc[((long)b * cls + c) * P + p];
            touched = true;
        }
        float v[ENS_MAX_CLS], m = -INFINITY;
#pragma unroll
        for (int c = 0; c < ENS_MAX_CLS; ++c)
            if (c < cls) {
                v[c] = l[(long)c * S];
                m = fmaxf(m, v[c]);
            }
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < ENS_MAX_CLS; ++c)
            if (c < cls) {
                v[c] = expf(v[c] - m);
                sum += v[c];
            }
#pragma unroll
        for (int c = 0; c < ENS_MAX_CLS; ++c)
            if (c < cls) a[c] += v[c] / sum;
    }
    if (touched) {
#pragma unroll
        for (int c = 0; c < ENS_MAX_CLS; ++c)
            if (c < cls) acc[((long)b * cls + c) * P + p] = a[c];
    }
}

}  // namespace

extern "C" size_t fsg_ensemble_accumulate_workspace_bytes(int R, int64_t n_points) {
    if (R <= 0 || n_points <= 0) return 0;
    return sizeof(int32_t) * (size_t)R * (size_t)n_points;
}

extern "C" int fsg_ensemble_accumulate_f32(const float *logits, int R, int B, int cls, int S, const int64_t *pts,
                                           int64_t n_points, float *acc, void *workspace, fsg_stream_t stream) {
    FSG_REQUIRE(R >= 0 && B >= 0 && S >= 0 && n_points >= 0, "fsg_ensemble_accumulate_f32: negative size");
    if (R == 0 || B == 0 || S == 0 || n_points == 0) return FSG_OK;
    FSG_REQUIRE(logits && pts && acc && workspace, "fsg_ensemble_accumulate_f32: NULL pointer");
    FSG_REQUIRE(cls >= 1 && cls <= ENS_MAX_CLS, "fsg_ensemble_accumulate_f32: cls=%d outside 1..%d", cls, ENS_MAX_CLS);
    FSG_REQUIRE(B <= 65535, "fsg_ensemble_accumulate_f32: B=%d too large", B);
    hipStream_t st = (hipStream_t)stream;
    int32_t *inv = (int32_t *)workspace;
    if (hipMemsetAsync(inv, 0xff, sizeof(int32_t) * (size_t)R * (size_t)n_points, st) != hipSuccess) {
        fsg_set_error("fsg_ensemble_accumulate_f32: memset failed");
        return FSG_ERR_HIP;
    }
    hipLaunchKernelGGL(ens_invert_kernel, dim3(fsg_cdiv((long)R * S, 256)), dim3(256), 0, st, pts, R, S, (long)n_points, inv);
    FSG_CHECK_LAUNCH("fsg_ensemble_accumulate_f32/invert");
    hipLaunchKernelGGL(ens_accumulate_kernel, dim3(fsg_cdiv(n_points, 256), B), dim3(256), 0, st, logits, inv, R, B, cls, S,
                       (long)n_points, acc);
    FSG_CHECK_LAUNCH("fsg_ensemble_accumulate_f32/accumulate");
    return FSG_OK;
}

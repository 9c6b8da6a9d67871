// On-device sampling + rigid/scale augmentation of point clouds -- include/fsg_hip.h: fsg_sample_transform_f32.
//
// The step in front of the hot path: data.py:435-460 (`PointDataset.__getitem__`) draws a random subset of `sample_points`
// columns of an item's (C, N_full) tensor after augmentations.py:52-113 has moved its three coordinate rows by a random
// rotation, scale and translation.  Both are per-point operations, so subset and transform commute and one pass does
// both: out[b, :, i] = (A_b x[b, 0:3, j] + t_b  |  x[b, 3:, j]),  j = sample[b, i].
// HBM-bound and tiny (C * S * B floats); the point is to keep the batch on the device, not the kernel's speed.
#include "fsg_common.h"

namespace {

__global__ __launch_bounds__(256) void sample_transform_kernel(const float *__restrict__ x, int C, long N,
                                                               const int64_t *__restrict__ sample, int S,
                                                               const float *__restrict__ affine, float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= S) return;
    long j = sample ? sample[(long)b * S + i] : i;
    if (j < 0 || j >= N) j = 0;   // (the host wrapper checks its indices; never read outside the cloud)
    const float *xb = x + (long)b * C * N;
    float *ob = out + (long)b * C * S;
    int c0 = 0;
    if (affine && C >= 3) {
        const float *m = affine + (long)b * 12;   // row-major [A | t], 3 x 4, column-vector form
        const float p0 = xb[j], p1 = xb[N + j], p2 = xb[2 * N + j];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            ob[(long)r * S + i] =
                __builtin_fmaf(m[4 * r + 2], p2, __builtin_fmaf(m[4 * r + 1], p1, m[4 * r] * p0)) + m[4 * r + 3];
        c0 = 3;
    }
    for (int c = c0; c < C; ++c) ob[(long)c * S + i] = xb[(long)c * N + j];
}

}  // namespace

extern "C" int fsg_sample_transform_f32(const float *x, int B, int C, int64_t N, const int64_t *sample, int S,
                                        const float *affine, float *out, fsg_stream_t stream) {
    FSG_REQUIRE(B >= 0 && C >= 1 && N >= 1 && S >= 0, "fsg_sample_transform_f32: bad shape B=%d C=%d N=%ld S=%d", B, C, (long)N, S);
    FSG_REQUIRE(affine == nullptr || C >= 3, "fsg_sample_transform_f32: a transform needs the three coordinate rows (C=%d)", C);
    FSG_REQUIRE(sample != nullptr || (int64_t)S == N, "fsg_sample_transform_f32: without a sample S must equal N");
    if (B == 0 || S == 0) return FSG_OK;
    FSG_REQUIRE(x && out, "fsg_sample_transform_f32: NULL pointer");
    FSG_REQUIRE(B <= 65535, "fsg_sample_transform_f32: B=%d too large", B);
    hipLaunchKernelGGL(sample_transform_kernel, dim3(fsg_cdiv(S, 256), B), dim3(256), 0, (hipStream_t)stream, x, C, (long)N,
                       sample, S, affine, out);
    FSG_CHECK_LAUNCH("fsg_sample_transform_f32");
    return FSG_OK;
}

// Chamfer nearest neighbour + its gradient -- include/fsg_hip.h: fsg_chamfer_nn_f32 / _bwd_f32.
// Replaces the pytorch3d.loss.chamfer_distance call of losses/chamfer_loss.py:19.
//
// VALU-bound (B*N*M pair evaluations, ~8 flop each; inputs are a few hundred KB): one lane per query
// point, the other cloud streamed through LDS in 1024-point tiles that every lane reads as a
// broadcast.  d = fma(dz,dz, fma(dy,dy, dx*dx)) -- bit-exact with oracle/fsg_oracle.c.
#include "fsg_common.h"

namespace {

constexpr int BLOCK = 128;
constexpr int TILE = 1024;

__global__ __launch_bounds__(BLOCK) void chamfer_nn_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                            int N, int M, float *__restrict__ dist,
                                                            int32_t *__restrict__ arg) {
    __shared__ float ty[TILE * 3];
    const int b = blockIdx.y;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const float *yb = y + (long)b * M * 3;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (i < N) {
        const float *p = x + ((long)b * N + i) * 3;
        px = p[0]; py = p[1]; pz = p[2];
    }
    float best = INFINITY;
    int bj = 0;
    for (int j0 = 0; j0 < M; j0 += TILE) {
        const int cnt = min(TILE, M - j0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt * 3; t += BLOCK) ty[t] = yb[(long)j0 * 3 + t];
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const float dx = px - ty[3 * j], dy = py - ty[3 * j + 1], dz = pz - ty[3 * j + 2];
            const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            if (d < best) { best = d; bj = j0 + j; }
        }
    }
    if (i < N) {
        dist[(long)b * N + i] = best;
        arg[(long)b * N + i] = bj;
    }
}

__global__ __launch_bounds__(256) void chamfer_bwd_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                           const int32_t *__restrict__ arg,
                                                           const float *__restrict__ g, int N, int M,
                                                           float *__restrict__ gx, float *__restrict__ gy) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const long xi = ((long)b * N + i) * 3;
    const int a = arg[(long)b * N + i];
    const long ya = ((long)b * M + a) * 3;
    const float s = 2.0f * g[(long)b * N + i];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float v = s * (x[xi + d] - y[ya + d]);
        atomicAdd(gx + xi + d, v);
        atomicAdd(gy + ya + d, -v);
    }
}

}  // namespace

extern "C" int fsg_chamfer_nn_f32(const float *x, const float *y, int B, int N, int M, float *dist, int32_t *arg,
                                  fsg_stream_t stream) {
    FSG_REQUIRE(x && y && dist && arg, "fsg_chamfer_nn_f32: NULL pointer");
    FSG_REQUIRE(B >= 0 && N > 0 && M > 0 && B <= 65535, "fsg_chamfer_nn_f32: bad shape B=%d N=%d M=%d", B, N, M);
    if (B == 0) return FSG_OK;
    hipLaunchKernelGGL(chamfer_nn_kernel, dim3(fsg_cdiv(N, BLOCK), B), dim3(BLOCK), 0, (hipStream_t)stream, x, y, N, M,
                       dist, arg);
    FSG_CHECK_LAUNCH("fsg_chamfer_nn_f32");
    return FSG_OK;
}

extern "C" int fsg_chamfer_nn_bwd_f32(const float *x, const float *y, const int32_t *arg, const float *g_dist, int B,
                                      int N, int M, float *grad_x, float *grad_y, fsg_stream_t stream) {
    FSG_REQUIRE(x && y && arg && g_dist && grad_x && grad_y, "fsg_chamfer_nn_bwd_f32: NULL pointer");
    FSG_REQUIRE(B >= 0 && N > 0 && M > 0 && B <= 65535, "fsg_chamfer_nn_bwd_f32: bad shape");
    if (B == 0) return FSG_OK;
    hipLaunchKernelGGL(chamfer_bwd_kernel, dim3(fsg_cdiv(N, 256), B), dim3(256), 0, (hipStream_t)stream, x, y, arg,
                       g_dist, N, M, grad_x, grad_y);
    FSG_CHECK_LAUNCH("fsg_chamfer_nn_bwd_f32");
    return FSG_OK;
}

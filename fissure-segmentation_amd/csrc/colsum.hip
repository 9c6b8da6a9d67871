// Column sums of a narrow row-major matrix -- include/fsg_hip.h: fsg_colsum_narrow_f32.
//
// The bias gradient of the last point-wise layer (models/dgcnn.py:146: Conv1d(128, num_classes) with bias) is the sum of a
// (B*N, num_classes) matrix over its rows.  ATen's reduction over dim 0 of a (16384, 4) tensor takes 17 us (one thread
// block per 4 columns walking 16 K strided rows); one 1024-thread workgroup reading the matrix as a flat coalesced stream
// does it in ~5 us: with C a power of two <= 32 a thread always meets the same column (1024 % C == 0), so it sums its
// stride-1024 elements in order and the columns are then folded through LDS in a fixed tree -- reproducible.
#include "fsg_common.h"

namespace {

__global__ __launch_bounds__(1024) void colsum_narrow_kernel(const float *__restrict__ x, long total, int C,
                                                             float *__restrict__ out) {
    __shared__ float red[1024];
    const int tid = threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // four independent chains: the loads of a thread pipeline
    long e = tid;
    for (; e + 3 * 1024 < total; e += 4 * 1024) {
        a0 += x[e];
        a1 += x[e + 1024];
        a2 += x[e + 2 * 1024];
        a3 += x[e + 3 * 1024];
    }
    for (; e < total; e += 1024) a0 += x[e];
    red[tid] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    for (int s = 512; s >= C; s >>= 1) {   // tid and tid + s hold the same column (s is a multiple of C)
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid < C) out[tid] = red[tid];
}

}  // namespace

extern "C" int fsg_colsum_narrow_f32(const float *x, int64_t M, int C, float *out, fsg_stream_t stream) {
    FSG_REQUIRE(M >= 0 && C >= 1 && C <= 32 && (C & (C - 1)) == 0, "fsg_colsum_narrow_f32: C=%d must be a power of two <= 32", C);
    FSG_REQUIRE(out && (x || M == 0), "fsg_colsum_narrow_f32: NULL pointer");
    hipLaunchKernelGGL(colsum_narrow_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, (long)M * C, C, out);
    FSG_CHECK_LAUNCH("fsg_colsum_narrow_f32");
    return FSG_OK;
}

// Micro-benchmark: what limits the distance phase of the kNN kernel?  One 1024-thread workgroup per CU (16 waves, like the
// production kernel), every wave runs T "tiles" of 32 v_mfma_f32_16x16x4_f32 in two alternating accumulator chains.
//   mode 0: both operands in registers                      -> the matrix pipe alone
//   mode 1: A operand re-read from LDS for every MFMA (QAL) -> + LDS operand traffic
//   mode 2: mode 1 + the epilogue (8 fma/add + 8 ds_write_b32 per tile)
//   mode 3: mode 2 + B operand from global memory, 16 x 16-byte loads per 4 tiles (the wide loads of the kernel)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate tools/micro/mfma_rate.hip ; run: ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int KS = 16;
template <int MODE>
__global__ __launch_bounds__(1024) void k(const float *__restrict__ x, float *__restrict__ out, int T, int N) {
    extern __shared__ float lds[];
    float *qal = lds;                    // [2][64 * KS]
    float *rows = lds + 2 * 64 * KS;     // [32][1028]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    for (int t = tid; t < 2 * 64 * KS; t += 1024) qal[t] = 0.001f * (t & 31);
    __syncthreads();
    float a0[KS], a1[KS], b[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { a0[s] = 0.01f * (s + lane); a1[s] = 0.02f * (s + lane); b[s] = 0.003f * (lane - s); }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, N * 64 * 4, 0x00020000);
    float sink = 0.f;
    for (int t = 0; t < T; t += 4) {
        f32x4 b4[KS];
        if (MODE >= 3) {
            const unsigned col = ((blockIdx.x * 7 + t * 16 + wave * 64) % (N - 64)) + 4 * l15;
#pragma unroll
            for (int s = 0; s < KS; ++s)
                b4[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((unsigned)l4 * N + col) * 4u, (unsigned)(16 * s) * N, 0));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
            int qoff = l4 * 16 + l15;
            asm volatile("" : "+v"(qoff));
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float bb = MODE >= 3 ? b4[s][u] : b[s];
                if (MODE >= 1) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qal[64 * s + qoff], bb, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qal[64 * KS + 64 * s + qoff], bb, acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], bb, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], bb, acc1, 0, 0, 0);
                }
            }
            if (MODE >= 2) {
                float *dst = rows + (l4 * 4) * 1028 + ((wave * 64 + 4 * l15 + u) & 1023);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dst[e * 1028] = (1.5f - 2.0f * acc0[e]) + 0.25f;
                    dst[(16 + e) * 1028] = (1.5f - 2.0f * acc1[e]) + 0.25f;
                }
            } else {
                sink += acc0[0] + acc1[1];
            }
        }
    }
    __syncthreads();
    if (MODE >= 2) sink = rows[(tid * 37) % (32 * 1028)];
    out[blockIdx.x * 1024 + tid] = sink;
}
template <int MODE> void run(const float *x, float *out, int T, int N) {
    const size_t lds = sizeof(float) * (2 * 64 * KS + 32 * 1028);
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), lds, 0, x, out, T, N);
    hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), lds, 0, x, out, T, N);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double flop = 256.0 * 16 * T * 32 * (2.0 * 16 * 16 * 4);
    printf("mode %d: %.1f us, %.1f TFLOP/s (%.2f of 157.3)\n", MODE, ms * 1e3, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3);
}
int main() {
    const int N = 2048, T = 64;
    float *x, *out; hipMalloc(&x, sizeof(float) * N * 64 * 8); hipMemset(x, 0, sizeof(float) * N * 64 * 8); hipMalloc(&out, sizeof(float) * 256 * 1024);
    run<0>(x, out, T, N); run<1>(x, out, T, N); run<2>(x, out, T, N); run<3>(x, out, T, N);
    return 0;
}

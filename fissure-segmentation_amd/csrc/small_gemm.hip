// Small / skinny fp32 GEMM on the matrix cores -- include/fsg_hip.h: fsg_gemm_small_f32.
//
// The point-wise Linears of the PointTransformer path (models/pointtransformer/seg_model.py: linear_q/k/v, linear1/3,
// TransitionDown/Up) are tiny: at the coarse levels the whole product is one 256 x 256 x 256 block, and its weight
// gradient dW = dY^T X has a few thousand outputs behind a reduction over all points.  Whenever BOTH output dimensions
// are <= 256 the vendor library answers with ONE 256x256 macro-tile = one workgroup = one of 256 CUs, so its time grows
// with I*J*K (tools/probe_vendor_gemm.py on MI355X: 22 us at 256x256x64, 63 us at x256, 173 us at x768; 4-6 us as soon
// as one dimension exceeds 256) -- 3.6 ms of a 15 ms PointTransformer step.  This kernel tiles the output 64 x 64 per workgroup (four waves, one 32 x 32 v_mfma_f32_32x32x2_f32
// accumulator each), stages both operands through LDS with whatever orientation is contiguous in memory (generic
// element strides: X W^T, dY W and dY^T X are the same kernel), and splits the reduction dimension over blockIdx.z
// when the output alone cannot fill the chip; the partial products are then summed in split order by a second
// kernel (no atomics: reproducible).  MFMA fp32 is an exact fp32 fma chain, so results match a plain fp32 GEMM up to
// summation order.
#include <stdlib.h>

#include "fsg_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TI = 64, TJ = 64, TK = 32, LDT = 65;  // k-major LDS tiles [TK][64 + 1]
constexpr int LPT = TI * TK / 256;                  // tile elements per thread and operand

__global__ __launch_bounds__(256) void gemm_small_kernel(const float *__restrict__ A, long sai, long sak,
                                                         const float *__restrict__ B, long sbk, long sbj,
                                                         const float *__restrict__ bias, float *__restrict__ C, long ldc,
                                                         int I, int J, int K, int kchunk, float *__restrict__ part,
                                                         float *__restrict__ rowsum) {
    __shared__ float As[TK * LDT], Bs[TK * LDT];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, ql = lane & 31, half = lane >> 5;
    const int i0 = blockIdx.x * TI, j0 = blockIdx.y * TJ;
    const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;  // this wave's 32 x 32 quadrant
    // loader mapping: consecutive threads along whichever tile dimension is contiguous in memory
    const bool a_kfast = sak == 1, b_kfast = sbk == 1;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // software pipeline: the global loads of tile t+1 are in flight while tile t runs on the matrix cores
    float av[LPT], bv[LPT];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int r = 0; r < LPT; ++r) {
            const int t = r * 256 + tid;
            const int ai = a_kfast ? t / TK : t % TI, ak = a_kfast ? t % TK : t / TI;
            const int bj = b_kfast ? t / TK : t % TJ, bk = b_kfast ? t % TK : t / TJ;
            av[r] = (i0 + ai < I && k0 + ak < kend) ? A[(long)(i0 + ai) * sai + (long)(k0 + ak) * sak] : 0.f;
            bv[r] = (j0 + bj < J && k0 + bk < kend) ? B[(long)(k0 + bk) * sbk + (long)(j0 + bj) * sbj] : 0.f;
        }
    };
    // optional by-product: rowsum[i] = sum_k A(i, k) -- with A = dY^T this is the bias gradient of the Linear whose weight
    // gradient the product is (one ATen column reduction less per Linear).  The first column of workgroups sums its A tiles as
    // they pass through LDS: thread i < 64 walks the tile's 32 k-rows of row i; k order is fixed, so it is reproducible.
    const bool do_rs = rowsum != nullptr && blockIdx.y == 0 && tid < TI;
    float rs = 0.f;
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
        __syncthreads();  // previous tile consumed
#pragma unroll
        for (int r = 0; r < LPT; ++r) {
            const int t = r * 256 + tid;
            const int ai = a_kfast ? t / TK : t % TI, ak = a_kfast ? t % TK : t / TI;
            const int bj = b_kfast ? t / TK : t % TJ, bk = b_kfast ? t % TK : t / TJ;
            As[ak * LDT + ai] = av[r];
            Bs[bk * LDT + bj] = bv[r];
        }
        __syncthreads();
        if (k0 + TK < kend) fetch(k0 + TK);
        if (do_rs) {
#pragma unroll
            for (int kk = 0; kk < TK; ++kk) rs += As[kk * LDT + tid];     // rows / k beyond the matrix were staged as zeros
        }
#pragma unroll
        for (int s = 0; s < TK / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(2 * s + half) * LDT + wi + ql], Bs[(2 * s + half) * LDT + wj + ql],
                                                       acc, 0, 0, 0);
    }
    if (do_rs && i0 + tid < I) {
        // partial sums of the splits behind the partial products: [S][I] at part + S * I * J (gridDim.z = S)
        if (part) part[(long)gridDim.z * I * J + (long)blockIdx.z * I + i0 + tid] = rs;
        else rowsum[i0 + tid] = rs;
    }
    const int col = j0 + wj + ql;
    if (col >= J) return;
    if (part) {
        float *dst = part + (long)blockIdx.z * I * J;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = i0 + wi + (e & 3) + 8 * (e >> 2) + 4 * half;
            if (row < I) dst[(long)row * J + col] = acc[e];
        }
    } else {
        const float bb = bias ? bias[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = i0 + wi + (e & 3) + 8 * (e >> 2) + 4 * half;
            if (row < I) C[(long)row * ldc + col] = acc[e] + bb;
        }
    }
}

// bf16 operand mode (BASELINE config 3 names bf16): the same tiling, split and epilogue, but both operands are rounded to bf16
// (round to nearest even, as torch's .bfloat16()) on their way into LDS and the products run on v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation -- two matrix instructions per 32-deep tile and wave instead of sixteen.  LDS tiles are row-major [64][32 + 8]
// bf16 (a fragment = eight consecutive k of one row = one 16-byte read).  The row-sum by-product sums the ROUNDED values.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
constexpr int LDH = TK + 8;   // bf16 row stride (80 bytes: 16-byte aligned fragments, rows spread over the banks)
__device__ __forceinline__ u16 to_bf16(float x) {
    const __bf16 h = (__bf16)x;
    return __builtin_bit_cast(u16, h);
}
__global__ __launch_bounds__(256) void gemm_small_bf16_kernel(const float *__restrict__ A, long sai, long sak,
                                                              const float *__restrict__ B, long sbk, long sbj,
                                                              const float *__restrict__ bias, float *__restrict__ C, long ldc,
                                                              int I, int J, int K, int kchunk, float *__restrict__ part,
                                                              float *__restrict__ rowsum) {
    __shared__ __attribute__((aligned(16))) u16 As[TI * LDH], Bs[TJ * LDH];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, ql = lane & 31, half = lane >> 5;
    const int i0 = blockIdx.x * TI, j0 = blockIdx.y * TJ;
    const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    const bool a_kfast = sak == 1, b_kfast = sbk == 1;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float av[LPT], bv[LPT];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int r = 0; r < LPT; ++r) {
            const int t = r * 256 + tid;
            const int ai = a_kfast ? t / TK : t % TI, ak = a_kfast ? t % TK : t / TI;
            const int bj = b_kfast ? t / TK : t % TJ, bk = b_kfast ? t % TK : t / TJ;
            av[r] = (i0 + ai < I && k0 + ak < kend) ? A[(long)(i0 + ai) * sai + (long)(k0 + ak) * sak] : 0.f;
            bv[r] = (j0 + bj < J && k0 + bk < kend) ? B[(long)(k0 + bk) * sbk + (long)(j0 + bj) * sbj] : 0.f;
        }
    };
    const bool do_rs = rowsum != nullptr && blockIdx.y == 0 && tid < TI;
    float rs = 0.f;
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < LPT; ++r) {
            const int t = r * 256 + tid;
            const int ai = a_kfast ? t / TK : t % TI, ak = a_kfast ? t % TK : t / TI;
            const int bj = b_kfast ? t / TK : t % TJ, bk = b_kfast ? t % TK : t / TJ;
            As[ai * LDH + ak] = to_bf16(av[r]);
            Bs[bj * LDH + bk] = to_bf16(bv[r]);
        }
        __syncthreads();
        if (k0 + TK < kend) fetch(k0 + TK);
        if (do_rs) {
#pragma unroll
            for (int kk = 0; kk < TK; ++kk) rs += __uint_as_float((unsigned)As[tid * LDH + kk] << 16);
        }
#pragma unroll
        for (int s = 0; s < TK / 16; ++s) {
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(&As[(wi + ql) * LDH + 16 * s + 8 * half]);
            const bf16x8 b = *reinterpret_cast<const bf16x8 *>(&Bs[(wj + ql) * LDH + 16 * s + 8 * half]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    }
    if (do_rs && i0 + tid < I) {
        if (part) part[(long)gridDim.z * I * J + (long)blockIdx.z * I + i0 + tid] = rs;
        else rowsum[i0 + tid] = rs;
    }
    const int col = j0 + wj + ql;
    if (col >= J) return;
    if (part) {
        float *dst = part + (long)blockIdx.z * I * J;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = i0 + wi + (e & 3) + 8 * (e >> 2) + 4 * half;
            if (row < I) dst[(long)row * J + col] = acc[e];
        }
    } else {
        const float bb = bias ? bias[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = i0 + wi + (e & 3) + 8 * (e >> 2) + 4 * half;
            if (row < I) C[(long)row * ldc + col] = acc[e] + bb;
        }
    }
}

// partial products [S][I*J] -> C: 16 outputs x 16 slices of the split range per workgroup (the slices of one output are
// combined in slice order through LDS: fixed order, reproducible)
__global__ __launch_bounds__(256) void gemm_small_reduce_flat_kernel(const float *__restrict__ part, int S, long IJ, int J,
                                                                     const float *__restrict__ bias, float *__restrict__ C,
                                                                     long ldc, int I, float *__restrict__ rowsum) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;   // few splits: one thread per output
    if (rowsum && t >= IJ && t < IJ + I) {                 // the row sums' partials sit behind the products': [S][I]
        float a = 0.f;
        for (int s = 0; s < S; ++s) a += part[(long)S * IJ + (long)s * I + (t - IJ)];
        rowsum[t - IJ] = a;
    }
    if (t >= IJ) return;
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += part[(long)s * IJ + t];
    const long row = t / J;
    const int col = (int)(t - row * J);
    C[row * ldc + col] = a + (bias ? bias[col] : 0.f);
}

__global__ __launch_bounds__(256) void gemm_small_reduce_kernel(const float *__restrict__ part, int S, long IJ, int J,
                                                                const float *__restrict__ bias, float *__restrict__ C,
                                                                long ldc, int I, float *__restrict__ rowsum) {
    __shared__ float red[16][17];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long t = (long)blockIdx.x * 16 + o;
    const long nb = (IJ + 15) / 16;                        // workgroups of the products; the rest fold the row sums
    if (rowsum && (long)blockIdx.x >= nb) {                // (their partials sit behind the products': [S][I])
        const long ri = ((long)blockIdx.x - nb) * 16 + o;
        float a = 0.f;
        if (ri < I)
            for (int s = sl; s < S; s += 16) a += part[(long)S * IJ + (long)s * I + ri];
        red[sl][o] = a;
        __syncthreads();
        if (sl == 0 && ri < I) {
            float r = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) r += red[i][o];
            rowsum[ri] = r;
        }
        return;
    }
    float a = 0.f;
    if (t < IJ)
        for (int s = sl; s < S; s += 16) a += part[(long)s * IJ + t];
    red[sl][o] = a;
    __syncthreads();
    if (sl == 0 && t < IJ) {
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) r += red[i][o];
        const long row = t / J;
        const int col = (int)(t - row * J);
        C[row * ldc + col] = r + (bias ? bias[col] : 0.f);
    }
}

// the reductions of up to FSG_GEMM_REDUCE_MAX_JOBS deferred products in ONE launch (fsg_gemm_small_reduce_many_f32): the weight
// gradients of a backward pass are read by nobody until the optimizer runs, so their split sums need not sit between the
// products on the stream.  One thread per output, splits summed in split order (fixed: reproducible).
__global__ __launch_bounds__(256) void gemm_small_reduce_many_kernel(fsg_gemm_reduce_jobs jobs) {
    int j = 0;
    long b = blockIdx.x;
    while (j < jobs.n - 1 && b >= jobs.blocks[j]) { b -= jobs.blocks[j]; ++j; }
    const float *part = jobs.part[j];
    const int S = jobs.S[j], I = jobs.I[j], J = jobs.J[j];
    const long IJ = (long)I * J, t = b * 256 + threadIdx.x;
    float *rowsum = jobs.rowsum[j];
    if (rowsum && t >= IJ && t < IJ + I) {
        float a = 0.f;
        for (int s = 0; s < S; ++s) a += part[(long)S * IJ + (long)s * I + (t - IJ)];
        rowsum[t - IJ] = a;
    }
    if (t >= IJ) return;
    float a = 0.f;
#pragma unroll 8
    for (int s = 0; s < S; ++s) a += part[(long)s * IJ + t];
    const long row = t / J;
    const int col = (int)(t - row * J);
    jobs.C[j][row * jobs.ldc[j] + col] = a;
}

// number of reduction splits: several workgroups per CU (the kernel is latency-bound: co-resident workgroups hide
// the operand loads), at least FSG_GEMM_SPLIT_MIN (default 64) reduction steps each (measured on the PointTransformer step: 32: 8.9, 64: 8.9, 128: 9.05, 256: 9.5 ms)
inline int splits_for(int I, int J, int K) {
    static const int min_k = getenv("FSG_GEMM_SPLIT_MIN") ? atoi(getenv("FSG_GEMM_SPLIT_MIN")) : 64;
    const long tiles = (long)fsg_cdiv(I, TI) * fsg_cdiv(J, TJ);
    if (tiles >= 256 || K < 2 * min_k) return 1;
    long s = (1024 + tiles - 1) / tiles;
    const long smax = K / min_k;
    if (s > smax) s = smax;
    if (s > 128) s = 128;
    return s < 1 ? 1 : (int)s;
}

}  // namespace

extern "C" size_t fsg_gemm_small_workspace_bytes(int I, int J, int K) {
    if (I <= 0 || J <= 0 || K <= 0) return 0;
    const int s = splits_for(I, J, K);
    return s > 1 ? sizeof(float) * (size_t)s * ((size_t)I * (size_t)J + (size_t)I) : 0;      // products + row sums of every split
}

extern "C" int fsg_gemm_small_rowsum_f32(const float *A, int64_t sa_i, int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_j,
                                         const float *bias, float *C, int64_t ldc, int I, int J, int K, float *rowsum,
                                         void *workspace, fsg_stream_t stream);

extern "C" int fsg_gemm_small_f32(const float *A, int64_t sa_i, int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_j,
                                  const float *bias, float *C, int64_t ldc, int I, int J, int K, void *workspace,
                                  fsg_stream_t stream) {
    return fsg_gemm_small_rowsum_f32(A, sa_i, sa_k, B, sb_k, sb_j, bias, C, ldc, I, J, K, nullptr, workspace, stream);
}

namespace {
int gemm_small_launch(const float *A, int64_t sa_i, int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_j, const float *bias,
                      float *C, int64_t ldc, int I, int J, int K, float *rowsum, void *workspace, int *deferred_splits,
                      fsg_stream_t stream, bool bf16 = false);
}

extern "C" int fsg_gemm_small_rowsum_f32(const float *A, int64_t sa_i, int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_j,
                                         const float *bias, float *C, int64_t ldc, int I, int J, int K, float *rowsum,
                                         void *workspace, fsg_stream_t stream) {
    return gemm_small_launch(A, sa_i, sa_k, B, sb_k, sb_j, bias, C, ldc, I, J, K, rowsum, workspace, nullptr, stream);
}

// The product WITHOUT its split reduction: *splits = S > 1 when the partial products [S][I*J] (+ row-sum partials [S][I]) were
// left in `workspace` for fsg_gemm_small_reduce_many_f32 (the caller keeps the workspace alive until then), 0 when the shape
// needed no split and C (and rowsum) are final.  No bias (a deferred product is a gradient).
extern "C" int fsg_gemm_small_deferred_f32(const float *A, int64_t sa_i, int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_j,
                                           float *C, int64_t ldc, int I, int J, int K, float *rowsum, void *workspace, int *splits,
                                           fsg_stream_t stream) {
    FSG_REQUIRE(splits, "fsg_gemm_small_deferred_f32: NULL splits");
    return gemm_small_launch(A, sa_i, sa_k, B, sb_k, sb_j, nullptr, C, ldc, I, J, K, rowsum, workspace, splits, stream);
}

// bf16 operand mode of the two entry points above (operands rounded to bf16 in the kernel, fp32 accumulation and output);
// `splits` NULL: reduce at once (bias allowed), else deferred as in fsg_gemm_small_deferred_f32 (bias must be NULL)
extern "C" int fsg_gemm_small_bf16(const float *A, int64_t sa_i, int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_j,
                                   const float *bias, float *C, int64_t ldc, int I, int J, int K, float *rowsum, void *workspace,
                                   int *splits, fsg_stream_t stream) {
    FSG_REQUIRE(!(splits && bias), "fsg_gemm_small_bf16: a deferred product takes no bias");
    return gemm_small_launch(A, sa_i, sa_k, B, sb_k, sb_j, bias, C, ldc, I, J, K, rowsum, workspace, splits, stream, true);
}

extern "C" int fsg_gemm_small_reduce_many_f32(const fsg_gemm_reduce_jobs *jobs, fsg_stream_t stream) {
    FSG_REQUIRE(jobs && jobs->n >= 1 && jobs->n <= FSG_GEMM_REDUCE_MAX_JOBS, "fsg_gemm_small_reduce_many_f32: bad job count");
    fsg_gemm_reduce_jobs j = *jobs;
    long total = 0;
    for (int i = 0; i < j.n; ++i) {
        FSG_REQUIRE(j.part[i] && j.C[i] && j.S[i] >= 1 && j.I[i] > 0 && j.J[i] > 0 && j.ldc[i] >= j.J[i],
                    "fsg_gemm_small_reduce_many_f32: bad job %d", i);
        j.blocks[i] = (int)fsg_cdiv((long)j.I[i] * j.J[i] + (j.rowsum[i] ? j.I[i] : 0), 256L);
        total += j.blocks[i];
    }
    FSG_REQUIRE(total < (1L << 31), "fsg_gemm_small_reduce_many_f32: too many outputs");
    hipLaunchKernelGGL(gemm_small_reduce_many_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, j);
    FSG_CHECK_LAUNCH("fsg_gemm_small_reduce_many_f32");
    return FSG_OK;
}

namespace {
int gemm_small_launch(const float *A, int64_t sa_i, int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_j, const float *bias,
                      float *C, int64_t ldc, int I, int J, int K, float *rowsum, void *workspace, int *deferred_splits,
                      fsg_stream_t stream, bool bf16) {
    FSG_REQUIRE(A && B && C, "fsg_gemm_small_f32: NULL pointer");
    FSG_REQUIRE(I > 0 && J > 0 && K > 0 && ldc >= J, "fsg_gemm_small_f32: bad shape I=%d J=%d K=%d ldc=%ld", I, J, K, (long)ldc);
    FSG_REQUIRE(fsg_cdiv(J, TJ) <= 65535, "fsg_gemm_small_f32: J too large");
    const int S = splits_for(I, J, K);
    FSG_REQUIRE(S == 1 || workspace, "fsg_gemm_small_f32: this shape needs the workspace");
    hipStream_t st = (hipStream_t)stream;
    int kchunk = fsg_cdiv(fsg_cdiv(K, S), TK) * TK;
    const int S_eff = fsg_cdiv(K, kchunk);  // every split non-empty
    float *part = S_eff > 1 ? (float *)workspace : nullptr;
    if (bf16)
        hipLaunchKernelGGL(gemm_small_bf16_kernel, dim3(fsg_cdiv(I, TI), fsg_cdiv(J, TJ), S_eff), dim3(256), 0, st, A, (long)sa_i,
                           (long)sa_k, B, (long)sb_k, (long)sb_j, bias, C, (long)ldc, I, J, K, kchunk, part, rowsum);
    else
        hipLaunchKernelGGL(gemm_small_kernel, dim3(fsg_cdiv(I, TI), fsg_cdiv(J, TJ), S_eff), dim3(256), 0, st, A, (long)sa_i,
                           (long)sa_k, B, (long)sb_k, (long)sb_j, bias, C, (long)ldc, I, J, K, kchunk, part, rowsum);
    FSG_CHECK_LAUNCH("fsg_gemm_small_f32");
    if (deferred_splits) {
        *deferred_splits = part ? S_eff : 0;
        return FSG_OK;
    }
    if (part) {
        const long IJ = (long)I * J, outs = IJ + (rowsum ? I : 0);
        if (S_eff >= 16)
            hipLaunchKernelGGL(gemm_small_reduce_kernel, dim3(fsg_cdiv(IJ, 16) + (rowsum ? fsg_cdiv(I, 16) : 0)), dim3(256), 0, st, part,
                               S_eff, IJ, J, bias, C, (long)ldc, I, rowsum);
        else
            hipLaunchKernelGGL(gemm_small_reduce_flat_kernel, dim3(fsg_cdiv(outs, 256)), dim3(256), 0, st, part, S_eff, IJ, J,
                               bias, C, (long)ldc, I, rowsum);
        FSG_CHECK_LAUNCH("fsg_gemm_small_f32/reduce");
    }
    return FSG_OK;
}
}  // namespace

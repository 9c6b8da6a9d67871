// Point-wise ("shared fully connected") layers of the DGCNN head on the bf16 matrix pipe with fp32-grade results --
// include/fsg_hip.h: fsg_pw_*.  Replaces the vendor fp32 GEMMs + separate BatchNorm passes behind
// models/dgcnn.py:123-137,156-160 (global feature 192 -> 1024 + max over the points, segmentation head
// 1216 -> 256 -> 256 -> 128 -> classes; every block = 1x1 Conv1d -> BatchNorm1d (train: batch statistics) -> LeakyReLU(0.2),
// models/dgcnn.py:282-323).
//
// Arithmetic.  gfx950 has no TF32/xf32 path and its exact-fp32 MFMA runs at 1/16 of the bf16 rate (157 vs 2500 TFLOP/s).
// Every fp32 operand x is split into three bf16 pieces x = h + m + l (round-to-nearest-even, residuals exact in fp32:
// |m| <= 2^-9 |x|, |l| <= 2^-18 |x|, remainder <= 2^-27 |x|) and a product a.b is the six bf16 MFMA products
//     ah.bh + ah.bm + am.bh + am.bm + ah.bl + al.bh        (dropped: am.bl + al.bm + al.bl <= 2^-26 |a||b|)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped terms are a quarter of ONE fp32 rounding of the product, so
// the result is as close to real arithmetic as an fp32 fma chain (tests compare both with fp64), at 16/6 = 2.7x the rate of
// the fp32 matrix instruction.  Weights are split once per call by pw_weight_image_kernel into the REGISTER IMAGE of the
// MFMA's B operand (1-KiB blocks: 32 output columns x 16 k, one per piece), activations are split on their way from HBM to
// LDS, where the prologue (BatchNorm + LeakyReLU of the previous layer, or the BatchNorm backward formula) is applied too.
//
// pw_rowgemm_kernel:  C (M, N) = pro(A) (M, K) . B (K, N)
//   tile 64 WM x 64 WN per workgroup of four waves (2 x 2, each WM x WN MFMA tiles of 32 x 32), BK = 32 per iteration,
//   A through registers (prefetched one iteration ahead) -> prologue -> split -> LDS fragment image (XOR-swizzled so that the
//   row-major writers and the lane-linear readers are both conflict-free), B image copied 16 bytes per lane.
//   Epilogues on the accumulator tile: store, per-column (n, mean, M2) records of the tile's rows (train-mode BatchNorm
//   statistics, merged in fp64 by the finalize kernels), per-column max of sgn * c with its row (the global max-pool over a
//   cloud through the monotone BatchNorm + LeakyReLU), partial sums of the BatchNorm backward (h, h * yhat).
#include <type_traits>

#include "fsg_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { PW_STORE = 1, PW_STATS = 2, PW_SEL = 4, PW_BWDSTATS = 8, PW_BIAS = 16 };
enum { PRO_NONE = 0, PRO_BNACT = 1, PRO_BNBWD = 2 };

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {   // (a, b) -> packed bf16 pair, a in the low half, RNE
    bf16x2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}

// eight fp32 values -> three bf16 pieces each (h, m, l as packed operand fragments)
__device__ __forceinline__ void split8(const float (&x)[8], u32x4 &h, u32x4 &m, u32x4 &l) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = x[2 * q], b = x[2 * q + 1];
        const unsigned ph = pk_bf16(a, b);
        const float ra = a - __uint_as_float(ph << 16), rb = b - __uint_as_float(ph & 0xffff0000u);
        const unsigned pm = pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(pm << 16), sb = rb - __uint_as_float(pm & 0xffff0000u);
        h[q] = ph;
        m[q] = pm;
        l[q] = pk_bf16(sa, sb);
    }
}

// B operand image of a (N, K) matrix W(n, k) = W[n * sn + k * sk] * scale:  blocks [nb][ks][piece][lane] of 16 bytes, lane
// (n = 32 nb + (lane & 31), k = 16 ks + 8 (lane >> 5) + 0..7); rows >= N and columns >= K read as zero.  `ks0`/`KS`: the
// image may be the concatenation of several matrices along k (this call fills k-steps ks0 .. ks0 + ceil(K/16) - 1 of KS).
__global__ __launch_bounds__(256) void pw_weight_image_kernel(const float *__restrict__ W, long sn, long sk, int N, int K,
                                                              float scale, int ks0, int KS, int pieces, u32x4 *__restrict__ img) {
    const int ksteps = (K + 15) / 16;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = (int)(t & 63);
    const long blk = t >> 6;
    const int ks = (int)(blk % ksteps), nb = (int)(blk / ksteps);
    if (nb * 32 >= ((N + 31) & ~31)) return;
    const int n = nb * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (n < N && k0 + j < K) ? W[(long)n * sn + (long)(k0 + j) * sk] * scale : 0.f;
    u32x4 h, m, l;
    split8(x, h, m, l);
    u32x4 *o = img + (((long)nb * KS + ks0 + ks) * pieces) * 64 + lane;
    o[0] = h;                          // pieces == 1: the bf16 operand mode keeps the leading piece only
    if (pieces == 3) {
        o[64] = m;
        o[128] = l;
    }
}

// several images in one launch (the weights of a whole head, both orientations): job j owns the 64-thread blocks
// [first[j], first[j + 1])
struct ImageJobs {
    const float *W[FSG_PW_MAX_IMAGE_JOBS];
    long sn[FSG_PW_MAX_IMAGE_JOBS], sk[FSG_PW_MAX_IMAGE_JOBS];
    int N[FSG_PW_MAX_IMAGE_JOBS], K[FSG_PW_MAX_IMAGE_JOBS], ks0[FSG_PW_MAX_IMAGE_JOBS], KS[FSG_PW_MAX_IMAGE_JOBS];
    float scale[FSG_PW_MAX_IMAGE_JOBS];
    u32x4 *img[FSG_PW_MAX_IMAGE_JOBS];
    long first[FSG_PW_MAX_IMAGE_JOBS + 1];
    int n;
};

__global__ __launch_bounds__(256) void pw_weight_images_kernel(const ImageJobs jobs) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = (int)(t & 63);
    const long blk = t >> 6;
    int j = 0;
#pragma unroll
    for (int q = 1; q < FSG_PW_MAX_IMAGE_JOBS; ++q)
        if (q < jobs.n && blk >= jobs.first[q]) j = q;
    if (blk >= jobs.first[jobs.n]) return;
    const long lb = blk - jobs.first[j];
    const int N = jobs.N[j], K = jobs.K[j], ksteps = (K + 15) / 16;
    const int ks = (int)(lb % ksteps), nb = (int)(lb / ksteps);
    const int n = nb * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
    const float *W = jobs.W[j];
    const long sn = jobs.sn[j], sk = jobs.sk[j];
    const float scale = jobs.scale[j];
    float x[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = (n < N && k0 + q < K) ? W[(long)n * sn + (long)(k0 + q) * sk] * scale : 0.f;
    u32x4 h, m, l;
    split8(x, h, m, l);
    u32x4 *o = jobs.img[j] + (((long)nb * jobs.KS[j] + jobs.ks0[j] + ks) * 3) * 64 + lane;
    o[0] = h;
    o[64] = m;
    o[128] = l;
}

struct RowGemmArgs {
    const float *A1;      // segment 1 of the A operand: (M, K1) rows, row stride lda1; PRO_BNBWD: the upstream gradient `da`
    const float *Y1;      // PRO_BNBWD: the pre-BatchNorm values y of the same shape / stride
    const float *A2;      // optional plain segment 2: (M, K2), row stride lda2
    long lda1, lda2;
    int K1, K2;
    const u32x4 *Bimg;    // weight image, (ceil(N/32), (K1+K2)/16, 3, 64) x 16 bytes
    int M, N, rows_per_cloud;
    // prologue tables, length K1 (delta / P: one row per cloud with stride tstride, or tstride = 0)
    const float *alpha, *delta, *P, *Q;
    int tstride;
    float slope;
    // epilogue
    float *C;             // columns >= store_n0 go to C[row * ldc + col - store_n0]
    long ldc;
    int store_n0;
    const float *bias;    // PW_BIAS: added per column (length N)
    float *rec;           // PW_STATS: (M / BM, 3, N) records (n, mean, M2) per row block and column
    const float *sgn;     // PW_SEL: columns < sel_n: sel_val[rb][col] = max over the block's rows of sgn[col] * c, sel_arg = its row
    float *sel_val;       //         within the cloud (lowest row on ties); layout (M / BM, sel_n)
    int *sel_arg;
    int sel_n;
    const float *Yp;      // PW_BWDSTATS: pre-BatchNorm values of the layer whose activation gradient this product is, (M, N)
    long ldyp;
    const float *ealpha, *edelta, *emu, *er;   // its tables (edelta, emu = mean - cloud shift: per cloud with stride etstride)
    int etstride;
    float *rec2;          // (M / BM, 2, N): sums of h = c f'(u) and h * yhat over the block's rows
};

template <int WM, int WN, int PRO, int EPI, int PIPE, int NP = 3>
__global__ __launch_bounds__(256) void pw_rowgemm_kernel(const RowGemmArgs p) {
    static_assert(NP == 3 || (NP == 1 && PIPE == 1 && PRO == PRO_NONE), "one-piece (bf16 operand) mode: pipelined plain products only");
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int AI = BM / 64;                       // A rows per thread and iteration
    constexpr int BU = BN * 4 * NP / 256;             // 16-byte units of the B image per thread and iteration
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NBUF = PIPE ? 2 : 1;
    u32x4 *Aimg = reinterpret_cast<u32x4 *>(smem);                            // [NBUF][BM/32][2][3][64]
    u32x4 *Bimg = Aimg + NBUF * (BM / 32) * 2 * NP * 64;                      // [NBUF][BN/32][2][NP][64]
    float *tab = reinterpret_cast<float *>(Bimg + NBUF * (BN / 32) * 2 * NP * 64);  // PRO tables: [4][K1]; epilogue scratch after the loop
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int MT = (p.M + BM - 1) / BM, NT = (p.N + BN - 1) / BN;
    // XCD-aware placement: consecutive workgroup ids go round-robin over the 8 XCDs; all column tiles of one row tile share
    // an XCD (its L2 then serves the A rows to the NT - 1 later tiles)
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int tm = (jj / NT) * 8 + xcd, tn = jj % NT;
    if (tm >= MT) return;
    const int row0 = tm * BM, col0 = tn * BN;
    const int K = p.K1 + p.K2, KT = K / 32, KS = K / 16;
    const int cloud = p.rows_per_cloud > 0 ? row0 / p.rows_per_cloud : 0;

    if (PRO != PRO_NONE) {
        for (int e = tid; e < p.K1; e += 256) {
            tab[e] = p.alpha[e];
            tab[p.K1 + e] = p.delta[(long)cloud * p.tstride + e];
            if (PRO == PRO_BNBWD) {
                tab[2 * p.K1 + e] = p.P[(long)cloud * p.tstride + e];
                tab[3 * p.K1 + e] = p.Q[e];
            }
        }
    }

    const int chunk = tid & 3, arow = tid >> 2;       // this thread's 8-wide k chunk and first row of the A tile
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if constexpr (PIPE) {
        // Software pipeline (one barrier per iteration, both LDS images double-buffered):
        //   iteration `it` runs the MFMAs of k-block it from LDS buffer it & 1 while (a) the B image of block it + 1 streams
        //   global -> LDS by DMA (global_load_lds, no registers, no VALU), (b) the raw A rows of block it + 2 are in flight to
        //   registers and (c) the raw A rows of block it + 1 (loaded an iteration ago) go through prologue + split into the other
        //   LDS buffer -- VALU work the scheduler places between this wave's own MFMAs.
        float4 ra[2][AI][2], ry[2][PRO == PRO_BNBWD ? AI : 1][2];
        constexpr int ABUF = (BM / 32) * 2 * NP * 64, BBUF = (BN / 32) * 2 * NP * 64;  // u32x4 units per buffer
        auto fetchA = [&](auto Pc, int it) {
            constexpr int P = decltype(Pc)::value;
            const int k0 = it * 32;
            const bool seg2 = k0 >= p.K1;
            const float *src = seg2 ? p.A2 + (k0 - p.K1) : p.A1 + k0;
            const long ld = seg2 ? p.lda2 : p.lda1;
    #pragma unroll
            for (int i = 0; i < AI; ++i) {
                // rows behind M (a ragged last tile) re-read row M - 1: their products land in output rows that are never stored (a
                // conditional zero-fill here crashes the ROCm 7.2 backend in Machine Copy Propagation)
                const int row = min(row0 + arow + 64 * i, p.M - 1);
                const float4 *g = reinterpret_cast<const float4 *>(src + (long)row * ld + chunk * 8);
                ra[P][i][0] = g[0];
                ra[P][i][1] = g[1];
                if constexpr (PRO == PRO_BNBWD) {
                    if (!seg2) {
                        const float4 *gy = reinterpret_cast<const float4 *>(p.Y1 + k0 + (long)row * ld + chunk * 8);
                        ry[P][i][0] = gy[0];
                        ry[P][i][1] = gy[1];
                    }
                }
            }
        };
        const int nbl = (p.N + 31) / 32 - col0 / 32;      // 32-column blocks of the image at / behind this tile
        auto dmaB = [&](int it, int buf) {
            const u32x4 *bsrc = p.Bimg + ((long)(col0 / 32) * KS + 2 * it) * (NP * 64);
            #pragma unroll
            for (int q = 0; q < BU; ++q) {
                const int u = tid + 256 * q, jb = min(u / (128 * NP), nbl - 1), rem = u % (128 * NP);   // blocks behind N: any valid one (never stored)
                // LDS destination: wave-uniform base, the hardware adds lane * 16
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bsrc + (long)jb * KS * (NP * 64) + rem),
                                                 (__attribute__((address_space(3))) void *)(Bimg + buf * BBUF + 256 * q + wave * 64), 16, 0, 0);
            }
        };
        auto transformA = [&](auto Pc, int it, int buf) {
            constexpr int P = decltype(Pc)::value;
            const int k0 = it * 32;
            const bool seg2 = k0 >= p.K1;
            const int kk = k0 + chunk * 8;                // column of segment 1 (tables)
    #pragma unroll
            for (int i = 0; i < AI; ++i) {
                float x[8] = {ra[P][i][0].x, ra[P][i][0].y, ra[P][i][0].z, ra[P][i][0].w,
                              ra[P][i][1].x, ra[P][i][1].y, ra[P][i][1].z, ra[P][i][1].w};
                if constexpr (PRO == PRO_BNACT) if (!seg2) {
    #pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float u = __builtin_fmaf(x[j], tab[kk + j], tab[p.K1 + kk + j]);
                        x[j] = u > 0.f ? u : u * p.slope;
                    }
                }
                if constexpr (PRO == PRO_BNBWD) if (!seg2) {
                    const float y[8] = {ry[P][i][0].x, ry[P][i][0].y, ry[P][i][0].z, ry[P][i][0].w,
                                        ry[P][i][1].x, ry[P][i][1].y, ry[P][i][1].z, ry[P][i][1].w};
    #pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float al = tab[kk + j], u = __builtin_fmaf(y[j], al, tab[p.K1 + kk + j]);
                        const float h = x[j] * (u > 0.f ? 1.f : p.slope);
                        // dy = alpha h - P - Q y
                        x[j] = __builtin_fmaf(al, h, -__builtin_fmaf(tab[3 * p.K1 + kk + j], y[j], tab[2 * p.K1 + kk + j]));
                    }
                }
                u32x4 h, m, l;
                split8(x, h, m, l);
                const int r = arow + 64 * i, mt = r >> 5, ks = chunk >> 1;
                const int slot = ((r & 31) + 32 * (chunk & 1)) ^ (4 * chunk);
                u32x4 *dst = Aimg + buf * ABUF + ((mt * 2 + ks) * NP) * 64 + slot;
                dst[0] = h;
                if constexpr (NP == 3) {
                    dst[64] = m;
                    dst[128] = l;
                }
            }
        };
        auto mfma_ks = [&](int ks, int buf, f32x16 (&acc)[WM][WN]) {
            u32x4 af[WM][3], bf[WN][3];
            const int aslot = lane ^ (4 * (2 * ks + (lane >> 5)));
    #pragma unroll
            for (int i = 0; i < WM; ++i)
    #pragma unroll
                for (int q = 0; q < NP; ++q) af[i][q] = Aimg[buf * ABUF + (((wm * WM + i) * 2 + ks) * NP + q) * 64 + aslot];
    #pragma unroll
            for (int j = 0; j < WN; ++j)
    #pragma unroll
                for (int q = 0; q < NP; ++q) bf[j][q] = Bimg[buf * BBUF + (((wn * WN + j) * 2 + ks) * NP + q) * 64 + lane];
    #pragma unroll
            for (int i = 0; i < WM; ++i)
    #pragma unroll
                for (int j = 0; j < WN; ++j) {
                    f32x16 c = acc[i][j];
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, af[i][0]), bh = __builtin_bit_cast(bf16x8, bf[j][0]);
                    if constexpr (NP == 3) {
                        const bf16x8 am = __builtin_bit_cast(bf16x8, af[i][1]), al = __builtin_bit_cast(bf16x8, af[i][2]);
                        const bf16x8 bm = __builtin_bit_cast(bf16x8, bf[j][1]), bl = __builtin_bit_cast(bf16x8, bf[j][2]);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
                    acc[i][j] = c;
                }
        };


        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        fetchA(P0{}, 0);
        dmaB(0, 0);
        if (KT > 1) fetchA(P1{}, 1);
        __syncthreads();                                  // the prologue tables are in LDS
        transformA(P0{}, 0, 0);
        __syncthreads();                                  // A and B images of block 0 are in LDS (the barrier drains the DMA)
        auto step = [&](auto Pc, int it) {
            constexpr int P = decltype(Pc)::value;
            if (it + 1 < KT) dmaB(it + 1, P ^ 1);
            if (it + 2 < KT) fetchA(Pc, it + 2);          // ra[P] is free: block `it` went through transformA an iteration ago
            mfma_ks(0, P, acc);
            if (it + 1 < KT) transformA(std::integral_constant<int, P ^ 1>{}, it + 1, P ^ 1);
            mfma_ks(1, P, acc);
            __syncthreads();
        };
        for (int it = 0; it < KT; it += 2) {
            step(P0{}, it);
            if (it + 1 < KT) step(P1{}, it + 1);
        }

    } else {
        // single LDS buffer, two barriers per iteration, A and B through registers (prefetched one iteration ahead): 48 KB of LDS
        // at 128 x 128 -> two workgroups per CU whose phases overlap -- the better structure for the widest product of the head
        // (16384 x 1280 x 192: 56 us against 64-69 us for the double-buffered loop, which holds one workgroup per CU there)
        float4 ra[AI][2], ry[PRO == PRO_BNBWD ? AI : 1][2];
        u32x4 rb[BU];
        auto fetch = [&](int it) {
            const int k0 = it * 32;
            const bool seg2 = k0 >= p.K1;
            const float *src = seg2 ? p.A2 + (k0 - p.K1) : p.A1 + k0;
            const long ld = seg2 ? p.lda2 : p.lda1;
    #pragma unroll
            for (int i = 0; i < AI; ++i) {
                // rows behind M (a ragged last tile) re-read row M - 1: their products land in output rows that are never stored (a
                // conditional zero-fill here crashes the ROCm 7.2 backend in Machine Copy Propagation)
                const int row = min(row0 + arow + 64 * i, p.M - 1);
                const float4 *g = reinterpret_cast<const float4 *>(src + (long)row * ld + chunk * 8);
                ra[i][0] = g[0];
                ra[i][1] = g[1];
                if constexpr (PRO == PRO_BNBWD) {
                    if (!seg2) {
                        const float4 *gy = reinterpret_cast<const float4 *>(p.Y1 + k0 + (long)row * ld + chunk * 8);
                        ry[i][0] = gy[0];
                        ry[i][1] = gy[1];
                    }
                }
            }
            const u32x4 *bsrc = p.Bimg + ((long)(col0 / 32) * KS + 2 * it) * 192;
            const int nbl = (p.N + 31) / 32 - col0 / 32;      // 32-column blocks of the image at / behind this tile
    #pragma unroll
            for (int q = 0; q < BU; ++q) {
                const int u = tid + 256 * q, jb = min(u / 384, nbl - 1), rem = u % 384;   // blocks behind N: any valid one (their columns are never stored) -- no branch per load
                rb[q] = bsrc[(long)jb * KS * 192 + rem];
            }
        };


        fetch(0);
        for (int it = 0; it < KT; ++it) {
            __syncthreads();   // tables written (first iteration) / the previous iteration's fragments consumed
            {
                const int k0 = it * 32;
                const bool seg2 = k0 >= p.K1;
                const int kk = k0 + chunk * 8;            // column of segment 1 (tables)
    #pragma unroll
                for (int i = 0; i < AI; ++i) {
                    float x[8] = {ra[i][0].x, ra[i][0].y, ra[i][0].z, ra[i][0].w, ra[i][1].x, ra[i][1].y, ra[i][1].z, ra[i][1].w};
                    if constexpr (PRO == PRO_BNACT) if (!seg2) {
    #pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float u = __builtin_fmaf(x[j], tab[kk + j], tab[p.K1 + kk + j]);
                            x[j] = u > 0.f ? u : u * p.slope;
                        }
                    }
                    if constexpr (PRO == PRO_BNBWD) if (!seg2) {
                        const float y[8] = {ry[i][0].x, ry[i][0].y, ry[i][0].z, ry[i][0].w, ry[i][1].x, ry[i][1].y, ry[i][1].z, ry[i][1].w};
    #pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float al = tab[kk + j], u = __builtin_fmaf(y[j], al, tab[p.K1 + kk + j]);
                            const float h = x[j] * (u > 0.f ? 1.f : p.slope);
                            // dy = alpha h - P - Q y
                            x[j] = __builtin_fmaf(al, h, -__builtin_fmaf(tab[3 * p.K1 + kk + j], y[j], tab[2 * p.K1 + kk + j]));
                        }
                    }
                    u32x4 h, m, l;
                    split8(x, h, m, l);
                    const int r = arow + 64 * i, mt = r >> 5, ks = chunk >> 1;
                    const int slot = ((r & 31) + 32 * (chunk & 1)) ^ (4 * chunk);
                    u32x4 *dst = Aimg + ((mt * 2 + ks) * 3) * 64 + slot;
                    dst[0] = h;
                    dst[64] = m;
                    dst[128] = l;
                }
    #pragma unroll
                for (int q = 0; q < BU; ++q) Bimg[tid + 256 * q] = rb[q];
            }
            __syncthreads();
            if (it + 1 < KT) fetch(it + 1);
    #pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 af[WM][3], bf[WN][3];
                const int aslot = lane ^ (4 * (2 * ks + (lane >> 5)));
    #pragma unroll
                for (int i = 0; i < WM; ++i)
    #pragma unroll
                    for (int q = 0; q < 3; ++q) af[i][q] = Aimg[(((wm * WM + i) * 2 + ks) * 3 + q) * 64 + aslot];
    #pragma unroll
                for (int j = 0; j < WN; ++j)
    #pragma unroll
                    for (int q = 0; q < 3; ++q) bf[j][q] = Bimg[(((wn * WN + j) * 2 + ks) * 3 + q) * 64 + lane];
    #pragma unroll
                for (int i = 0; i < WM; ++i)
    #pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, af[i][0]), am = __builtin_bit_cast(bf16x8, af[i][1]),
                                     al = __builtin_bit_cast(bf16x8, af[i][2]);
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, bf[j][0]), bm = __builtin_bit_cast(bf16x8, bf[j][1]),
                                     bl = __builtin_bit_cast(bf16x8, bf[j][2]);
                        f32x16 c = acc[i][j];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
                        acc[i][j] = c;
                    }
            }
        }

    }
    // ---------------------------------------------------------------- epilogue: lane holds column (lane & 31) of each of its
    // tiles and rows (e & 3) + 8 (e >> 2) + 4 (lane >> 5), e = 0..15
    const int half = lane >> 5, lc = lane & 31;
    if ((EPI & PW_STORE) && col0 >= p.store_n0) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int col = col0 + (wn * WN + j) * 32 + lc;
                float bv = 0.f;                       // the bias is applied on the way out (PW_BIAS comes with PW_STORE only)
                if constexpr ((EPI & PW_BIAS) != 0) bv = col < p.N ? p.bias[col] : 0.f;
                if (row0 + BM <= p.M && col0 + BN <= p.N) {       // tile inside the matrix (workgroup-uniform): sixteen plain stores
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = row0 + (wm * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                        p.C[(long)row * p.ldc + (col - p.store_n0)] = acc[i][j][e] + bv;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = row0 + (wm * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                        if (row < p.M && col < p.N) p.C[(long)row * p.ldc + (col - p.store_n0)] = acc[i][j][e] + bv;
                    }
                }
            }
    }
    if (EPI & (PW_STATS | PW_SEL | PW_BWDSTATS)) {
        __syncthreads();                              // the loop's LDS is free: scratch [2 (wn)][WN * 32][up to 3]
        float *scr = reinterpret_cast<float *>(smem);
        const int rb_ = tm;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int ctile = (wn * WN + j) * 32 + lc, col = col0 + ctile;
            if (EPI & PW_STATS) {
                // (n, mean, M2) of this wave's 32 WM rows of the column: per lane shifted by its first value, Chan merges
                float mean = 0.f, M2 = 0.f, n = 0.f;
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    const float sh = acc[i][j][0];
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int e = 1; e < 16; ++e) {
                        const float d = acc[i][j][e] - sh;
                        s1 += d;
                        s2 = __builtin_fmaf(d, d, s2);
                    }
                    float mu = sh + s1 * (1.f / 16.f), m2 = fmaxf(s2 - s1 * s1 * (1.f / 16.f), 0.f);
                    const float omu = __shfl_xor(mu, 32), om2 = __shfl_xor(m2, 32);
                    const float d0 = omu - mu;
                    m2 = m2 + om2 + d0 * d0 * 8.f;       // 16 + 16 rows
                    mu = 0.5f * (mu + omu);
                    if (i == 0) { mean = mu; M2 = m2; n = 32.f; }
                    else {
                        const float tot = n + 32.f, dl = mu - mean;
                        mean += dl * (32.f / tot);
                        M2 += m2 + dl * dl * (n * 32.f / tot);
                        n = tot;
                    }
                }
                if (wm == 1 && half == 0) { scr[(wn * WN * 32 + j * 32 + lc) * 2] = mean; scr[(wn * WN * 32 + j * 32 + lc) * 2 + 1] = M2; }
                __syncthreads();
                if (wm == 0 && half == 0 && col < p.N) {
                    const float omu = scr[(wn * WN * 32 + j * 32 + lc) * 2], om2 = scr[(wn * WN * 32 + j * 32 + lc) * 2 + 1];
                    const float dl = omu - mean;
                    float *pr = p.rec + (long)rb_ * 3 * p.N;
                    pr[col] = 2.f * n;
                    pr[p.N + col] = 0.5f * (mean + omu);
                    pr[2 * p.N + col] = M2 + om2 + dl * dl * (0.5f * n);
                }
                __syncthreads();
            }
            if ((EPI & PW_SEL) && col0 < p.sel_n) {
                const float sg = (col < p.sel_n && p.sgn[col] < 0.f) ? -1.f : 1.f;      // sgn = the BatchNorm weight: only its sign is used
                float best = -INFINITY;
                int brow = 0;
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float v = sg * acc[i][j][e];
                        const int r = (wm * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                        if (v > best) { best = v; brow = r; }
                    }
                const float ob = __shfl_xor(best, 32);
                const int orow = __shfl_xor(brow, 32);
                if (ob > best || (ob == best && orow < brow)) { best = ob; brow = orow; }
                if (wm == 1 && half == 0) {
                    scr[(wn * WN * 32 + j * 32 + lc) * 2] = best;
                    scr[(wn * WN * 32 + j * 32 + lc) * 2 + 1] = __int_as_float(brow);
                }
                __syncthreads();
                if (wm == 0 && half == 0 && col < p.sel_n) {
                    const float ob2 = scr[(wn * WN * 32 + j * 32 + lc) * 2];
                    const int or2 = __float_as_int(scr[(wn * WN * 32 + j * 32 + lc) * 2 + 1]);
                    if (ob2 > best) { best = ob2; brow = or2; }     // wave-row 1 holds the higher rows: ties keep the lower
                    p.sel_val[(long)rb_ * p.sel_n + col] = best;
                    p.sel_arg[(long)rb_ * p.sel_n + col] = row0 - cloud * p.rows_per_cloud + brow;
                }
                __syncthreads();
            }
            if (EPI & PW_BWDSTATS) {
                float sb = 0.f, sg2 = 0.f;
                if (col < p.N) {
                    const float al = p.ealpha[col], de = p.edelta[(long)cloud * p.etstride + col];
                    const float mu = p.emu[(long)cloud * p.etstride + col], rr = p.er[col];   // emu: mean minus the cloud's shift
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int row = row0 + (wm * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                            const float yv = p.Yp[(long)row * p.ldyp + col];
                            const float u = __builtin_fmaf(yv, al, de);
                            const float h = acc[i][j][e] * (u > 0.f ? 1.f : p.slope);
                            sb += h;
                            sg2 = __builtin_fmaf(h, (yv - mu) * rr, sg2);
                        }
                }
                sb += __shfl_xor(sb, 32);
                sg2 += __shfl_xor(sg2, 32);
                if (wm == 1 && half == 0) { scr[(wn * WN * 32 + j * 32 + lc) * 2] = sb; scr[(wn * WN * 32 + j * 32 + lc) * 2 + 1] = sg2; }
                __syncthreads();
                if (wm == 0 && half == 0 && col < p.N) {
                    float *pr = p.rec2 + (long)rb_ * 2 * p.N;
                    pr[col] = sb + scr[(wn * WN * 32 + j * 32 + lc) * 2];
                    pr[p.N + col] = sg2 + scr[(wn * WN * 32 + j * 32 + lc) * 2 + 1];
                }
                __syncthreads();
            }
        }
    }
}

template <int WM, int WN, int PRO, int EPI, int PIPE = 1, int NP = 3>
int launch_rowgemm(const RowGemmArgs &a, hipStream_t st, const char *name) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    const int MT = (a.M + BM - 1) / BM, NT = (a.N + BN - 1) / BN;
    const int grid = ((MT + 7) / 8) * 8 * NT;
    size_t lds = (PIPE ? 2 : 1) * ((size_t)(BM / 32) * 2 * NP + (size_t)(BN / 32) * 2 * NP) * 1024 + sizeof(float) * 4 * (size_t)a.K1;
    const size_t scratch = sizeof(float) * 2 * 2 * WN * 32;
    if (lds < scratch) lds = scratch;
    static FsgLdsGrant grant;      // (per template instantiation; the grant itself is per device)
    if (!grant.raise((const void *)pw_rowgemm_kernel<WM, WN, PRO, EPI, PIPE, NP>, lds)) {
        fsg_set_error("%s: cannot raise dynamic LDS to %zu", name, lds);
        return FSG_ERR_HIP;
    }
    hipLaunchKernelGGL((pw_rowgemm_kernel<WM, WN, PRO, EPI, PIPE, NP>), dim3(grid), dim3(256), lds, st, a);
    FSG_CHECK_LAUNCH(name);
    return FSG_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// pw_tn_kernel:  part[s] (N1, N2) = sum over the rows m of slice s of  L'(m, n1) R'(m, n2)  -- the weight gradients
// dW = dy^T a (and the Gram matrix X^T X of the global-feature backward).  Both operands are (M, .) row-major activations,
// the contraction runs over ROWS: lane = column (coalesced 256-byte row segments), a thread collects eight consecutive rows
// of its column -- exactly one operand fragment (k = 8 h + j) -- applies the prologue with per-COLUMN constants it keeps in
// registers, splits, and writes 16 bytes per piece into the LDS fragment image.  Tile 64 T1 x 64 T2, four waves (2 x 2),
// 32 rows per iteration.  The S row slices are summed in slice order by pw_tn_reduce_kernel: no atomics, reproducible.
struct TnArgs {
    const float *L1, *LY1;   // left segment 1: (M, N1a), row stride ldl1; lpro == PRO_BNBWD: L1 = upstream gradient, LY1 = y
    const float *L2;         // left segment 2 (plain): (M, N1b), row stride ldl2
    long ldl1, ldl2;
    int N1a, N1b, lpro;
    const float *lalpha, *ldelta, *lP, *lQ;   // left tables (length N1a; ldelta / lP per cloud with stride lts)
    int lts;
    const float *R;          // right operand (M, N2), row stride ldr; rpro == PRO_BNACT: f(ralpha y + rdelta[cloud])
    long ldr;
    int N2, rpro;
    const float *ralpha, *rdelta;
    int rts;
    float slope;
    int M, rows_per_cloud, rows_per_slice;
    int ones;                // 1: one more left column behind segment 2 that is all ones -> result row N1a + N1b = column sums of R'
    float *part;             // (S, N1a + N1b + ones, N2)
};

template <int T1, int T2, int NP = 3>
__global__ __launch_bounds__(256, 2) void pw_tn_kernel(const TnArgs p) {   // (<= 256 VGPRs: two workgroups = two waves per SIMD, one transforming while the other multiplies)
    constexpr int BT1 = 64 * T1, BT2 = 64 * T2;
    __shared__ __attribute__((aligned(16))) u32x4 Limg[(BT1 / 32) * 2 * NP * 64];
    __shared__ __attribute__((aligned(16))) u32x4 Rimg[(BT2 / 32) * 2 * NP * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N1 = p.N1a + p.N1b + p.ones;
    const int nt2 = (p.N2 + BT2 - 1) / BT2;
    const int t1 = blockIdx.x / nt2, t2 = blockIdx.x - t1 * nt2, s = blockIdx.y;
    const int c1_0 = t1 * BT1, c2_0 = t2 * BT2;
    const int m0 = s * p.rows_per_slice, m1 = min(p.M, m0 + p.rows_per_slice);
    const int cloud = p.rows_per_cloud > 0 ? m0 / p.rows_per_cloud : 0;
    const int ks = wave >> 1, hh = wave & 1;            // this wave's fragment rows: 16 ks + 8 hh + 0..7 of every 32-row step
    // per-column constants of this thread's columns
    float la[T1], ld_[T1], lp[T1], lq[T1], lmask[T1], lone[T1];
    const float *lsrc[T1], *lysrc[T1];
    long lld[T1];
    bool lbwd[T1];
#pragma unroll
    for (int g = 0; g < T1; ++g) {
        const int col = c1_0 + 64 * g + lane;
        const bool seg2 = (c1_0 + 64 * g) >= p.N1a && p.N1b > 0;     // 64-column groups never straddle the segments (host check)
        const int cc = min(col, p.N1a + p.N1b - 1);
        lmask[g] = col < N1 ? 1.f : 0.f;
        lone[g] = (p.ones && col == p.N1a + p.N1b) ? 1.f : 0.f;
        lbwd[g] = !seg2 && p.lpro == PRO_BNBWD;
        lsrc[g] = seg2 ? p.L2 + (cc - p.N1a) : p.L1 + cc;
        lysrc[g] = lbwd[g] ? p.LY1 + cc : lsrc[g];
        lld[g] = seg2 ? p.ldl2 : p.ldl1;
        la[g] = ld_[g] = lp[g] = lq[g] = 0.f;
        if (lbwd[g]) {
            la[g] = p.lalpha[cc];
            ld_[g] = p.ldelta[(long)cloud * p.lts + cc];
            lp[g] = p.lP[(long)cloud * p.lts + cc];
            lq[g] = p.lQ[cc];
        }
    }
    float ralp[T2], rdel[T2], rmask[T2];
    const float *rsrc[T2];
#pragma unroll
    for (int g = 0; g < T2; ++g) {
        const int col = c2_0 + 64 * g + lane, cc = min(col, p.N2 - 1);
        rmask[g] = col < p.N2 ? 1.f : 0.f;
        rsrc[g] = p.R + cc;
        ralp[g] = p.rpro == PRO_BNACT ? p.ralpha[cc] : 1.f;
        rdel[g] = p.rpro == PRO_BNACT ? p.rdelta[(long)cloud * p.rts + cc] : 0.f;
    }
    float lv[T1][8], lyv[T1][8], rv[T2][8];
    auto fetch = [&](int mb) {                         // rows mb + 16 ks + 8 hh + j (clamped; rows >= m1 are zeroed at use)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long row = min(mb + 16 * ks + 8 * hh + j, p.M - 1);
#pragma unroll
            for (int g = 0; g < T1; ++g) {
                lv[g][j] = lsrc[g][row * lld[g]];
                lyv[g][j] = lysrc[g][row * lld[g]];
            }
#pragma unroll
            for (int g = 0; g < T2; ++g) rv[g][j] = rsrc[g][row * p.ldr];
        }
    };
    f32x16 acc[T1][T2];
#pragma unroll
    for (int i = 0; i < T1; ++i)
#pragma unroll
        for (int j = 0; j < T2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int w1 = wave >> 1, w2 = wave & 1;
    fetch(m0);
    for (int mb = m0; mb < m1; mb += 32) {
        __syncthreads();
#pragma unroll
        for (int g = 0; g < T1; ++g) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float live = (mb + 16 * ks + 8 * hh + j < m1) ? lmask[g] : 0.f;
                float v = lv[g][j];
                if (lbwd[g]) {
                    const float y = lyv[g][j], u = __builtin_fmaf(y, la[g], ld_[g]);
                    const float h = v * (u > 0.f ? 1.f : p.slope);
                    v = __builtin_fmaf(la[g], h, -__builtin_fmaf(lq[g], y, lp[g]));
                }
                v = lone[g] != 0.f ? 1.f : v;
                x[j] = v * live;
            }
            u32x4 h, m, l;
            split8(x, h, m, l);
            u32x4 *dst = Limg + (((2 * g + (lane >> 5)) * 2 + ks) * NP) * 64 + (lane & 31) + 32 * hh;
            dst[0] = h;
            if constexpr (NP == 3) {
                dst[64] = m;
                dst[128] = l;
            }
        }
#pragma unroll
        for (int g = 0; g < T2; ++g) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float live = (mb + 16 * ks + 8 * hh + j < m1) ? rmask[g] : 0.f;
                float v = rv[g][j];
                if (p.rpro == PRO_BNACT) {
                    const float u = __builtin_fmaf(v, ralp[g], rdel[g]);
                    v = u > 0.f ? u : u * p.slope;
                }
                x[j] = v * live;
            }
            u32x4 h, m, l;
            split8(x, h, m, l);
            u32x4 *dst = Rimg + (((2 * g + (lane >> 5)) * 2 + ks) * NP) * 64 + (lane & 31) + 32 * hh;
            dst[0] = h;
            if constexpr (NP == 3) {
                dst[64] = m;
                dst[128] = l;
            }
        }
        __syncthreads();
        if (mb + 32 < m1) fetch(mb + 32);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            u32x4 af[T1][3], bf[T2][3];
#pragma unroll
            for (int i = 0; i < T1; ++i)
#pragma unroll
                for (int q = 0; q < NP; ++q) af[i][q] = Limg[(((w1 * T1 + i) * 2 + k2) * NP + q) * 64 + lane];
#pragma unroll
            for (int j = 0; j < T2; ++j)
#pragma unroll
                for (int q = 0; q < NP; ++q) bf[j][q] = Rimg[(((w2 * T2 + j) * 2 + k2) * NP + q) * 64 + lane];
#pragma unroll
            for (int i = 0; i < T1; ++i)
#pragma unroll
                for (int j = 0; j < T2; ++j) {
                    f32x16 c = acc[i][j];
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, af[i][0]), bh = __builtin_bit_cast(bf16x8, bf[j][0]);
                    if constexpr (NP == 3) {
                        const bf16x8 am = __builtin_bit_cast(bf16x8, af[i][1]), al = __builtin_bit_cast(bf16x8, af[i][2]);
                        const bf16x8 bm = __builtin_bit_cast(bf16x8, bf[j][1]), bl = __builtin_bit_cast(bf16x8, bf[j][2]);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
    }
    float *out = p.part + (long)s * N1 * p.N2;
    const int half = lane >> 5, lc = lane & 31;
#pragma unroll
    for (int i = 0; i < T1; ++i)
#pragma unroll
        for (int j = 0; j < T2; ++j) {
            const int col = c2_0 + (w2 * T2 + j) * 32 + lc;
            if (c1_0 + BT1 <= N1 && c2_0 + BT2 <= p.N2) {        // block inside the output (workgroup-uniform): plain stores
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = c1_0 + (w1 * T1 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    out[(long)row * p.N2 + col] = acc[i][j][e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = c1_0 + (w1 * T1 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    if (row < N1 && col < p.N2) out[(long)row * p.N2 + col] = acc[i][j][e];
                }
            }
        }
}

// out rows [0, N1a) -> C1 (row stride ldc1), rows [N1a, N1) -> C2 (row stride ldc2): sum of the S slices in slice order
__global__ __launch_bounds__(256) void pw_tn_reduce_kernel(const float *__restrict__ part, int S, int N1, int N2, int N1a,
                                                           float *__restrict__ C1, long ldc1, float *__restrict__ C2, long ldc2) {
    const long total = (long)N1 * N2;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int q = 0;
        for (; q + 4 <= S; q += 4) {       // four loads in flight; fixed association ((s0+s4+..) + (s1+..)) + ...
            a0 += part[(long)q * total + t];
            a1 += part[(long)(q + 1) * total + t];
            a2 += part[(long)(q + 2) * total + t];
            a3 += part[(long)(q + 3) * total + t];
        }
        for (; q < S; ++q) a0 += part[(long)q * total + t];
        const float v = (a0 + a1) + (a2 + a3);
        const int r = (int)(t / N2), c = (int)(t - (long)r * N2);
        if (r < N1a) C1[(long)r * ldc1 + c] = v;
        else C2[(long)(r - N1a) * ldc2 + c] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Train-mode BatchNorm statistics from the (n, mean, M2) records of the row-GEMM epilogue (Chan's merge in fp64), with an
// optional per-cloud shift added to every record's mean (the first head layer: y = y0 + c[cloud], models/dgcnn.py:159-160),
// the running-statistics update of torch (unbiased variance) and the prologue tables of the consumer:
//   alpha[c] = gamma r,  delta[b][c] = alpha (shift[b][c] - mean) + beta,  emu[b][c] = mean - shift[b][c]
// 4 channels per workgroup, 64 contiguous record slices per channel (two to eight records each), LDS merge.
constexpr int FS = 64, FC = 4;   // record slices and channels per workgroup: few dependent round trips per thread
// optional tail (round 4): the global max-pool's finish (pw_max_finish_kernel) for the same channels, in the same launch -- the
// BatchNorm whose statistics this is feeds a max over the points through the monotone BatchNorm + LeakyReLU, and the (cloud,
// channel) maxima need exactly the alpha / delta this workgroup has just computed
struct MaxFinish {
    const float *sel_val;
    const int *sel_arg;
    const float *sgn;
    int tiles;
    float slope;
    float *out, *ysel;
    int *arg;
};
__global__ __launch_bounds__(256) void pw_bn_finalize_kernel(const MaxFinish mf, const float *__restrict__ rec, int R, int ldn, int c0, int C,
                                                             const float *__restrict__ shift, int B, int training,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             float eps, float momentum, float *__restrict__ running_mean,
                                                             float *__restrict__ running_var, float *__restrict__ mean_out,
                                                             float *__restrict__ invstd_out, float *__restrict__ alpha,
                                                             float *__restrict__ delta, float *__restrict__ emu,
                                                             float *__restrict__ cloud_mean, const float *__restrict__ gfeat,
                                                             const float *__restrict__ Wglob, long ldwg, int CGf,
                                                             float *__restrict__ shift_out) {
    __shared__ double red[3][FS][FC];
    __shared__ float stat[2][FC];
    __shared__ float part[8][FS][FC];
    __shared__ float shl[64][FC];                        // the in-kernel shift (B <= 64)
    const int ch = threadIdx.x & (FC - 1), sl = threadIdx.x / FC;
    const int c = blockIdx.x * FC + ch;
    const bool live = c < C;
    auto shift_of = [&](int b) -> float { return gfeat ? shl[b][ch] : shift[(long)b * C + c]; };
    if (gfeat) {
        // shift[b][c] = sum_j gfeat[b][j] Wglob[c][j]: the per-cloud constant of the first head layer (models/dgcnn.py:159-160:
        // the repeated global feature times its block of the weight), computed here instead of by a vendor GEMM launch.
        // Slice sl sums j = sl, sl + FS, ...; fixed-order LDS fold.  Written to shift_out (B, C), which the caller passes as
        // `shift` too.
        for (int b0 = 0; b0 < B; b0 += 8) {
            float a[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = 0.f;
            if (live)
                for (int j = sl; j < CGf; j += FS) {
                    const float w = Wglob[(long)c * ldwg + j];
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (b0 + q < B) a[q] = __builtin_fmaf(gfeat[(long)(b0 + q) * CGf + j], w, a[q]);
                }
#pragma unroll
            for (int q = 0; q < 8; ++q) part[q][sl][ch] = a[q];
            __syncthreads();
            if (sl < 8 && b0 + sl < B && live) {          // slice index doubles as the cloud inside the chunk
                float t = 0.f;
#pragma unroll 8
                for (int q = 0; q < FS; ++q) t += part[sl][q][ch];
                shift_out[(long)(b0 + sl) * C + c] = t;
                shl[b0 + sl][ch] = t;
            }
            __syncthreads();
        }
    }
    // max-pool finish, first half (independent of the statistics, so its loads fly with the records'): slice sl takes (cloud,
    // tile group) pairs -- up to 8 groups of row blocks per cloud -- and leaves (best value, lowest row) per pair in LDS
    __shared__ float pbv[FS * 8][FC];
    __shared__ int pba[FS * 8][FC];
    if (mf.sel_val && live) {
        const int TG = mf.tiles < 8 ? mf.tiles : 8;
        const int tper = (mf.tiles + TG - 1) / TG;
        for (int pr0 = sl; pr0 < B * TG; pr0 += FS) {
            const int b = pr0 / TG, tg = pr0 - b * TG;
            float bv = -INFINITY;
            int ba = 0x7fffffff;
            for (int t = tg * tper; t < min(mf.tiles, (tg + 1) * tper); ++t) {
                const float v = mf.sel_val[((long)b * mf.tiles + t) * C + c];
                const int a = mf.sel_arg[((long)b * mf.tiles + t) * C + c];
                if (v > bv || (v == bv && a < ba)) { bv = v; ba = a; }
            }
            pbv[pr0][ch] = bv;
            pba[pr0][ch] = ba;
        }
    }
    const int rpc = B > 0 ? R / B : R;                  // records per cloud
    // slice sl owns the records [sl * per, (sl + 1) * per): contiguous, so a slice stays inside one cloud when per | rpc
    const int per = (R + FS - 1) / FS;
    if (training) {
        double a = 0.0, bm = 0.0, cm = 0.0;
        if (live) {
            const int r0 = sl * per, r1 = min(R, r0 + per);
            for (int rb = r0; rb < r1; rb += 8) {           // eight records' loads in flight together (one round trip, not eight)
                float nv[8], mv[8], qv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float *pr = rec + (long)min(rb + u, r1 - 1) * 3 * ldn + c0 + c;
                    nv[u] = pr[0];
                    mv[u] = pr[ldn];
                    qv[u] = pr[2 * ldn];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (rb + u < r1) {
                        const double n = nv[u];
                        double mu = mv[u];
                        if (shift) mu += (double)shift_of((rb + u) / rpc);
                        a += n;
                        bm += n * mu;
                        cm += (double)qv[u] + n * mu * mu;
                    }
            }
        }
        red[0][sl][ch] = a;
        red[1][sl][ch] = bm;
        red[2][sl][ch] = cm;
        __syncthreads();
        if (sl == 0 && live) {
            double n = 0.0;
            bm = 0.0;
            cm = 0.0;
#pragma unroll 8
            for (int q = 0; q < FS; ++q) { n += red[0][q][ch]; bm += red[1][q][ch]; cm += red[2][q][ch]; }
            const double mu = n > 0.0 ? bm / n : 0.0;
            double M2 = n > 0.0 ? cm - bm * mu : 0.0;
            M2 = M2 > 0.0 ? M2 : 0.0;
            const double var = n > 0.0 ? M2 / n : 0.0;
            const float muf = (float)mu, rf = (float)(1.0 / sqrt(var + (double)eps));
            mean_out[c] = muf;
            invstd_out[c] = rf;
            stat[0][ch] = muf;
            stat[1][ch] = rf;
            if (running_mean) {
                const double unbiased = n > 1.0 ? M2 / (n - 1.0) : var;
                running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
                running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
            }
        }
    } else if (sl == 0 && live) {                        // eval: running statistics (the caller passes them as mean / invstd)
        stat[0][ch] = mean_out[c];
        stat[1][ch] = invstd_out[c];
    }
    __syncthreads();
    if (!live) return;
    const float mu = stat[0][ch], r = stat[1][ch];
    const float al = gamma[c] * r;
    if (sl == 0) alpha[c] = al;
    const int nb = shift ? B : 1;
    for (int b = sl; b < nb; b += FS) {
        const float sh = shift ? shift_of(b) : 0.f;
        delta[(long)b * C + c] = __builtin_fmaf(al, sh - mu, beta[c]);
        if (emu) emu[(long)b * C + c] = mu - sh;
    }
    if (mf.sel_val) {       // (never together with a shift: delta has one row) -- one slice per cloud folds its tile groups in order
        const int TG = mf.tiles < 8 ? mf.tiles : 8;
        const float de = __builtin_fmaf(al, -mu, beta[c]);
        for (int b = sl; b < B; b += FS) {
            float bv = -INFINITY;
            int ba = 0x7fffffff;
            for (int tg = 0; tg < TG; ++tg) {
                const float v = pbv[b * TG + tg][ch];
                const int a = pba[b * TG + tg][ch];
                if (v > bv || (v == bv && a < ba)) { bv = v; ba = a; }
            }
            const float yv = (mf.sgn[c] < 0.f ? -1.f : 1.f) * bv;
            mf.ysel[(long)b * C + c] = yv;
            mf.arg[(long)b * C + c] = ba == 0x7fffffff ? 0 : ba;
            const float u = __builtin_fmaf(yv, al, de);
            mf.out[(long)b * C + c] = u > 0.f ? u : u * mf.slope;
        }
    }
    if (cloud_mean && training) {                        // unshifted per-cloud mean of the records (first head layer's backward)
        const bool aligned = per > 0 && rpc % per == 0 && R % per == 0;
        for (int b = sl; b < B; b += FS) {
            double n = 0.0, sm = 0.0;
            if (aligned) {                               // whole slices per cloud: their (n, n mu) sums are in LDS already; the
                for (int q = b * (rpc / per); q < (b + 1) * (rpc / per); ++q) { n += red[0][q][ch]; sm += red[1][q][ch]; }
                if (shift && n > 0.0) sm -= n * (double)shift_of(b);      // slices carry the cloud's shift: take it out
            } else {
                for (int q = 0; q < rpc; ++q) {
                    const float *pr = rec + (long)(b * rpc + q) * 3 * ldn + c0 + c;
                    n += pr[0];
                    sm += (double)pr[0] * pr[ldn];
                }
            }
            cloud_mean[(long)b * C + c] = (float)(n > 0.0 ? sm / n : 0.0);
        }
    }
}

// out[b, k] = sum_j x[b, j] W[k * ldw + j]  (B x C0 outputs behind a CG-long reduction: the per-cloud constant of the first head
// layer, models/dgcnn.py:159-160).  One wave per output, lanes stride over j, DPP-free shuffle fold in a fixed order.
__global__ __launch_bounds__(256) void pw_cloud_linear_kernel(const float *__restrict__ x, const float *__restrict__ W, long ldw,
                                                              int B, int C0, int CG, float *__restrict__ out) {
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (task >= B * C0) return;
    const int b = task / C0, k = task - b * C0;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int j = lane;
    for (; j + 192 < CG; j += 256) {
        a0 = __builtin_fmaf(x[(long)b * CG + j], W[(long)k * ldw + j], a0);
        a1 = __builtin_fmaf(x[(long)b * CG + j + 64], W[(long)k * ldw + j + 64], a1);
        a2 = __builtin_fmaf(x[(long)b * CG + j + 128], W[(long)k * ldw + j + 128], a2);
        a3 = __builtin_fmaf(x[(long)b * CG + j + 192], W[(long)k * ldw + j + 192], a3);
    }
    for (; j < CG; j += 64) a0 = __builtin_fmaf(x[(long)b * CG + j], W[(long)k * ldw + j], a0);
    float a = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if (lane == 0) out[(long)b * C0 + k] = a;
}

// sel records of the row-GEMM epilogue -> global max-pool through the monotone BatchNorm + LeakyReLU (models/dgcnn.py:
// 134-137,156): per (cloud, channel) the best of the cloud's row blocks (lowest row on ties); ysel = sgn * best,
// out = lrelu(alpha ysel + delta)
__global__ __launch_bounds__(256) void pw_max_finish_kernel(const float *__restrict__ sel_val, const int *__restrict__ sel_arg,
                                                            const float *__restrict__ sgn, const float *__restrict__ alpha,
                                                            const float *__restrict__ delta, int tiles, int C, float slope,
                                                            float *__restrict__ out, float *__restrict__ ysel,
                                                            int *__restrict__ arg) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float bv = -INFINITY;
    int ba = 0;
    for (int t = 0; t < tiles; ++t) {
        const float v = sel_val[((long)b * tiles + t) * C + c];
        const int a = sel_arg[((long)b * tiles + t) * C + c];
        if (v > bv || (v == bv && a < ba)) { bv = v; ba = a; }
    }
    const float yv = (sgn[c] < 0.f ? -1.f : 1.f) * bv;
    ysel[(long)b * C + c] = yv;
    arg[(long)b * C + c] = ba;
    const float u = __builtin_fmaf(yv, alpha[c], delta[c]);
    out[(long)b * C + c] = u > 0.f ? u : u * slope;
}

// BatchNorm backward sums from the (sum h, sum h yhat) records -> dbeta, dgamma and the prologue tables of the products
// that consume dy = alpha h - P - Q y:   Q = alpha r dgamma / M,   P[b] = alpha (dbeta / M - emu[b] r dgamma / M)
// and, for the first head layer, the per-cloud column sums of dy (the gradient of the per-cloud constant c):
//   dc[b] = alpha sum_{m in b} h - n_b P[b] - Q n_b cloud_mean[b]
__global__ __launch_bounds__(256) void pw_bnbwd_finalize_kernel(const float *__restrict__ rec2, int R, int C, int B, long M,
                                                                int training, const float *__restrict__ alpha,
                                                                const float *__restrict__ invstd, const float *__restrict__ emu,
                                                                int emu_per_cloud, const float *__restrict__ cloud_mean,
                                                                float *__restrict__ dbeta, float *__restrict__ dgamma,
                                                                float *__restrict__ P, float *__restrict__ Q,
                                                                float *__restrict__ dc) {
    __shared__ double red[2][FS][FC];
    __shared__ float tot[2][FC];
    const int ch = threadIdx.x & (FC - 1), sl = threadIdx.x / FC;
    const int c = blockIdx.x * FC + ch;
    const bool live = c < C;
    const int per = (R + FS - 1) / FS;
    double sb = 0.0, sg = 0.0;
    if (live) {
        const int r0 = sl * per, r1 = min(R, r0 + per);
        for (int rb = r0; rb < r1; rb += 8) {               // eight records' loads in flight together
            float bv[8], gv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long r = min(rb + u, r1 - 1);
                bv[u] = rec2[r * 2 * C + c];
                gv[u] = rec2[r * 2 * C + C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (rb + u < r1) {
                    sb += (double)bv[u];
                    sg += (double)gv[u];
                }
        }
    }
    red[0][sl][ch] = sb;
    red[1][sl][ch] = sg;
    __syncthreads();
    if (sl == 0 && live) {
        sb = sg = 0.0;
#pragma unroll 8
        for (int q = 0; q < FS; ++q) { sb += red[0][q][ch]; sg += red[1][q][ch]; }
        dbeta[c] = (float)sb;
        dgamma[c] = (float)sg;
        tot[0][ch] = (float)sb;
        tot[1][ch] = (float)sg;
    }
    __syncthreads();
    if (!live) return;
    const float invM = 1.0f / (float)M;
    const float al = alpha[c], r = invstd[c];
    const float db = training ? tot[0][ch] * invM : 0.f, dg = training ? tot[1][ch] * invM * r : 0.f;
    const float q = al * dg;
    if (sl == 0) Q[c] = q;
    const int nb = emu_per_cloud ? B : 1;
    const int rpc = B > 0 ? R / B : R;
    const bool aligned = per > 0 && rpc % per == 0 && R % per == 0;
    for (int b = sl; b < nb; b += FS) {
        const float pb = al * (db - emu[(long)b * C + c] * dg);
        P[(long)b * C + c] = pb;
        if (dc) {
            double sh = 0.0;
            if (aligned) {
                for (int t = b * (rpc / per); t < (b + 1) * (rpc / per); ++t) sh += red[0][t][ch];
            } else {
                for (int t = 0; t < rpc; ++t) sh += (double)rec2[(long)(b * rpc + t) * 2 * C + c];
            }
            const float nbf = (float)(M / B);
            dc[(long)b * C + c] = al * (float)sh - nbf * pb - q * nbf * cloud_mean[(long)b * C + c];
        }
    }
}

// last layer, backward (models/dgcnn.py:146: Conv1d(128, classes) with bias, no BatchNorm behind it):
//   da[m, k] = sum_j g[m, j] W3[j, k]  (classes <= 8), stored, and the BatchNorm backward sums of the layer in front:
//   h = da f'(alpha y + delta), records (sum h, sum h yhat) per 32-row block.  lane = channel, 4 waves x 8 rows, all loads of a
//   wave's rows in flight together.
constexpr int LB_ROWS = 32;
__global__ __launch_bounds__(256) void pw_logits_bwd_kernel(const float *__restrict__ g, int cls, const float *__restrict__ W3,
                                                            const float *__restrict__ y, const float *__restrict__ alpha,
                                                            const float *__restrict__ delta, const float *__restrict__ mean,
                                                            const float *__restrict__ invstd, long M, int C, float slope,
                                                            float *__restrict__ da, float *__restrict__ rec2) {
    __shared__ float red[2][4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long r0 = (long)blockIdx.x * LB_ROWS + wave * 8;
    float gv[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[i][j] = (j < cls && r0 + i < M) ? g[(r0 + i) * cls + j] : 0.f;
    for (int cb = 0; cb < C; cb += 64) {
        const int c = cb + lane;
        float w[8], yv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = j < cls ? W3[(long)j * C + c] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) yv[i] = y[min(r0 + i, M - 1) * C + c];
        const float al = alpha[c], de = delta[c], mu = mean[c], rr = invstd[c];
        float sb = 0.f, sg = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) a = __builtin_fmaf(gv[i][j], w[j], a);
            if (r0 + i < M) da[(r0 + i) * C + c] = a;
            const float u = __builtin_fmaf(yv[i], al, de);
            const float h = a * (u > 0.f ? 1.f : slope);      // rows behind M carry g = 0, so h = 0
            sb += h;
            sg = __builtin_fmaf(h, (yv[i] - mu) * rr, sg);
        }
        red[0][wave][lane] = sb;
        red[1][wave][lane] = sg;
        __syncthreads();
        if (wave == 0) {
            float *pr = rec2 + (long)blockIdx.x * 2 * C;
            pr[c] = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
            pr[C + c] = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
        }
        __syncthreads();
    }
}

// global-feature backward, per channel j (the Gram form of DESIGN.md).  First the two tiny products around the per-cloud
// constant c = g W0_global^T of the first head layer (models/dgcnn.py:159-160), which a vendor GEMM would need a launch each for:
//   dg[b, j] = sum_k dc[b, k] W0g[k, j]   (gradient reaching the global feature),   dW0g[k, j] = sum_b dc[b, k] g[b, j]
// then, with h[b] = dg[b] f'(alpha ysel[b] + delta):
//   dbeta = sum_b h,  dgamma = sum_b h yhat_sel,  Q = alpha r dgamma / M,  P = alpha (dbeta / M - mean r dgamma / M),
//   coef[b] = alpha h[b]   (weight of the selected row arg[b] in dy)
// <clouds a thread carries, waves per workgroup>: <8, 16> up to 8 clouds per rank (each wave walks every 16th row of W0g, all
// its loads in flight at once), <32, 4> up to 32 (the reference's experiment scripts train with 32 clouds per GPU): the
// per-wave partial table [waves][clouds][64] must fit LDS next to the (B, C0) block of dc
template <int GP_MAXB, int GP_WAVES>
__global__ __launch_bounds__(64 * GP_WAVES) void pw_gf_prep_kernel(const float *__restrict__ dc, const float *__restrict__ W0g, long ldw0,
                                                         int C0, const float *__restrict__ gfeat, float *__restrict__ dW0g,
                                                         long lddw0, const float *__restrict__ dg_in,
                                                         const float *__restrict__ ysel,
                                                         const float *__restrict__ alpha, const float *__restrict__ delta,
                                                         const float *__restrict__ mean, const float *__restrict__ invstd,
                                                         int B, int C, long M, int training, float slope,
                                                         float *__restrict__ dbeta, float *__restrict__ dgamma,
                                                         float *__restrict__ P, float *__restrict__ Q, float *__restrict__ coef,
                                                         const float *__restrict__ Wgl, long ldwgl, int KL,
                                                         float *__restrict__ Wq, long ldwq) {
    extern __shared__ float dcs[];     // (B, C0) when dc is given, then [GP_WAVES][GP_MAXB][64] partial dg
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = min(blockIdx.x * 64 + lane, C - 1);       // 64 channels per workgroup, the four waves split k
    const bool livec = blockIdx.x * 64 + lane < C;
    float dgv[GP_MAXB];
#pragma unroll
    for (int b = 0; b < GP_MAXB; ++b) dgv[b] = 0.f;
    // what the first wave needs after the contraction is requested now (one memory round trip less at the end)
    float al = 0.f, de = 0.f, mu = 0.f, r = 0.f, ysv[GP_MAXB];
#pragma unroll
    for (int b = 0; b < GP_MAXB; ++b) ysv[b] = 0.f;
    if (wave == 0) {
        al = alpha[c]; de = delta[c]; mu = mean[c]; r = invstd[c];
#pragma unroll
        for (int b = 0; b < GP_MAXB; ++b)
            if (b < B) ysv[b] = ysel[(long)b * C + c];
    }
    if (dc) {
        float *pd = dcs + GP_MAXB * C0;
        for (int e = threadIdx.x; e < GP_MAXB * C0; e += 64 * GP_WAVES) dcs[e] = e < B * C0 ? dc[e] : 0.f;   // clouds behind B: zeros, no tests below
        __syncthreads();
        float gv[GP_MAXB];
#pragma unroll
        for (int b = 0; b < GP_MAXB; ++b) gv[b] = b < B ? gfeat[(long)b * C + c] : 0.f;
        (void)gv;
        (void)dW0g;
        (void)lddw0;
        // sixteen rows of W0g per round, every load issued before the first fma (with four waves and eight rows in flight the
        // kernel was eight dependent round trips long: 16.8 us)
        for (int k0 = wave; k0 < C0; k0 += 16 * GP_WAVES) {
            float w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = W0g[(long)min(k0 + u * GP_WAVES, C0 - 1) * ldw0 + c];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = k0 + u * GP_WAVES, kc = min(k, C0 - 1);
                const float wz = k < C0 ? w[u] : 0.f;              // rows behind C0 contribute zero: a select, not a branch
#pragma unroll
                for (int b = 0; b < GP_MAXB; ++b) dgv[b] = __builtin_fmaf(dcs[b * C0 + kc], wz, dgv[b]);
            }
        }
#pragma unroll
        for (int b = 0; b < GP_MAXB; ++b) pd[(wave * GP_MAXB + b) * 64 + lane] = dgv[b];
        __syncthreads();
#pragma unroll
        for (int b = 0; b < GP_MAXB; ++b) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < GP_WAVES; ++q) t += pd[(q * GP_MAXB + b) * 64 + lane];     // wave order: fixed
            dgv[b] = t;
        }
    } else {
#pragma unroll
        for (int b = 0; b < GP_MAXB; ++b) dgv[b] = b < B ? dg_in[(long)b * C + c] : 0.f;
    }
    if (wave == 0 && livec) {
        float sb = 0.f, sg = 0.f;
#pragma unroll
        for (int b = 0; b < GP_MAXB; ++b)
            if (b < B) {
                const float ys = ysv[b], u = __builtin_fmaf(ys, al, de);
                const float h = dgv[b] * (u > 0.f ? 1.f : slope);
                coef[(long)b * C + c] = al * h;
                sb += h;
                sg = __builtin_fmaf(h, (ys - mu) * r, sg);
            }
        dbeta[c] = sb;
        dgamma[c] = sg;
        const float invM = 1.0f / (float)M;
        const float db = training ? sb * invM : 0.f, dgm = training ? sg * invM * r : 0.f;
        const float q = al * dgm, pp = al * (db - mu * dgm);
        Q[c] = q;
        P[c] = pp;
    }
}

// dW0g[k, j] = sum_b dc[b, k] g[b, j]  (C0 x CG outputs, B terms each): thread = column j, eight rows k per workgroup row
template <int GP_MAXB>
__global__ __launch_bounds__(256) void pw_outer_kernel(const float *__restrict__ dc, const float *__restrict__ g, int B, int C0,
                                                       int CG, float *__restrict__ out, long ldo, const float *__restrict__ Q,
                                                       const float *__restrict__ P, const float *__restrict__ Wgl, long ldwgl,
                                                       int KL, float *__restrict__ Wq, long ldwq) {
    if (Wq) {
        // rows [Q[c] W[c, :] | -P[c]] of the CG channels: the left operand of [M1 ; npvec] = [Q o W | -P]^T W, spread over the
        // whole grid (thread = column, workgroup = a run of channel rows)
        const int nwg = gridDim.x * gridDim.y, wg = blockIdx.y * gridDim.x + blockIdx.x;
        const int per = (CG + nwg - 1) / nwg;
        for (int c = wg * per; c < min(CG, (wg + 1) * per); ++c) {
            const float q = Q[c], pp = P[c];
            for (int jj = threadIdx.x; jj <= KL; jj += 256)
                Wq[(long)c * ldwq + jj] = jj < KL ? q * Wgl[(long)c * ldwgl + jj] : -pp;
        }
    }
    const int j = blockIdx.x * 256 + threadIdx.x, k0 = blockIdx.y * 8;
    if (j >= CG) return;
    float gv[GP_MAXB];
#pragma unroll
    for (int b = 0; b < GP_MAXB; ++b) gv[b] = b < B ? g[(long)b * CG + j] : 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = k0 + q;
        if (k >= C0) break;
        float a = 0.f;
#pragma unroll
        for (int b = 0; b < GP_MAXB; ++b) a = __builtin_fmaf(b < B ? dc[(long)b * C0 + k] : 0.f, gv[b], a);
        out[(long)k * ldo + j] = a;
    }
}

// dX rows of the selected points: for every cloud, dX[b N + arg[b,c], :] += coef[b,c] W[c, :], summed per destination row in
// channel order (no atomics, reproducible).  pw_sort_sel_kernel sorts the cloud's keys (arg << 12 | c) once (bitonic in LDS,
// one key per thread); pw_scatter_rows_kernel gives every workgroup a chunk of the sorted list: it owns the row segments whose
// first entry lies in its chunk, issues ALL of the chunk's loads (weights rows, coefficients, old dX values) before the first
// use, accumulates a segment in registers (thread = column) and writes its dX row once.
constexpr int SC_CHUNK = 16;
// 1024 keys, one workgroup of 256 threads per cloud, four keys per thread in registers: compare-exchanges with a partner inside
// the thread are register swaps (19 of the 55 stages), partners inside the wave come through __shfl_xor (33 stages), only the
// three stages whose partner sits in another wave go through LDS and a barrier
__global__ __launch_bounds__(256) void pw_sort_sel_1024_kernel(const int *__restrict__ arg, int C, unsigned *__restrict__ sorted) {
    constexpr int KPL = 4, NK = 1024;
    __shared__ unsigned xch[NK];
    const int b = blockIdx.x, tid = threadIdx.x;
    unsigned v[KPL];
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
        const int i = tid * KPL + r;
        v[r] = i < C ? ((unsigned)arg[(long)b * C + i] << 12) | (unsigned)i : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int k = 2; k <= NK; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < KPL) {
#pragma unroll
                for (int r = 0; r < KPL; ++r) {
                    if ((r & j) == 0) {
                        const bool up = ((tid * KPL + r) & k) == 0;
                        const unsigned a = v[r], c = v[r | j];
                        const unsigned lo = a < c ? a : c, hi = a < c ? c : a;
                        v[r] = up ? lo : hi;
                        v[r | j] = up ? hi : lo;
                    }
                }
            } else {
                const int J = j / KPL;                  // partner thread = tid ^ J
                const bool lower = (tid & J) == 0;
                unsigned other[KPL];
                if (J < 64) {
#pragma unroll
                    for (int r = 0; r < KPL; ++r) other[r] = __shfl_xor(v[r], J);
                } else {
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < KPL; ++r) xch[tid * KPL + r] = v[r];
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < KPL; ++r) other[r] = xch[(tid ^ J) * KPL + r];
                }
#pragma unroll
                for (int r = 0; r < KPL; ++r) {
                    const bool up = ((tid * KPL + r) & k) == 0;
                    const unsigned lo = v[r] < other[r] ? v[r] : other[r], hi = v[r] < other[r] ? other[r] : v[r];
                    v[r] = (lower == up) ? lo : hi;
                }
            }
        }
    }
    uint4 o;
    o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
    reinterpret_cast<uint4 *>(sorted + (long)b * NK)[tid] = o;
}

// larger channel counts: bitonic sort in LDS
__global__ __launch_bounds__(1024) void pw_sort_sel_kernel(const int *__restrict__ arg, int C, int Cp2, unsigned *__restrict__ sorted) {
    extern __shared__ unsigned keys[];   // [Cp2]
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < Cp2; i += 1024) keys[i] = i < C ? ((unsigned)arg[(long)b * C + i] << 12) | (unsigned)i : 0xFFFFFFFFu;
    __syncthreads();
    for (int k = 2; k <= Cp2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < Cp2; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned a = keys[i], c = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { keys[i] = c; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < Cp2; i += 1024) sorted[(long)b * Cp2 + i] = keys[i];
}

__global__ __launch_bounds__(256) void pw_scatter_rows_kernel(const float *__restrict__ coef, const unsigned *__restrict__ sorted,
                                                              const float *__restrict__ W, long ldw, int C, int Cp2, int K,
                                                              int Npts, float *__restrict__ dX, long ldx) {
    const int b = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
    const unsigned *keys = sorted + (long)b * Cp2;
    const int c0 = chunk * SC_CHUNK;
    if (c0 >= C) return;
    // the keys are the same for every thread of the workgroup: held as SCALARS (readfirstlane), so that the segment logic below
    // is scalar control flow and the row / channel offsets of the loads are scalar too
    unsigned key[SC_CHUNK];
    const unsigned prev = __builtin_amdgcn_readfirstlane(c0 > 0 ? keys[c0 - 1] : 0xFFFFFFFFu);
    const unsigned next = __builtin_amdgcn_readfirstlane(c0 + SC_CHUNK < C ? keys[c0 + SC_CHUNK] : 0xFFFFFFFFu);
#pragma unroll
    for (int i = 0; i < SC_CHUNK; ++i) key[i] = __builtin_amdgcn_readfirstlane(c0 + i < C ? keys[c0 + i] : 0xFFFFFFFFu);
    for (int k0 = 0; k0 < K; k0 += 256) {
        const int k = min(k0 + tid, K - 1);
        float wv[SC_CHUNK], cf[SC_CHUNK], old[SC_CHUNK];
#pragma unroll
        for (int i = 0; i < SC_CHUNK; ++i) {      // every load unconditional (padded entries read row 0 / channel 0 and are never used)
            const bool live = key[i] != 0xFFFFFFFFu;
            const unsigned r = live ? key[i] >> 12 : 0u, c = live ? key[i] & 4095u : 0u;
            wv[i] = W[(long)c * ldw + k];
            cf[i] = coef[(long)b * C + c];
            old[i] = dX[((long)b * Npts + r) * ldx + k];
        }
        const bool act = k0 + tid < K;      // threads behind K keep running (clamped column): only their stores are masked, the segment logic stays scalar
        float acc = 0.f, base = 0.f;
        bool open = false;                                  // a segment whose head lies in this chunk is being summed
        unsigned row = 0;
#pragma unroll
        for (int i = 0; i < SC_CHUNK; ++i) {
            if (key[i] == 0xFFFFFFFFu) break;
            const unsigned r = key[i] >> 12;
            const bool head = (i == 0 ? (prev >> 12) : (key[i - 1] >> 12)) != r || (i == 0 && prev == 0xFFFFFFFFu);
            if (head) {
                if (open && act) dX[((long)b * Npts + row) * ldx + k] = base + acc;
                open = true;
                row = r;
                base = old[i];
                acc = 0.f;
            }
            if (open) acc = __builtin_fmaf(cf[i], wv[i], acc);
        }
        if (open) {                                          // the last segment may run on into the next chunks (rare)
            if ((next >> 12) == row)
            for (int e = c0 + SC_CHUNK; e < C && (keys[e] >> 12) == row; ++e) {
                const unsigned c = keys[e] & 4095u;
                acc = __builtin_fmaf(coef[(long)b * C + c], W[(long)c * ldw + k], acc);
            }
            if (act) dX[((long)b * Npts + row) * ldx + k] = base + acc;
        }
    }
}

// dW of the global-feature layer, Gram form:  dW[c, :] = sum_b coef[b,c] X[b N + arg[b,c], :] - P[c] s - Q[c] (W G)[c, :]
// (the small product W G is formed here: four channels per workgroup, their W rows in LDS; the four waves split the rows i of G,
// lane = column: one G load per (i, k) serves the four channels; fixed-order LDS fold over the waves)
__global__ __launch_bounds__(256) void pw_gf_dw_kernel(const float *__restrict__ coef, const int *__restrict__ arg,
                                                       const float *__restrict__ X, long ldx, const float *__restrict__ s,
                                                       const float *__restrict__ W, long ldw, const float *__restrict__ G,
                                                       const float *__restrict__ P, const float *__restrict__ Q, int B, int C,
                                                       int K, int Npts, float *__restrict__ dW, long lddw) {
    extern __shared__ float wr[];      // [4][K] rows of W, then [4 waves][4 channels][64] partial products
    float *red = wr + 4 * K;
    const int c0 = blockIdx.x * 4, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int e = tid; e < 4 * K; e += 256) {
        const int q = e / K, i = e - q * K;
        wr[e] = c0 + q < C ? W[(long)(c0 + q) * ldw + i] : 0.f;
    }
    __syncthreads();
    const int per = (K + 3) / 4, i0 = wave * per, i1 = min(K, i0 + per);
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = min(k0 + lane, K - 1);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
        for (int i = i0; i < i1; ++i) {
            const float g = G[(long)i * K + k];
            a0 = __builtin_fmaf(wr[i], g, a0);
            a1 = __builtin_fmaf(wr[K + i], g, a1);
            a2 = __builtin_fmaf(wr[2 * K + i], g, a2);
            a3 = __builtin_fmaf(wr[3 * K + i], g, a3);
        }
        red[(wave * 4 + 0) * 64 + lane] = a0;
        red[(wave * 4 + 1) * 64 + lane] = a1;
        red[(wave * 4 + 2) * 64 + lane] = a2;
        red[(wave * 4 + 3) * 64 + lane] = a3;
        __syncthreads();
        // wave q finishes channel c0 + q
        const int c = c0 + wave;
        if (c < C && k0 + lane < K) {
            const float wg = (red[(0 * 4 + wave) * 64 + lane] + red[(1 * 4 + wave) * 64 + lane]) +
                             (red[(2 * 4 + wave) * 64 + lane] + red[(3 * 4 + wave) * 64 + lane]);
            float a = -__builtin_fmaf(Q[c], wg, P[c] * s[k]);
            float xv[8];
            int bb = 0;
            for (; bb + 8 <= B; bb += 8) {
#pragma unroll
                for (int q = 0; q < 8; ++q) xv[q] = X[((long)(bb + q) * Npts + arg[(long)(bb + q) * C + c]) * ldx + k];
#pragma unroll
                for (int q = 0; q < 8; ++q) a = __builtin_fmaf(coef[(long)(bb + q) * C + c], xv[q], a);
            }
            for (; bb < B; ++bb) a = __builtin_fmaf(coef[(long)bb * C + c], X[((long)bb * Npts + arg[(long)bb * C + c]) * ldx + k], a);
            dW[(long)c * lddw + k] = a;
        }
        __syncthreads();
    }
}

// several deferred reductions in one launch: job j owns the elements [first[j], first[j + 1])
struct TnReduceJobs {
    const float *part[FSG_PW_MAX_REDUCE_JOBS];
    float *C1[FSG_PW_MAX_REDUCE_JOBS], *C2[FSG_PW_MAX_REDUCE_JOBS];
    long ldc1[FSG_PW_MAX_REDUCE_JOBS], ldc2[FSG_PW_MAX_REDUCE_JOBS], first[FSG_PW_MAX_REDUCE_JOBS + 1];
    int S[FSG_PW_MAX_REDUCE_JOBS], N1[FSG_PW_MAX_REDUCE_JOBS], N2[FSG_PW_MAX_REDUCE_JOBS], N1a[FSG_PW_MAX_REDUCE_JOBS];
    int n;
};
__global__ __launch_bounds__(256) void pw_tn_reduce_many_kernel(const TnReduceJobs jobs) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= jobs.first[jobs.n]) return;
    int j = 0;
#pragma unroll
    for (int q = 1; q < FSG_PW_MAX_REDUCE_JOBS; ++q)
        if (q < jobs.n && g >= jobs.first[q]) j = q;
    const long t = g - jobs.first[j];
    const int S = jobs.S[j], N2 = jobs.N2[j];
    const long total = (long)jobs.N1[j] * N2;
    const float *part = jobs.part[j];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int q = 0;
    for (; q + 4 <= S; q += 4) {
        a0 += part[(long)q * total + t];
        a1 += part[(long)(q + 1) * total + t];
        a2 += part[(long)(q + 2) * total + t];
        a3 += part[(long)(q + 3) * total + t];
    }
    for (; q < S; ++q) a0 += part[(long)q * total + t];
    const float v = (a0 + a1) + (a2 + a3);
    const int r = (int)(t / N2), c = (int)(t - (long)r * N2);
    if (r < jobs.N1a[j]) jobs.C1[j][(long)r * jobs.ldc1[j] + c] = v;
    else jobs.C2[j][(long)(r - jobs.N1a[j]) * jobs.ldc2[j] + c] = v;
}

template <int T1, int T2, int NP = 3>
int launch_tn(const TnArgs &a, int S, hipStream_t st) {
    const int N1 = a.N1a + a.N1b;
    const dim3 grid(((N1 + 64 * T1 - 1) / (64 * T1)) * ((a.N2 + 64 * T2 - 1) / (64 * T2)), S);
    hipLaunchKernelGGL((pw_tn_kernel<T1, T2, NP>), grid, dim3(256), 0, st, a);
    FSG_CHECK_LAUNCH("fsg_pw_tn_f32");
    return FSG_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------- C ABI
extern "C" size_t fsg_pw_weight_image_bytes(int N, int K) {
    return (size_t)((N + 31) / 32) * ((K + 15) / 16) * 3 * 1024;
}

static int weight_image(const float *W, int64_t stride_n, int64_t stride_k, int N, int K, float scale, int ks0, int KS, int pieces,
                        void *image, fsg_stream_t stream) {
    FSG_REQUIRE(W && image, "fsg_pw_weight_image_f32: NULL pointer");
    FSG_REQUIRE(N > 0 && K > 0 && ks0 >= 0 && ks0 + (K + 15) / 16 <= KS, "fsg_pw_weight_image_f32: bad shape N=%d K=%d ks0=%d KS=%d",
                N, K, ks0, KS);
    const long threads = (long)((N + 31) / 32) * ((K + 15) / 16) * 64;
    FSG_REQUIRE(pieces == 1 || pieces == 3, "fsg_pw_weight_image: pieces = %d", pieces);
    hipLaunchKernelGGL(pw_weight_image_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                       (long)stride_n, (long)stride_k, N, K, scale, ks0, KS, pieces, reinterpret_cast<u32x4 *>(image));
    FSG_CHECK_LAUNCH("fsg_pw_weight_image_f32");
    return FSG_OK;
}

extern "C" int fsg_pw_weight_image_f32(const float *W, int64_t stride_n, int64_t stride_k, int N, int K, float scale, int ks0,
                                       int KS, void *image, fsg_stream_t stream) {
    return weight_image(W, stride_n, stride_k, N, K, scale, ks0, KS, 3, image, stream);
}

// bf16 operand mode (BASELINE configs 3-5): ONE bf16 piece per operand (round-to-nearest-even), one MFMA product, fp32
// accumulation -- the same kernels with the two correction pieces compiled out.  Image: a third of the fp32-grade image's bytes.
extern "C" int fsg_pw_weight_image_bf16(const float *W, int64_t stride_n, int64_t stride_k, int N, int K, void *image,
                                        fsg_stream_t stream) {
    return weight_image(W, stride_n, stride_k, N, K, 1.0f, 0, (K + 15) / 16, 1, image, stream);
}

// C (M, N) = bf16(A) (M, K) . bf16(W)^T (+ bias), fp32 accumulation; W as fsg_pw_weight_image_bf16.  tile: 2 = 64 x 128, 3 = 64 x 64
extern "C" int fsg_pw_linear_bf16(const float *A, int64_t lda, const void *image, const float *bias, float *C, int64_t ldc, int M,
                                  int N, int K, int tile, fsg_stream_t stream) {
    FSG_REQUIRE(A && image && C, "fsg_pw_linear_bf16: NULL pointer");
    FSG_REQUIRE(M > 0 && N > 0 && K > 0 && K % 32 == 0 && lda % 4 == 0 && ((uintptr_t)A & 15) == 0,
                "fsg_pw_linear_bf16: bad shape M=%d N=%d K=%d lda=%ld (K %% 32 == 0, lda %% 4 == 0, 16-byte aligned rows)", M, N, K, (long)lda);
    RowGemmArgs a{};
    a.A1 = A; a.lda1 = lda; a.K1 = K; a.K2 = 0;
    a.Bimg = reinterpret_cast<const u32x4 *>(image);
    a.M = M; a.N = N; a.rows_per_cloud = 0;
    a.C = C; a.ldc = ldc; a.store_n0 = 0; a.bias = bias;
    hipStream_t st = (hipStream_t)stream;
    if (tile == 0) tile = (long)((M + 63) / 64) * ((N + 127) / 128) >= 256 ? 2 : 3;
    if (bias) {
        if (tile == 2) return launch_rowgemm<1, 2, PRO_NONE, PW_STORE | PW_BIAS, 1, 1>(a, st, "fsg_pw_linear_bf16");
        return launch_rowgemm<1, 1, PRO_NONE, PW_STORE | PW_BIAS, 1, 1>(a, st, "fsg_pw_linear_bf16");
    }
    if (tile == 2) return launch_rowgemm<1, 2, PRO_NONE, PW_STORE, 1, 1>(a, st, "fsg_pw_linear_bf16");
    return launch_rowgemm<1, 1, PRO_NONE, PW_STORE, 1, 1>(a, st, "fsg_pw_linear_bf16");
}

extern "C" int fsg_pw_weight_images_f32(const fsg_pw_image_jobs *jobs, fsg_stream_t stream) {
    FSG_REQUIRE(jobs && jobs->n >= 1 && jobs->n <= FSG_PW_MAX_IMAGE_JOBS, "fsg_pw_weight_images_f32: 1..%d jobs", FSG_PW_MAX_IMAGE_JOBS);
    ImageJobs k{};
    k.n = jobs->n;
    long blocks = 0;
    for (int j = 0; j < jobs->n; ++j) {
        FSG_REQUIRE(jobs->W[j] && jobs->image[j] && jobs->N[j] > 0 && jobs->K[j] > 0 && jobs->ks0[j] >= 0 &&
                        jobs->ks0[j] + (jobs->K[j] + 15) / 16 <= jobs->KS[j], "fsg_pw_weight_images_f32: bad job %d", j);
        k.W[j] = jobs->W[j]; k.sn[j] = jobs->stride_n[j]; k.sk[j] = jobs->stride_k[j]; k.N[j] = jobs->N[j]; k.K[j] = jobs->K[j];
        k.ks0[j] = jobs->ks0[j]; k.KS[j] = jobs->KS[j]; k.scale[j] = jobs->scale[j];
        k.img[j] = reinterpret_cast<u32x4 *>(jobs->image[j]);
        k.first[j] = blocks;
        blocks += (long)((jobs->N[j] + 31) / 32) * ((jobs->K[j] + 15) / 16);
    }
    k.first[jobs->n] = blocks;
    hipLaunchKernelGGL(pw_weight_images_kernel, dim3((unsigned)((blocks * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, k);
    FSG_CHECK_LAUNCH("fsg_pw_weight_images_f32");
    return FSG_OK;
}

// C (M, N) = A (M, K) . W^T (+ bias) with W given as its image (fsg_pw_weight_image_f32 of the (N, K) weight): the plain member
// of the family (tests, benchmarks, any y = x W^T + b of the models).  K % 32 == 0, lda % 4 == 0, 16-byte aligned A.
extern "C" int fsg_pw_linear_f32(const float *A, int64_t lda, const void *image, const float *bias, float *C, int64_t ldc,
                                 int M, int N, int K, int tile, fsg_stream_t stream) {
    FSG_REQUIRE(A && image && C, "fsg_pw_linear_f32: NULL pointer");
    FSG_REQUIRE(M > 0 && N > 0 && K > 0 && K % 32 == 0 && lda % 4 == 0 && ((uintptr_t)A & 15) == 0,
                "fsg_pw_linear_f32: bad shape M=%d N=%d K=%d lda=%ld (K %% 32 == 0, lda %% 4 == 0, 16-byte aligned rows)", M, N, K, (long)lda);
    RowGemmArgs a{};
    a.A1 = A; a.lda1 = lda; a.K1 = K; a.K2 = 0;
    a.Bimg = reinterpret_cast<const u32x4 *>(image);
    a.M = M; a.N = N; a.rows_per_cloud = 0;
    a.C = C; a.ldc = ldc; a.store_n0 = 0; a.bias = bias;
    hipStream_t st = (hipStream_t)stream;
    // tile: 0 = by shape; 1 = 128 x 128, 2 = 64 x 128, 3 = 64 x 64, 4 = 128 x 64
    if (tile == 0) {
        const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
        tile = t128 >= 512 ? 1 : ((long)((M + 63) / 64) * ((N + 127) / 128) >= 384 ? 2 : 3);
    }
    // 128 x 128: the single-buffered loop (two workgroups per CU) is the faster one there, see pw_rowgemm_kernel
    if (bias) {
        if (tile == 1) return launch_rowgemm<2, 2, PRO_NONE, PW_STORE | PW_BIAS, 0>(a, st, "fsg_pw_linear_f32");
        if (tile == 2) return launch_rowgemm<1, 2, PRO_NONE, PW_STORE | PW_BIAS>(a, st, "fsg_pw_linear_f32");
        if (tile == 4) return launch_rowgemm<2, 1, PRO_NONE, PW_STORE | PW_BIAS>(a, st, "fsg_pw_linear_f32");
        return launch_rowgemm<1, 1, PRO_NONE, PW_STORE | PW_BIAS>(a, st, "fsg_pw_linear_f32");
    }
    if (tile == 1) return launch_rowgemm<2, 2, PRO_NONE, PW_STORE, 0>(a, st, "fsg_pw_linear_f32");
    // tools/bench_pw.py: the other loop structure of each tile (11 = 128 x 128 double-buffered, 12-14 = single-buffered)
    if (tile == 11) return launch_rowgemm<2, 2, PRO_NONE, PW_STORE, 1>(a, st, "fsg_pw_linear_f32");
    if (tile == 12) return launch_rowgemm<1, 2, PRO_NONE, PW_STORE, 0>(a, st, "fsg_pw_linear_f32");
    if (tile == 13) return launch_rowgemm<1, 1, PRO_NONE, PW_STORE, 0>(a, st, "fsg_pw_linear_f32");
    if (tile == 14) return launch_rowgemm<2, 1, PRO_NONE, PW_STORE, 0>(a, st, "fsg_pw_linear_f32");
    if (tile == 2) return launch_rowgemm<1, 2, PRO_NONE, PW_STORE>(a, st, "fsg_pw_linear_f32");
    if (tile == 4) return launch_rowgemm<2, 1, PRO_NONE, PW_STORE>(a, st, "fsg_pw_linear_f32");
    return launch_rowgemm<1, 1, PRO_NONE, PW_STORE>(a, st, "fsg_pw_linear_f32");
}

// ---- generic members of the family behind the fused DGCNN head (functional.py composes them; include/fsg_hip.h documents
//      every argument).  Tile codes: 1 = 128 x 128, 2 = 64 x 128, 3 = 64 x 64, 4 = 128 x 64 (rows x columns of C).
extern "C" int fsg_pw_tile_rows(int tile) { return (tile == 1 || tile == 4) ? 128 : 64; }     // tiles 2, 3, 5: 64 rows

static int check_rowgemm(const fsg_pw_rowgemm_args *a, int pro, int epi, int BM) {
    FSG_REQUIRE(a && a->A1 && a->Bimg, "fsg_pw_rowgemm_f32: NULL pointer");
    FSG_REQUIRE(a->M > 0 && a->N > 0 && a->K1 > 0 && a->K1 % 32 == 0 && a->K2 >= 0 && a->K2 % 32 == 0 && a->lda1 % 4 == 0 &&
                    ((uintptr_t)a->A1 & 15) == 0,
                "fsg_pw_rowgemm_f32: bad shape M=%d N=%d K1=%d K2=%d lda1=%ld", a->M, a->N, a->K1, a->K2, (long)a->lda1);
    FSG_REQUIRE(a->K2 == 0 || (a->A2 && a->lda2 % 4 == 0 && ((uintptr_t)a->A2 & 15) == 0), "fsg_pw_rowgemm_f32: bad segment 2");
    FSG_REQUIRE(pro == PRO_NONE || (a->alpha && a->delta), "fsg_pw_rowgemm_f32: prologue tables missing");
    FSG_REQUIRE(pro != PRO_BNBWD || (a->Y1 && a->P && a->Q && ((uintptr_t)a->Y1 & 15) == 0), "fsg_pw_rowgemm_f32: BNBWD needs Y1, P, Q");
    FSG_REQUIRE(!(epi & PW_STORE) || a->C, "fsg_pw_rowgemm_f32: STORE needs C");
    FSG_REQUIRE(!(epi & PW_BIAS) || a->bias, "fsg_pw_rowgemm_f32: BIAS needs bias");
    if (epi & (PW_STATS | PW_SEL | PW_BWDSTATS)) {
        FSG_REQUIRE(a->M % BM == 0 && (a->rows_per_cloud == 0 || a->rows_per_cloud % BM == 0),
                    "fsg_pw_rowgemm_f32: reducing epilogues need M (%d) and rows_per_cloud (%d) to be multiples of the tile rows %d",
                    a->M, a->rows_per_cloud, BM);
        FSG_REQUIRE(!(epi & PW_STATS) || a->rec, "fsg_pw_rowgemm_f32: STATS needs rec");
        FSG_REQUIRE(!(epi & PW_SEL) || (a->sgn && a->sel_val && a->sel_arg && a->sel_n > 0 && a->rows_per_cloud > 0),
                    "fsg_pw_rowgemm_f32: SEL needs sgn, sel_val, sel_arg, sel_n, rows_per_cloud");
        FSG_REQUIRE(!(epi & PW_BWDSTATS) || (a->Yp && a->ealpha && a->edelta && a->emu && a->er && a->rec2),
                    "fsg_pw_rowgemm_f32: BWDSTATS needs Yp, its tables and rec2");
    }
    if ((pro != PRO_NONE && a->tstride != 0) || ((epi & PW_BWDSTATS) && a->etstride != 0))
        FSG_REQUIRE(a->rows_per_cloud > 0 && a->rows_per_cloud % BM == 0, "fsg_pw_rowgemm_f32: per-cloud tables need rows_per_cloud %% %d == 0", BM);
    return FSG_OK;
}

static RowGemmArgs to_kernel_args(const fsg_pw_rowgemm_args *a) {
    RowGemmArgs k{};
    k.A1 = a->A1; k.Y1 = a->Y1; k.A2 = a->A2; k.lda1 = a->lda1; k.lda2 = a->lda2; k.K1 = a->K1; k.K2 = a->K2;
    k.Bimg = reinterpret_cast<const u32x4 *>(a->Bimg);
    k.M = a->M; k.N = a->N; k.rows_per_cloud = a->rows_per_cloud;
    k.alpha = a->alpha; k.delta = a->delta; k.P = a->P; k.Q = a->Q; k.tstride = a->tstride; k.slope = a->slope;
    k.C = a->C; k.ldc = a->ldc; k.store_n0 = a->store_n0; k.bias = a->bias; k.rec = a->rec;
    k.sgn = a->sgn; k.sel_val = a->sel_val; k.sel_arg = a->sel_arg; k.sel_n = a->sel_n;
    k.Yp = a->Yp; k.ldyp = a->ldyp; k.ealpha = a->ealpha; k.edelta = a->edelta; k.emu = a->emu; k.er = a->er;
    k.etstride = a->etstride; k.rec2 = a->rec2;
    return k;
}

extern "C" int fsg_pw_rowgemm_f32(const fsg_pw_rowgemm_args *a, int pro, int epi, int tile, fsg_stream_t stream) {
    FSG_REQUIRE(tile >= 1 && tile <= 5, "fsg_pw_rowgemm_f32: tile %d not in 1..5", tile);
    const int rc = check_rowgemm(a, pro, epi, fsg_pw_tile_rows(tile));
    if (rc != FSG_OK) return rc;
    const RowGemmArgs k = to_kernel_args(a);
    hipStream_t st = (hipStream_t)stream;
    const int key = pro * 1000 + epi * 10 + tile;
#define PW_CASE(PRO, EPI, TILE, WMv, WNv) \
    case (PRO) * 1000 + (EPI) * 10 + (TILE): return launch_rowgemm<WMv, WNv, PRO, EPI>(k, st, "fsg_pw_rowgemm_f32")
    switch (key) {
        PW_CASE(PRO_NONE, PW_STORE, 1, 2, 2);
        PW_CASE(PRO_NONE, PW_STORE, 2, 1, 2);
        PW_CASE(PRO_NONE, PW_STORE, 3, 1, 1);
        PW_CASE(PRO_NONE, PW_STORE, 4, 2, 1);
        PW_CASE(PRO_NONE, PW_STORE | PW_BIAS, 1, 2, 2);
        PW_CASE(PRO_NONE, PW_STORE | PW_BIAS, 2, 1, 2);
        PW_CASE(PRO_NONE, PW_STORE | PW_BIAS, 3, 1, 1);
        PW_CASE(PRO_NONE, PW_STORE | PW_BIAS, 4, 2, 1);
        PW_CASE(PRO_NONE, PW_STORE | PW_STATS, 1, 2, 2);
        PW_CASE(PRO_NONE, PW_STORE | PW_STATS, 2, 1, 2);
        case PRO_NONE * 1000 + (PW_STORE | PW_STATS | PW_SEL) * 10 + 1:      // the widest product: single-buffered loop, 2 workgroups / CU
            return launch_rowgemm<2, 2, PRO_NONE, PW_STORE | PW_STATS | PW_SEL, 0>(k, st, "fsg_pw_rowgemm_f32");
        PW_CASE(PRO_NONE, PW_STORE | PW_STATS | PW_SEL, 2, 1, 2);
        PW_CASE(PRO_BNACT, PW_STORE | PW_STATS, 1, 2, 2);
        PW_CASE(PRO_BNACT, PW_STORE | PW_STATS, 2, 1, 2);
        PW_CASE(PRO_BNACT, PW_STORE | PW_STATS, 3, 1, 1);
        PW_CASE(PRO_BNACT, PW_STORE | PW_BIAS, 3, 1, 1);
        PW_CASE(PRO_BNACT, PW_STORE, 3, 1, 1);
        PW_CASE(PRO_BNBWD, PW_STORE | PW_BWDSTATS, 1, 2, 2);
        PW_CASE(PRO_BNBWD, PW_STORE | PW_BWDSTATS, 2, 1, 2);
        PW_CASE(PRO_BNBWD, PW_STORE | PW_BWDSTATS, 3, 1, 1);
        PW_CASE(PRO_BNBWD, PW_STORE | PW_BIAS, 1, 2, 2);
        PW_CASE(PRO_BNBWD, PW_STORE | PW_BIAS, 3, 1, 1);
        PW_CASE(PRO_BNBWD, PW_STORE | PW_BIAS, 4, 2, 1);
        PW_CASE(PRO_BNBWD, PW_STORE, 3, 1, 1);
        // whole-width tile 64 x 192: the prologue + split of an A row is done once instead of once per column tile (a 64 x 256
        // tile for the 256-wide products holds one workgroup per CU and was slower: 24.9 vs 20.7 us)
        PW_CASE(PRO_BNBWD, PW_STORE | PW_BIAS, 5, 1, 3);
        default: break;
    }
#undef PW_CASE
    fsg_set_error("fsg_pw_rowgemm_f32: combination prologue %d / epilogue %d / tile %d is not instantiated", pro, epi, tile);
    return FSG_ERR_UNSUPPORTED;
}

extern "C" size_t fsg_pw_tn_workspace_bytes(int N1, int N2, int M, int rows_per_slice) {
    if (rows_per_slice <= 0) return 0;
    return sizeof(float) * (size_t)((M + rows_per_slice - 1) / rows_per_slice) * N1 * N2;
}

extern "C" int fsg_pw_tn_f32(const fsg_pw_tn_args *a, int tile, void *workspace, size_t workspace_bytes, float *C1,
                             int64_t ldc1, float *C2, int64_t ldc2, fsg_stream_t stream) {
    FSG_REQUIRE(a && a->L1 && a->R && workspace, "fsg_pw_tn_f32: NULL pointer");
    const int N1 = a->N1a + a->N1b + (a->ones ? 1 : 0);
    FSG_REQUIRE(a->M > 0 && a->N1a > 0 && a->N1b >= 0 && a->N2 > 0 && a->rows_per_slice > 0 && a->rows_per_slice % 32 == 0,
                "fsg_pw_tn_f32: bad shape M=%d N1=%d+%d N2=%d rows_per_slice=%d", a->M, a->N1a, a->N1b, a->N2, a->rows_per_slice);
    FSG_REQUIRE(a->N1b == 0 || (a->N1a % 64 == 0 && a->L2 && (C2 || !C1)), "fsg_pw_tn_f32: two left segments need N1a %% 64 == 0, L2 and C2");
    FSG_REQUIRE(a->lpro == PRO_NONE || (a->lpro == PRO_BNBWD && a->LY1 && a->lalpha && a->ldelta && a->lP && a->lQ),
                "fsg_pw_tn_f32: left prologue %d needs LY1 and its tables", a->lpro);
    FSG_REQUIRE(a->rpro == PRO_NONE || (a->rpro == PRO_BNACT && a->ralpha && a->rdelta), "fsg_pw_tn_f32: right prologue %d", a->rpro);
    const bool per_cloud = (a->lpro != PRO_NONE && a->lts != 0) || (a->rpro != PRO_NONE && a->rts != 0);
    FSG_REQUIRE(!per_cloud || (a->rows_per_cloud > 0 && a->rows_per_cloud % a->rows_per_slice == 0),
                "fsg_pw_tn_f32: per-cloud tables need rows_per_cloud %% rows_per_slice == 0");
    const int S = (a->M + a->rows_per_slice - 1) / a->rows_per_slice;
    FSG_REQUIRE(workspace_bytes >= fsg_pw_tn_workspace_bytes(N1, a->N2, a->M, a->rows_per_slice), "fsg_pw_tn_f32: workspace too small");
    TnArgs k{};
    k.L1 = a->L1; k.LY1 = a->LY1; k.L2 = a->L2; k.ldl1 = a->ldl1; k.ldl2 = a->ldl2; k.N1a = a->N1a; k.N1b = a->N1b; k.lpro = a->lpro;
    k.lalpha = a->lalpha; k.ldelta = a->ldelta; k.lP = a->lP; k.lQ = a->lQ; k.lts = a->lts;
    k.R = a->R; k.ldr = a->ldr; k.N2 = a->N2; k.rpro = a->rpro; k.ralpha = a->ralpha; k.rdelta = a->rdelta; k.rts = a->rts;
    k.slope = a->slope; k.M = a->M; k.rows_per_cloud = a->rows_per_cloud; k.rows_per_slice = a->rows_per_slice;
    k.ones = a->ones ? 1 : 0;
    k.part = reinterpret_cast<float *>(workspace);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    // tile: 1 = 128 x 128, 2 = 64 x 128, 3 = 64 x 64, 5 = 128 x 192
    if (tile == 1) rc = launch_tn<2, 2>(k, S, st);
    else if (tile == 2) rc = launch_tn<1, 2>(k, S, st);
    else if (tile == 3) rc = launch_tn<1, 1>(k, S, st);
    else if (tile == 5) rc = launch_tn<2, 3>(k, S, st);
    else { fsg_set_error("fsg_pw_tn_f32: tile %d", tile); return FSG_ERR_ARG; }
    if (rc != FSG_OK) return rc;
    if (!C1) return FSG_OK;          // the caller folds the slices later (fsg_pw_tn_reduce_f32: several products, one launch)
    const long total = (long)N1 * a->N2;
    hipLaunchKernelGGL(pw_tn_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, k.part, S, N1, a->N2, a->N1a,
                       C1, (long)ldc1, C2, (long)ldc2);
    FSG_CHECK_LAUNCH("fsg_pw_tn_f32/reduce");
    return FSG_OK;
}

// bf16 operand mode of the row contraction (plain operands only): dW = bf16(dY)^T bf16(X), fp32 accumulation
extern "C" int fsg_pw_tn_bf16(const fsg_pw_tn_args *a, int tile, void *workspace, size_t workspace_bytes, float *C1, int64_t ldc1,
                              fsg_stream_t stream) {
    FSG_REQUIRE(a && a->L1 && a->R && workspace && C1, "fsg_pw_tn_bf16: NULL pointer");
    FSG_REQUIRE(a->M > 0 && a->N1a > 0 && a->N1b == 0 && a->N2 > 0 && a->rows_per_slice > 0 && a->rows_per_slice % 32 == 0 &&
                    a->lpro == PRO_NONE && a->rpro == PRO_NONE, "fsg_pw_tn_bf16: plain single-segment operands only");
    const int S = (a->M + a->rows_per_slice - 1) / a->rows_per_slice;
    FSG_REQUIRE(workspace_bytes >= fsg_pw_tn_workspace_bytes(a->N1a, a->N2, a->M, a->rows_per_slice), "fsg_pw_tn_bf16: workspace too small");
    TnArgs k{};
    k.L1 = a->L1; k.ldl1 = a->ldl1; k.N1a = a->N1a; k.N1b = 0; k.lpro = PRO_NONE;
    k.R = a->R; k.ldr = a->ldr; k.N2 = a->N2; k.rpro = PRO_NONE; k.slope = 0.f;
    k.M = a->M; k.rows_per_cloud = 0; k.rows_per_slice = a->rows_per_slice;
    k.part = reinterpret_cast<float *>(workspace);
    hipStream_t st = (hipStream_t)stream;
    const int rc = tile == 2 ? launch_tn<1, 2, 1>(k, S, st) : launch_tn<1, 1, 1>(k, S, st);
    if (rc != FSG_OK) return rc;
    const long total = (long)a->N1a * a->N2;
    hipLaunchKernelGGL(pw_tn_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, k.part, S, a->N1a, a->N2, a->N1a,
                       C1, (long)ldc1, (float *)nullptr, 0L);
    FSG_CHECK_LAUNCH("fsg_pw_tn_bf16/reduce");
    return FSG_OK;
}

extern "C" int fsg_pw_tn_reduce_f32(const fsg_pw_tn_reduce_jobs *jobs, fsg_stream_t stream) {
    FSG_REQUIRE(jobs && jobs->n >= 1 && jobs->n <= FSG_PW_MAX_REDUCE_JOBS, "fsg_pw_tn_reduce_f32: 1..%d jobs", FSG_PW_MAX_REDUCE_JOBS);
    TnReduceJobs k{};
    k.n = jobs->n;
    long total = 0;
    for (int j = 0; j < jobs->n; ++j) {
        FSG_REQUIRE(jobs->workspace[j] && jobs->C1[j] && jobs->S[j] > 0 && jobs->N1[j] > 0 && jobs->N2[j] > 0 &&
                        (jobs->N1a[j] >= jobs->N1[j] || jobs->C2[j]), "fsg_pw_tn_reduce_f32: bad job %d", j);
        k.part[j] = reinterpret_cast<const float *>(jobs->workspace[j]);
        k.C1[j] = jobs->C1[j]; k.C2[j] = jobs->C2[j]; k.ldc1[j] = jobs->ldc1[j]; k.ldc2[j] = jobs->ldc2[j];
        k.S[j] = jobs->S[j]; k.N1[j] = jobs->N1[j]; k.N2[j] = jobs->N2[j]; k.N1a[j] = jobs->N1a[j];
        k.first[j] = total;
        total += (long)jobs->N1[j] * jobs->N2[j];
    }
    k.first[jobs->n] = total;
    hipLaunchKernelGGL(pw_tn_reduce_many_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, k);
    FSG_CHECK_LAUNCH("fsg_pw_tn_reduce_f32");
    return FSG_OK;
}

extern "C" int fsg_pw_bn_finalize_f32(const float *rec, int R, int ldn, int c0, int C, const float *shift, int B, int training,
                                      const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                                      float *running_var, float *mean, float *invstd, float *alpha, float *delta, float *emu,
                                      float *cloud_mean, const float *gfeat, const float *Wglob, int64_t ldwg, int CG,
                                      float *shift_out, fsg_stream_t stream) {
    FSG_REQUIRE(gamma && beta && mean && invstd && alpha && delta, "fsg_pw_bn_finalize_f32: NULL pointer");
    FSG_REQUIRE(!gfeat || (Wglob && shift_out && shift == shift_out && CG > 0 && B > 0 && B <= 64),
                "fsg_pw_bn_finalize_f32: an in-kernel shift needs gfeat, Wglob, CG and shift == shift_out");
    FSG_REQUIRE(C > 0 && (!training || (rec && R > 0 && ldn >= c0 + C)) && (!shift || (B > 0 && R % B == 0)) &&
                    (!cloud_mean || (B > 0 && R % B == 0)),
                "fsg_pw_bn_finalize_f32: bad shape R=%d ldn=%d c0=%d C=%d B=%d", R, ldn, c0, C, B);
    hipLaunchKernelGGL(pw_bn_finalize_kernel, dim3((C + FC - 1) / FC), dim3(256), 0, (hipStream_t)stream, MaxFinish{}, rec, R, ldn, c0, C,
                       shift, B, training, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, alpha, delta, emu,
                       cloud_mean, gfeat, Wglob, (long)ldwg, CG, shift_out);
    FSG_CHECK_LAUNCH("fsg_pw_bn_finalize_f32");
    return FSG_OK;
}

// fsg_pw_bn_finalize_f32 (no shift) + fsg_pw_max_finish_f32 in ONE launch: the statistics of the BatchNorm in front of a global
// max-pool and the pool's finish from the SEL records of the same product (models/dgcnn.py:134-137,156).  B <= 64.
extern "C" int fsg_pw_bn_finalize_max_f32(const float *rec, int R, int ldn, int c0, int C, int B, int training, const float *gamma,
                                          const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                                          float *mean, float *invstd, float *alpha, float *delta, const float *sel_val,
                                          const int32_t *sel_arg, const float *sgn, int tiles, float slope, float *out, float *ysel,
                                          int32_t *arg, fsg_stream_t stream) {
    FSG_REQUIRE(gamma && beta && mean && invstd && alpha && delta && sel_val && sel_arg && sgn && out && ysel && arg,
                "fsg_pw_bn_finalize_max_f32: NULL pointer");
    FSG_REQUIRE(C > 0 && B > 0 && B <= 64 && tiles > 0 && (!training || (rec && R > 0 && ldn >= c0 + C)),
                "fsg_pw_bn_finalize_max_f32: bad shape R=%d ldn=%d c0=%d C=%d B=%d tiles=%d", R, ldn, c0, C, B, tiles);
    const MaxFinish mf{sel_val, sel_arg, sgn, tiles, slope, out, ysel, arg};
    hipLaunchKernelGGL(pw_bn_finalize_kernel, dim3((C + FC - 1) / FC), dim3(256), 0, (hipStream_t)stream, mf, rec, R, ldn, c0, C,
                       nullptr, B, training, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, alpha, delta, nullptr,
                       nullptr, nullptr, nullptr, 0L, 0, nullptr);
    FSG_CHECK_LAUNCH("fsg_pw_bn_finalize_max_f32");
    return FSG_OK;
}

extern "C" int fsg_pw_cloud_linear_f32(const float *x, const float *W, int64_t ldw, int B, int C0, int CG, float *out,
                                       fsg_stream_t stream) {
    FSG_REQUIRE(x && W && out && B > 0 && C0 > 0 && CG > 0, "fsg_pw_cloud_linear_f32: bad arguments");
    hipLaunchKernelGGL(pw_cloud_linear_kernel, dim3((B * C0 + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, W, (long)ldw, B, C0, CG, out);
    FSG_CHECK_LAUNCH("fsg_pw_cloud_linear_f32");
    return FSG_OK;
}

extern "C" int fsg_pw_max_finish_f32(const float *sel_val, const int32_t *sel_arg, const float *sgn, const float *alpha,
                                     const float *delta, int B, int tiles, int C, float slope, float *out, float *ysel,
                                     int32_t *arg, fsg_stream_t stream) {
    FSG_REQUIRE(sel_val && sel_arg && sgn && alpha && delta && out && ysel && arg && B > 0 && tiles > 0 && C > 0,
                "fsg_pw_max_finish_f32: bad arguments");
    hipLaunchKernelGGL(pw_max_finish_kernel, dim3((C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, sel_val, sel_arg, sgn,
                       alpha, delta, tiles, C, slope, out, ysel, arg);
    FSG_CHECK_LAUNCH("fsg_pw_max_finish_f32");
    return FSG_OK;
}

extern "C" int fsg_pw_bnbwd_finalize_f32(const float *rec2, int R, int C, int B, int64_t M, int training, const float *alpha,
                                         const float *invstd, const float *emu, int emu_per_cloud, const float *cloud_mean,
                                         float *dbeta, float *dgamma, float *P, float *Q, float *dc, fsg_stream_t stream) {
    FSG_REQUIRE(rec2 && alpha && invstd && emu && dbeta && dgamma && P && Q && R > 0 && C > 0 && M > 0 && B > 0 && R % B == 0 &&
                    (!dc || (cloud_mean && emu_per_cloud && M % B == 0)),
                "fsg_pw_bnbwd_finalize_f32: bad arguments");
    hipLaunchKernelGGL(pw_bnbwd_finalize_kernel, dim3((C + FC - 1) / FC), dim3(256), 0, (hipStream_t)stream, rec2, R, C, B, (long)M,
                       training, alpha, invstd, emu, emu_per_cloud, cloud_mean, dbeta, dgamma, P, Q, dc);
    FSG_CHECK_LAUNCH("fsg_pw_bnbwd_finalize_f32");
    return FSG_OK;
}

extern "C" int fsg_pw_logits_bwd_f32(const float *g, int classes, const float *W3, const float *y, const float *alpha,
                                     const float *delta, const float *mean, const float *invstd, int64_t M, int C, float slope,
                                     float *da, float *rec2, fsg_stream_t stream) {
    FSG_REQUIRE(g && W3 && y && alpha && delta && mean && invstd && da && rec2 && M > 0 && C > 0 && C % 64 == 0 && classes >= 1 &&
                    classes <= 8, "fsg_pw_logits_bwd_f32: bad arguments (C %% 64 == 0, 1 <= classes <= 8)");
    hipLaunchKernelGGL(pw_logits_bwd_kernel, dim3((unsigned)((M + LB_ROWS - 1) / LB_ROWS)), dim3(256), 0, (hipStream_t)stream, g, classes, W3,
                       y, alpha, delta, mean, invstd, (long)M, C, slope, da, rec2);
    FSG_CHECK_LAUNCH("fsg_pw_logits_bwd_f32");
    return FSG_OK;
}

extern "C" int fsg_pw_gf_prep_f32(const float *dc, const float *W0g, int64_t ldw0, int C0, const float *gfeat, float *dW0g,
                                  int64_t lddw0, const float *dg, const float *ysel, const float *alpha, const float *delta,
                                  const float *mean, const float *invstd, int B, int C, int64_t M, int training, float slope,
                                  float *dbeta, float *dgamma, float *P, float *Q, float *coef, const float *W, int64_t ldw, int K,
                                  float *Wq, int64_t ldwq, fsg_stream_t stream) {
    FSG_REQUIRE(!Wq || (W && K > 0 && ldwq >= K + 1 && dc), "fsg_pw_gf_prep_f32: Wq needs W, K, ldwq >= K + 1 and the dc form");
    FSG_REQUIRE(ysel && alpha && delta && mean && invstd && dbeta && dgamma && P && Q && coef && B > 0 && B <= 32 && C > 0 && M > 0,
                "fsg_pw_gf_prep_f32: bad arguments (B <= 32)");
    FSG_REQUIRE((dc && W0g && gfeat && dW0g && C0 > 0 && (size_t)(B <= 8 ? 8 : 32) * C0 * 4 <= 32 * 1024) || (!dc && dg),
                "fsg_pw_gf_prep_f32: either dc + W0g + gfeat + dW0g (C0 <= 1024 up to 8 clouds, <= 256 up to 32) or dg");
#define FSG_GF_PREP(MAXB, WAVES)                                                                                                     \
    do {                                                                                                                         \
        hipLaunchKernelGGL((pw_gf_prep_kernel<MAXB, WAVES>), dim3((C + 63) / 64), dim3(64 * WAVES),                                \
                           dc ? sizeof(float) * (MAXB * C0 + WAVES * MAXB * 64) : 0, (hipStream_t)stream, dc, W0g, (long)ldw0, C0, gfeat, \
                           dW0g, (long)lddw0, dg, ysel, alpha, delta, mean, invstd, B, C, (long)M, training, slope, dbeta, dgamma,  \
                           P, Q, coef, W, (long)ldw, K, Wq, (long)ldwq);                                                           \
        FSG_CHECK_LAUNCH("fsg_pw_gf_prep_f32");                                                                                    \
        if (dc) {                                                                                                                \
            hipLaunchKernelGGL((pw_outer_kernel<MAXB>), dim3((C + 255) / 256, (C0 + 7) / 8), dim3(256), 0, (hipStream_t)stream, dc, \
                               gfeat, B, C0, C, dW0g, (long)lddw0, Q, P, W, (long)ldw, K, Wq, (long)ldwq);                         \
            FSG_CHECK_LAUNCH("fsg_pw_gf_prep_f32/outer");                                                                          \
        }                                                                                                                        \
    } while (0)
    if (B <= 8) FSG_GF_PREP(8, 16);
    else FSG_GF_PREP(32, 4);
#undef FSG_GF_PREP
    return FSG_OK;
}

static int scatter_cp2(int C) {
    int Cp2 = 1;
    while (Cp2 < C) Cp2 <<= 1;
    return Cp2;
}

extern "C" size_t fsg_pw_scatter_rows_workspace_bytes(int B, int C) { return sizeof(unsigned) * (size_t)(B > 0 ? B : 0) * scatter_cp2(C); }

extern "C" int fsg_pw_scatter_rows_f32(const float *coef, const int32_t *arg, const float *W, int64_t ldw, int B, int C, int K,
                                       int Npts, float *dX, int64_t ldx, void *workspace, fsg_stream_t stream) {
    FSG_REQUIRE(coef && arg && W && dX && workspace && B > 0 && C > 0 && C <= 4096 && K > 0 && Npts > 0 && Npts <= (1 << 20),
                "fsg_pw_scatter_rows_f32: bad arguments (C <= 4096, Npts <= 2^20)");
    const int Cp2 = scatter_cp2(C);
    unsigned *sorted = reinterpret_cast<unsigned *>(workspace);
    if (Cp2 == 1024) hipLaunchKernelGGL(pw_sort_sel_1024_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, arg, C, sorted);
    else hipLaunchKernelGGL(pw_sort_sel_kernel, dim3(B), dim3(1024), sizeof(unsigned) * Cp2, (hipStream_t)stream, arg, C, Cp2, sorted);
    FSG_CHECK_LAUNCH("fsg_pw_scatter_rows_f32/sort");
    hipLaunchKernelGGL(pw_scatter_rows_kernel, dim3(B, (C + SC_CHUNK - 1) / SC_CHUNK), dim3(256), 0, (hipStream_t)stream, coef,
                       sorted, W, (long)ldw, C, Cp2, K, Npts, dX, (long)ldx);
    FSG_CHECK_LAUNCH("fsg_pw_scatter_rows_f32");
    return FSG_OK;
}

extern "C" int fsg_pw_gf_dw_f32(const float *coef, const int32_t *arg, const float *X, int64_t ldx, const float *s,
                                const float *W, int64_t ldw, const float *G, const float *P, const float *Q, int B, int C, int K,
                                int Npts, float *dW, int64_t lddw, fsg_stream_t stream) {
    FSG_REQUIRE(coef && arg && X && s && W && G && P && Q && dW && B > 0 && C > 0 && K > 0 && Npts > 0, "fsg_pw_gf_dw_f32: bad arguments");
    FSG_REQUIRE(K <= 4096, "fsg_pw_gf_dw_f32: K = %d > 4096", K);
    hipLaunchKernelGGL(pw_gf_dw_kernel, dim3((C + 3) / 4), dim3(256), sizeof(float) * (4 * K + 16 * 64), (hipStream_t)stream, coef, arg, X,
                       (long)ldx, s, W, (long)ldw, G, P, Q, B, C, K, Npts, dW, (long)lddw);
    FSG_CHECK_LAUNCH("fsg_pw_gf_dw_f32");
    return FSG_OK;
}

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
                                                              float scale, int ks0, int KS, u32x4 *__restrict__ img) {
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
    u32x4 *o = img + (((long)nb * KS + ks0 + ks) * 3) * 64 + lane;
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

template <int WM, int WN, int PRO, int EPI>
__global__ __launch_bounds__(256) void pw_rowgemm_kernel(const RowGemmArgs p) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int AI = BM / 64;                       // A rows per thread and iteration
    constexpr int BU = BN * 12 / 256;                 // 16-byte units of the B image per thread and iteration
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *Aimg = reinterpret_cast<u32x4 *>(smem);                            // [BM/32][2][3][64]
    u32x4 *Bimg = Aimg + (BM / 32) * 2 * 3 * 64;                              // [BN/32][2][3][64]
    float *tab = reinterpret_cast<float *>(Bimg + (BN / 32) * 2 * 3 * 64);    // PRO tables: [4][K1]; epilogue scratch after the loop
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
            const int u = tid + 256 * q, jb = u / 384, rem = u - jb * 384;
            rb[q] = jb < nbl ? bsrc[(long)jb * KS * 192 + rem] : u32x4{0u, 0u, 0u, 0u};
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

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
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row0 + (wm * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    if (row < p.M && col < p.N) p.C[(long)row * p.ldc + (col - p.store_n0)] = acc[i][j][e] + bv;
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
                const float sg = col < p.sel_n ? p.sgn[col] : 1.f;
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

template <int WM, int WN, int PRO, int EPI>
int launch_rowgemm(const RowGemmArgs &a, hipStream_t st, const char *name) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    const int MT = (a.M + BM - 1) / BM, NT = (a.N + BN - 1) / BN;
    const int grid = ((MT + 7) / 8) * 8 * NT;
    size_t lds = ((size_t)(BM / 32) * 6 + (size_t)(BN / 32) * 6) * 1024 + sizeof(float) * 4 * (size_t)a.K1;
    const size_t scratch = sizeof(float) * 2 * 2 * WN * 32;
    if (lds < scratch) lds = scratch;
    static size_t granted = 64 * 1024;
    if (lds > granted) {
        if (hipFuncSetAttribute((const void *)pw_rowgemm_kernel<WM, WN, PRO, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            fsg_set_error("%s: cannot raise dynamic LDS to %zu", name, lds);
            return FSG_ERR_HIP;
        }
        granted = lds;
    }
    hipLaunchKernelGGL((pw_rowgemm_kernel<WM, WN, PRO, EPI>), dim3(grid), dim3(256), lds, st, a);
    FSG_CHECK_LAUNCH(name);
    return FSG_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------- C ABI
extern "C" size_t fsg_pw_weight_image_bytes(int N, int K) {
    return (size_t)((N + 31) / 32) * ((K + 15) / 16) * 3 * 1024;
}

extern "C" int fsg_pw_weight_image_f32(const float *W, int64_t stride_n, int64_t stride_k, int N, int K, float scale, int ks0,
                                       int KS, void *image, fsg_stream_t stream) {
    FSG_REQUIRE(W && image, "fsg_pw_weight_image_f32: NULL pointer");
    FSG_REQUIRE(N > 0 && K > 0 && ks0 >= 0 && ks0 + (K + 15) / 16 <= KS, "fsg_pw_weight_image_f32: bad shape N=%d K=%d ks0=%d KS=%d",
                N, K, ks0, KS);
    const long threads = (long)((N + 31) / 32) * ((K + 15) / 16) * 64;
    hipLaunchKernelGGL(pw_weight_image_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                       (long)stride_n, (long)stride_k, N, K, scale, ks0, KS, reinterpret_cast<u32x4 *>(image));
    FSG_CHECK_LAUNCH("fsg_pw_weight_image_f32");
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
    if (bias) {
        if (tile == 1) return launch_rowgemm<2, 2, PRO_NONE, PW_STORE | PW_BIAS>(a, st, "fsg_pw_linear_f32");
        if (tile == 2) return launch_rowgemm<1, 2, PRO_NONE, PW_STORE | PW_BIAS>(a, st, "fsg_pw_linear_f32");
        if (tile == 4) return launch_rowgemm<2, 1, PRO_NONE, PW_STORE | PW_BIAS>(a, st, "fsg_pw_linear_f32");
        return launch_rowgemm<1, 1, PRO_NONE, PW_STORE | PW_BIAS>(a, st, "fsg_pw_linear_f32");
    }
    if (tile == 1) return launch_rowgemm<2, 2, PRO_NONE, PW_STORE>(a, st, "fsg_pw_linear_f32");
    if (tile == 2) return launch_rowgemm<1, 2, PRO_NONE, PW_STORE>(a, st, "fsg_pw_linear_f32");
    if (tile == 4) return launch_rowgemm<2, 1, PRO_NONE, PW_STORE>(a, st, "fsg_pw_linear_f32");
    return launch_rowgemm<1, 1, PRO_NONE, PW_STORE>(a, st, "fsg_pw_linear_f32");
}

// Fused EdgeConv with TWO shared-MLP layers (ec1 of DGCNNSeg: 2C -> 64 -> 64; the spatial transformer's
// 6 -> 64 -> 128), forward and backward, no per-edge tensor in HBM on the forward and one (du1) on the backward.
// include/fsg_hip.h: fsg_edgeconv2_{fwd,bwd}_f32.  Replaces models/dgcnn.py:234-241 for len(shared_mlp) == 2.
//
// Layer 1 is decomposed exactly like the one-layer kernel (edgeconv.hip): y1(i,s) = P_j + Q_i from per-point rows.
// Layer 2 is a genuine per-edge contraction y2 = W2 z1, z1 = LeakyReLU(BN1(y1)): a (B N k) x 64 x C2 GEMM that
// runs on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 fma chains) out of LDS:
//   a workgroup owns TP points = R = TP*k edge rows (padded to a multiple of 32, <= RMAX);
//   phase 1  lanes = channels: gather P rows, y1 -> LDS tile Y1[R][65]
//   phase 2  waves take (32-row, 32-column) output tiles; A operand z1 = f(a1*y1+b1) is formed on the fly from Y1,
//            B operand W2 comes from an LDS copy; 32 MFMAs per tile
//   phase 3  accumulators -> LDS tile Y2[R][C2+1] (aliases Y1 after a barrier)
//   phase 4  lanes = channels: BN2 statistics (shifted sums), max/min selection over the k rows of each point.
// BN2 + LeakyReLU are applied to the selected values by ec1_apply_kernel (monotonicity, see edgeconv.hip).
// Backward recomputes y1, z1, y2 per tile, forms dy2 = a2 (h2[s = arg] - db2 - yhat2 dg2) in LDS, and runs two more
// MFMA products per tile: dz1 = dy2 W2 (-> du1 = dz1 f'(u1), written once to HBM, and the dbeta1/dgamma1 sums) and
// dW2 += dy2^T z1 (accumulated in registers over all tiles of the workgroup).  dP/dQ then come from the
// reverse-graph gather over du1 (no float atomics anywhere).
#include <stdlib.h>

#include "fsg_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// bf16 operand mode (fsg_edgeconv2_{fwd,bwd}_bf16): the per-edge products run on v_mfma_f32_32x32x16_bf16 -- operands
// rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on their way out of LDS, fp32 accumulation, everything else
// (gathers, BatchNorm statistics, selection, stored tensors) unchanged in fp32.  Lane (ql = lane % 32, half = lane / 32)
// holds the eight k-values 16 s + 8 half + 0..7 of K-block s for row / column ql of both operands.
__device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)v[i];
    return r;
}

constexpr int C1 = 64;          // width of the first layer (all reference configurations)
constexpr int LD1 = C1 + 1;     // padded LDS row: column reads by 32 lanes hit 32 banks
constexpr int MAXPAIR = 3;      // output tiles per wave

__device__ __forceinline__ float lrelu(float u, float slope) { return u > 0.f ? u : u * slope; }

struct Tile {
    int TP, R, Rpad;
};

// ------------------------------------------------------------------------------------------------ forward
template <int C2, bool BF>
__global__ __launch_bounds__(256) void ec2_fwd_kernel(const float *__restrict__ pq, const int32_t *__restrict__ idx,
                                                       const float *__restrict__ w2, const float *__restrict__ gamma1,
                                                       const float *__restrict__ beta1, const float *__restrict__ mean1,
                                                       const float *__restrict__ invstd1,
                                                       const float *__restrict__ gamma2, int N, int k, int TP, int Rpad,
                                                       int training, float slope, float *__restrict__ ysel,
                                                       uint8_t *__restrict__ arg, float *__restrict__ ssum,
                                                       float *__restrict__ partials) {
    constexpr int LD2 = C2 + 1;
    constexpr int CT = C2 / 32;   // column tiles
    constexpr int CG = C2 / 64;   // channel groups of 64 lanes
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *W = sm;                           // [C2][LD1]   W2 row-major, padded
    float *Y = W + C2 * LD1;                 // [Rpad][max(LD1, LD2)]  y1, later y2
    float *red = Y + Rpad * (LD2 > LD1 ? LD2 : LD1);  // [3][4][C2]

    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform: keep it scalar
    const int ld = 2 * C1;
    const RowGather rows_pq(pq + (long)b * N * ld, (long)N * ld * 4);   // [P | Q] rows of this cloud
    const int R = TP * k;
    const int ntiles = (N + TP - 1) / TP;

    for (int t = threadIdx.x; t < C2 * C1; t += 256) W[(t / C1) * LD1 + (t % C1)] = w2[t];
    const float a1 = gamma1[lane] * invstd1[lane], b1 = beta1[lane] - mean1[lane] * a1;  // lane = channel of layer 1
    float sgn[CG], shift[CG], s1[CG], s2[CG];
    bool first[CG];
#pragma unroll
    for (int g = 0; g < CG; ++g) {
        sgn[g] = gamma2[g * 64 + lane] >= 0.f ? 1.f : -1.f;
        shift[g] = s1[g] = s2[g] = 0.f;
        first[g] = true;
    }
    float cnt = 0.f;
    // per-lane constants of the MFMA phase
    const int ql = lane & 31, half = lane >> 5;
    __syncthreads();

    for (int tile = blockIdx.y; tile < ntiles; tile += gridDim.y) {
        const int i0 = tile * TP;
        // ---- phase 1: z1 rows into LDS (lanes = layer-1 channels).  The R edge rows of a tile are contiguous in idx: every
        // wave takes 16 of them per pass and has all 16 neighbour rows (+ their centre rows) in flight at once -- the
        // gather is pure L2 latency, so the number of sequential round trips is what phase 1 costs.
        {
            const long ebase = ((long)b * N + i0) * k;
            for (int r0 = wave * 16; r0 < Rpad; r0 += 64) {
                const int rr = r0 + (lane & 15);
                int myj = 0;
                if (rr < R && i0 + rr / k < N) myj = idx[ebase + rr];
                float y[16], qv[16];
                int p = r0 / k, sl = r0 - p * k;   // point / slot of row r0 + u, advanced incrementally (uniform)
                const int p_first = p, s_first = sl;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const bool ok = r0 + u < R && i0 + p < N;
                    const int j = __builtin_amdgcn_readlane(myj, u);
                    y[u] = rows_pq.load(ok, j, ld, lane);
                    qv[u] = rows_pq.load(ok, i0 + p, ld, C1 + lane);
                    if (++sl == k) { sl = 0; ++p; }
                }
                p = p_first;
                sl = s_first;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const bool ok = r0 + u < R && i0 + p < N;
                    Y[(r0 + u) * LD1 + lane] = ok ? lrelu(__builtin_fmaf(y[u] + qv[u], a1, b1), slope) : 0.f;
                    if (++sl == k) { sl = 0; ++p; }
                }
            }
        }
        __syncthreads();

        // ---- phase 2: y2 = z1 W2^T on the matrix cores; wave takes (row tile, column tile) pairs
        const int npairs = (Rpad / 32) * CT;
        f32x16 acc[MAXPAIR];
#pragma unroll
        for (int pi = 0; pi < MAXPAIR; ++pi) {
            const int pr = wave + 4 * pi;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[pi][e] = 0.f;
            if (pr < npairs) {
                const int rt = pr / CT, ct = pr - rt * CT;
                const float *yrow = Y + (rt * 32 + ql) * LD1 + half;
                const float *wrow = W + (ct * 32 + ql) * LD1 + half;
                if (BF) {
#pragma unroll
                    for (int s = 0; s < C1 / 16; ++s) {
                        float a[8], w[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            a[i] = yrow[16 * s + 7 * half + i];     // (yrow already carries + half)
                            w[i] = wrow[16 * s + 7 * half + i];
                        }
                        acc[pi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(a), pack8(w), acc[pi], 0, 0, 0);
                    }
                } else {
#pragma unroll 8
                    for (int s = 0; s < C1 / 2; ++s)
                        acc[pi] = __builtin_amdgcn_mfma_f32_32x32x2f32(yrow[2 * s], wrow[2 * s], acc[pi], 0, 0, 0);
                }
            }
        }
        __syncthreads();  // every wave is done reading z1
        // ---- phase 3: accumulators -> Y2[row][col]
#pragma unroll
        for (int pi = 0; pi < MAXPAIR; ++pi) {
            const int pr = wave + 4 * pi;
            if (pr < npairs) {
                const int rt = pr / CT, ct = pr - rt * CT;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    Y[row * LD2 + ct * 32 + ql] = acc[pi][e];
                }
            }
        }
        __syncthreads();
        // ---- phase 4: statistics + selection (lanes = layer-2 channels)
        for (int p = wave; p < TP; p += 4) {
            const int i = i0 + p;
            if (i >= N) break;
#pragma unroll
            for (int g = 0; g < CG; ++g) {
                const int c = g * 64 + lane;
                float best = -INFINITY, tot = 0.f;
                int barg = 0;
                for (int s = 0; s < k; ++s) {
                    const float y = Y[(p * k + s) * LD2 + c];
                    tot += y;
                    const float v = sgn[g] * y;
                    if (v > best) { best = v; barg = s; }
                    if (training) {
                        if (first[g]) { shift[g] = y; first[g] = false; }
                        const float d = y - shift[g];
                        s1[g] += d;
                        s2[g] = __builtin_fmaf(d, d, s2[g]);
                    }
                }
                const long o = ((long)b * N + i) * C2 + c;
                ysel[o] = sgn[g] * best;
                arg[o] = (uint8_t)barg;
                if (ssum) ssum[o] = tot;
            }
            cnt += (float)k;
        }
        __syncthreads();  // Y is rewritten by the next tile
    }
    if (!training) return;
#pragma unroll
    for (int g = 0; g < CG; ++g) {
        float mean = 0.f, m2 = 0.f;
        if (cnt > 0.f) {
            mean = shift[g] + s1[g] / cnt;
            m2 = fmaxf(s2[g] - s1[g] * s1[g] / cnt, 0.f);
        }
        red[(0 * 4 + wave) * C2 + g * 64 + lane] = cnt;
        red[(1 * 4 + wave) * C2 + g * 64 + lane] = mean;
        red[(2 * 4 + wave) * C2 + g * 64 + lane] = m2;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C2; c += 256) {
        float n = red[(0 * 4 + 0) * C2 + c], mu = red[(1 * 4 + 0) * C2 + c], M2 = red[(2 * 4 + 0) * C2 + c];
        for (int w = 1; w < 4; ++w) {
            const float nb = red[(0 * 4 + w) * C2 + c];
            if (nb > 0.f) {
                const float tot = n + nb, delta = red[(1 * 4 + w) * C2 + c] - mu;
                mu += delta * (nb / tot);
                M2 += red[(2 * 4 + w) * C2 + c] + delta * delta * (n * nb / tot);
                n = tot;
            }
        }
        const long rec = (long)b * gridDim.y + blockIdx.y;
        float *pr = partials + rec * 3 * C2;
        pr[c] = n;
        pr[C2 + c] = mu;
        pr[2 * C2 + c] = M2;
    }
}

// ------------------------------------------------------------------------------------------------ backward
template <int C2, bool BF>
__global__ __launch_bounds__(256) void ec2_bwd_kernel(
    const float *__restrict__ pq, const int32_t *__restrict__ idx, const float *__restrict__ w2,
    const float *__restrict__ gamma1, const float *__restrict__ beta1, const float *__restrict__ mean1,
    const float *__restrict__ invstd1, const float *__restrict__ gamma2, const float *__restrict__ mean2,
    const float *__restrict__ invstd2, const float *__restrict__ dbeta2, const float *__restrict__ dgamma2,
    const float *__restrict__ h2, const uint8_t *__restrict__ arg2, int N, int k, int TP, int Rpad, int training,
    float invM, float slope, float *__restrict__ du1, float *__restrict__ dw2_part, float *__restrict__ part1) {
    constexpr int LD2 = C2 + 1;
    constexpr int CT2 = C2 / 32;          // column tiles of layer 2
    constexpr int CT1 = C1 / 32;          // column tiles of layer 1 (2)
    constexpr int NW = CT2 * CT1;         // dW2 tiles (4 or 8)
    constexpr int WPW = (NW + 3) / 4;     // dW2 tiles per wave
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *W = sm;                         // [C2][LD1]
    float *Y1 = W + C2 * LD1;              // [Rpad][LD1]  y1
    float *D = Y1 + Rpad * LD1;            // [Rpad][LD2]  dy2
    float *Hs = D + Rpad * LD2;            // [TP][C2]     h2 rows of the tile's points
    float *A1s = Hs + TP * C2;             // [C1] a1, [C1] b1
    float *red1 = A1s + 2 * C1;            // [2][C1]  sums of du1 and du1*yhat1
    uint8_t *As = reinterpret_cast<uint8_t *>(red1 + 2 * C1);   // [TP][C2] arg2

    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform: keep it scalar
    const int ql = lane & 31, half = lane >> 5;
    const int ld = 2 * C1;
    const RowGather rows_pq(pq + (long)b * N * ld, (long)N * ld * 4);   // [P | Q] rows of this cloud
    const int R = TP * k;
    const int ntiles = (N + TP - 1) / TP;

    for (int t = threadIdx.x; t < C2 * C1; t += 256) W[(t / C1) * LD1 + (t % C1)] = w2[t];
    if (threadIdx.x < C1) {
        const float a = gamma1[threadIdx.x] * invstd1[threadIdx.x];
        A1s[threadIdx.x] = a;
        A1s[C1 + threadIdx.x] = beta1[threadIdx.x] - mean1[threadIdx.x] * a;
    }
    f32x16 accw[WPW];
#pragma unroll
    for (int wi = 0; wi < WPW; ++wi)
#pragma unroll
        for (int e = 0; e < 16; ++e) accw[wi][e] = 0.f;
    // dbeta1 / dgamma1 partial sums of this lane's layer-1 channel c1 = (wave % CT1) * 32 + ql: the (row tile, column tile)
    // pairs of a wave all share the column tile (4 waves, CT1 = 2), so the sums live in registers across every tile of
    // the workgroup and are folded in wave order at the end -- no LDS float atomics, the result is reproducible
    static_assert(4 % CT1 == 0, "a wave must keep its layer-1 column tile");
    float sb_acc = 0.f, sg_acc = 0.f;
    __syncthreads();

    for (int tile = blockIdx.y; tile < ntiles; tile += gridDim.y) {
        const int i0 = tile * TP;
        const int pvalid = min(TP, N - i0);   // valid points of this tile
        const int rvalid = pvalid * k;        // valid rows
        // ---- phase 1: y1 rows (lanes = layer-1 channels), 16 rows per wave and pass with all gathers in flight (see the
        // forward kernel); h2/arg2 rows of the tile's points
        {
            const long ebase = ((long)b * N + i0) * k;
            for (int r0 = wave * 16; r0 < Rpad; r0 += 64) {
                const int rr = r0 + (lane & 15);
                int myj = 0;
                if (rr < R && i0 + rr / k < N) myj = idx[ebase + rr];
                float y[16], qv[16];
                int p = r0 / k, sl = r0 - p * k;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const bool ok = r0 + u < R && i0 + p < N;
                    const int j = __builtin_amdgcn_readlane(myj, u);
                    y[u] = rows_pq.load(ok, j, ld, lane);
                    qv[u] = rows_pq.load(ok, i0 + p, ld, C1 + lane);
                    if (++sl == k) { sl = 0; ++p; }
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) Y1[(r0 + u) * LD1 + lane] = y[u] + qv[u];   // rows beyond the tile: 0 + 0
            }
        }
        for (int p = wave; p < TP; p += 4) {
            const int i = i0 + p;
            for (int c = lane; c < C2; c += 64) {
                Hs[p * C2 + c] = i < N ? h2[((long)b * N + i) * C2 + c] : 0.f;
                As[p * C2 + c] = i < N ? arg2[((long)b * N + i) * C2 + c] : (uint8_t)255;
            }
        }
        __syncthreads();

        // ---- phase 2+3: y2 = z1 W2^T (MFMA), then dy2 = a2 (h2[s=arg] - db2 - yhat2 dg2) -> D
        const int npairs = (Rpad / 32) * CT2;
        for (int pr = wave; pr < npairs; pr += 4) {
            const int rt = pr / CT2, ct = pr - rt * CT2;
            const float *yrow = Y1 + (rt * 32 + ql) * LD1 + half;
            const float *wrow = W + (ct * 32 + ql) * LD1 + half;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            const bool rowok = rt * 32 + ql < rvalid;  // padding rows must contribute z1 = 0, not f(b1)
            if (BF) {
#pragma unroll
                for (int s = 0; s < C1 / 16; ++s) {
                    float a[8], w[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int ch = 16 * s + 8 * half + i;
                        const float z = lrelu(__builtin_fmaf(yrow[ch - half], A1s[ch], A1s[C1 + ch]), slope);
                        a[i] = rowok ? z : 0.f;
                        w[i] = wrow[ch - half];
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(a), pack8(w), acc, 0, 0, 0);
                }
            } else {
#pragma unroll 8
                for (int s = 0; s < C1 / 2; ++s) {
                    const int ch = 2 * s + half;
                    float z = lrelu(__builtin_fmaf(yrow[2 * s], A1s[ch], A1s[C1 + ch]), slope);
                    z = rowok ? z : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z, wrow[2 * s], acc, 0, 0, 0);
                }
            }
            // dy2 = a2 (h2 [s = arg] - db2 - yhat2 dg2) splits into a part that is affine in y2 -- every edge, ONE fma per
            // accumulator element: A + Bc y2 with A = a2 (mu2 r2 dg2 - db2), Bc = -a2 r2 dg2 -- and the selected-edge term
            // a2 h2, which exists for one row per (point, column) and is added by the small pass after the barrier.  (Per
            // element the unsplit form cost ~17 VALU issues: two row-table reads, the slot compare, the select.)
            const int col = ct * 32 + ql;
            const float r2 = invstd2[col], a2 = gamma2[col] * r2, mu2 = mean2[col];
            const float db2 = training ? dbeta2[col] * invM : 0.f, dg2 = training ? dgamma2[col] * invM : 0.f;
            const float Bc = -(a2 * r2) * dg2, A = a2 * (mu2 * r2 * dg2 - db2);
            const bool whole = rt * 32 + 32 <= rvalid;   // wave-uniform: every row of this row tile is a real edge
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                float dy = __builtin_fmaf(Bc, acc[e], A);
                if (!whole && row >= rvalid) dy = 0.f;
                D[row * LD2 + col] = dy;
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < pvalid * C2; t += 256) {   // + a2 h2 on the selected edge of every (point, column)
            const int p = t / C2, col = t - p * C2;
            const int sl = As[t];
            if (sl < k) D[(p * k + sl) * LD2 + col] += gamma2[col] * invstd2[col] * Hs[t];
        }
        __syncthreads();

        // ---- phase 4: dz1 = dy2 W2 (MFMA) -> du1 = dz1 f'(u1); sums for dbeta1 / dgamma1; du1 rows to HBM
        const int npairs1 = (Rpad / 32) * CT1;
        for (int pr = wave; pr < npairs1; pr += 4) {
            const int rt = pr / CT1, ct = pr - rt * CT1;
            const float *drow = D + (rt * 32 + ql) * LD2 + half;
            const float *wcol = W + half * LD1 + ct * 32 + ql;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            if (BF) {
#pragma unroll
                for (int s = 0; s < C2 / 16; ++s) {
                    float a[8], w[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int kk = 16 * s + 8 * half + i;           // layer-2 channel
                        a[i] = drow[kk - half];                          // (drow already carries + half)
                        w[i] = W[kk * LD1 + ct * 32 + ql];
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(a), pack8(w), acc, 0, 0, 0);
                }
            } else {
#pragma unroll 8
                for (int s = 0; s < C2 / 2; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(drow[2 * s], wcol[2 * s * LD1], acc, 0, 0, 0);
            }
            const int c1 = ct * 32 + ql;
            const float a1 = A1s[c1], b1 = A1s[C1 + c1], mu1 = mean1[c1], r1 = invstd1[c1];
            float sb = 0.f, sg = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (row < rvalid) {
                    const float y1 = Y1[row * LD1 + c1];
                    const float u1 = __builtin_fmaf(y1, a1, b1);
                    const float du = acc[e] * (u1 > 0.f ? 1.f : slope);
                    du1[(((long)b * N + i0) * k + row) * C1 + c1] = du;
                    sb += du;
                    sg = __builtin_fmaf(du, (y1 - mu1) * r1, sg);
                }
            }
            sb_acc += sb;
            sg_acc += sg;
        }

        // ---- phase 5: dW2 += dy2^T z1 (MFMA, K = rows); accumulators live across the tiles of this workgroup
#pragma unroll
        for (int wi = 0; wi < WPW; ++wi) {
            const int tw = wave + 4 * wi;
            if (tw < NW) {
                const int ct2 = tw / CT1, ct1 = tw - ct2 * CT1;
                const int c1 = ct1 * 32 + ql;
                const float a1 = A1s[c1], b1 = A1s[C1 + c1];
                const float *dcol = D + half * LD2 + ct2 * 32 + ql;
                const float *ycol = Y1 + half * LD1 + c1;
                if (BF) {
                    for (int s = 0; s < Rpad / 16; ++s) {     // K = edge rows, 16 per MFMA
                        float a[8], zz[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int row = 16 * s + 8 * half + i;
                            a[i] = D[row * LD2 + ct2 * 32 + ql];
                            const float z = lrelu(__builtin_fmaf(Y1[row * LD1 + c1], a1, b1), slope);
                            zz[i] = row < rvalid ? z : 0.f;
                        }
                        accw[wi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(a), pack8(zz), accw[wi], 0, 0, 0);
                    }
                } else {
                    for (int s = 0; s < Rpad / 2; ++s) {
                        const int row = 2 * s + half;
                        float z = lrelu(__builtin_fmaf(ycol[2 * s * LD1], a1, b1), slope);
                        z = row < rvalid ? z : 0.f;
                        accw[wi] = __builtin_amdgcn_mfma_f32_32x32x2f32(dcol[2 * s * LD2], z, accw[wi], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();  // Y1 / D / Hs are rewritten by the next tile
    }
    // ---- epilogue: dW2 partial tile(s) and the dbeta1/dgamma1 partial sums of this workgroup
    const long rec = (long)b * gridDim.y + blockIdx.y;
#pragma unroll
    for (int wi = 0; wi < WPW; ++wi) {
        const int tw = wave + 4 * wi;
        if (tw < NW) {
            const int ct2 = tw / CT1, ct1 = tw - ct2 * CT1;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c2 = ct2 * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                dw2_part[(rec * C2 + c2) * C1 + ct1 * 32 + ql] = accw[wi][e];
            }
        }
    }
    __syncthreads();   // the tiles are done: D is free
    {
        float *redw = D;   // [4 waves][2][C1]
        for (int t = threadIdx.x; t < 4 * 2 * C1; t += 256) redw[t] = 0.f;
        __syncthreads();
        const float sbw = sb_acc + __shfl_xor(sb_acc, 32), sgw = sg_acc + __shfl_xor(sg_acc, 32);   // the two k-halves
        if (half == 0) {
            const int c1 = (wave % CT1) * 32 + ql;
            redw[(wave * 2) * C1 + c1] = sbw;
            redw[(wave * 2 + 1) * C1 + c1] = sgw;
        }
        __syncthreads();
        if (threadIdx.x < 2 * C1) {
            const int which = threadIdx.x / C1, c = threadIdx.x % C1;
            part1[rec * 2 * C1 + threadIdx.x] = redw[(0 * 2 + which) * C1 + c] + redw[(1 * 2 + which) * C1 + c] +
                                                redw[(2 * 2 + which) * C1 + c] + redw[(3 * 2 + which) * C1 + c];
        }
    }
}

// one wave per destination point j: dP_j from the in-edges' du1 rows (reverse graph), dQ_j from its own k rows.  Round 4
// (edgeconv.hip, ec1_bwd_gather_kernel): a lane owns FOUR channels, group g = lane / 16 takes the rows t = g (mod 4) -- a wave
// instruction reads four whole 256-byte rows, the sums run on packed fp32 adds, the groups' partial sums meet through LDS in a
// fixed order ((g0 + g1) + (g2 + g3)): reproducible bit for bit from run to run.
__global__ __launch_bounds__(256) void ec2_bwd_gather_kernel(
    const float *__restrict__ pq, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ du1, const float *__restrict__ ssum1, const float *__restrict__ gamma1,
    const float *__restrict__ mean1, const float *__restrict__ invstd1, const float *__restrict__ dbeta1,
    const float *__restrict__ dgamma1, int N, int k, int training, float invM, float *__restrict__ grad_pq) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ float comb[4][3][4][C1];          // [wave][ad | aq | own][group][channel]
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform: keep it scalar
    const int j = blockIdx.y * 4 + wave;
    if (j >= N) return;                          // (wave-uniform; no workgroup barrier below)
    const int g = lane >> 4, c0 = 4 * (lane & 15);
    const int ld = 2 * C1;
    const float *P = pq + (long)b * N * ld;
    const float *Q = P + C1;
    const int beg = rowptr[(long)b * (N + 1) + j], end = rowptr[(long)b * (N + 1) + j + 1];
    const int32_t *cb = col + (long)b * N * k;
    const float *db_ = du1 + (long)b * N * k * C1;
    f2 ad[2] = {{0.f, 0.f}, {0.f, 0.f}}, aq[2] = {{0.f, 0.f}, {0.f, 0.f}}, own[2] = {{0.f, 0.f}, {0.f, 0.f}};
    constexpr int RU = 5;                        // rows in flight per group
    // the point's own k rows (contiguous): the first round is requested before the in-edge loop, consumed behind it
    const float *dj = db_ + ((long)j * k) * C1 + c0;
    f4 ow[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) ow[u] = *reinterpret_cast<const f4 *>(dj + (long)min(4 * u + g, k - 1) * C1);
    for (int t0 = beg; t0 < end; t0 += 64) {
        const int mye = (t0 + lane < end) ? cb[t0 + lane] : 0;
        const int cnt = min(64, end - t0);
        // the du1 rows come from HBM (84 MB, written by the kernel before): the number of independent loads in flight is what
        // this loop costs
        for (int t = 0; t < cnt; t += 4 * RU) {
            f4 d[RU], q[RU];
            float m[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int tt = t + 4 * u + g;
                m[u] = tt < cnt ? 1.f : 0.f;
                const int e = __builtin_amdgcn_ds_bpermute(4 * min(tt, cnt - 1), mye);
                d[u] = *reinterpret_cast<const f4 *>(db_ + ((long)(e >> 6) * k + (e & 63)) * C1 + c0);
                q[u] = training ? *reinterpret_cast<const f4 *>(Q + (long)(e >> 6) * ld + c0) : f4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const f2 mm = {m[u], m[u]};
                ad[0] += f2{d[u][0], d[u][1]} * mm;
                ad[1] += f2{d[u][2], d[u][3]} * mm;
                if (training) {
                    aq[0] += f2{q[u][0], q[u][1]} * mm;
                    aq[1] += f2{q[u][2], q[u][3]} * mm;
                }
            }
        }
    }
    for (int s0 = 0; s0 < k; s0 += 4 * RU) {
        if (s0 > 0) {
#pragma unroll
            for (int u = 0; u < RU; ++u) ow[u] = *reinterpret_cast<const f4 *>(dj + (long)min(s0 + 4 * u + g, k - 1) * C1);
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const float mk = (s0 + 4 * u + g < k) ? 1.f : 0.f;
            own[0] += f2{ow[u][0], ow[u][1]} * f2{mk, mk};
            own[1] += f2{ow[u][2], ow[u][3]} * f2{mk, mk};
        }
    }
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
        comb[wave][0][g][c0 + e4] = ad[e4 >> 1][e4 & 1];
        comb[wave][1][g][c0 + e4] = aq[e4 >> 1][e4 & 1];
        comb[wave][2][g][c0 + e4] = own[e4 >> 1][e4 & 1];
    }
    __builtin_amdgcn_wave_barrier();            // (a wave's own LDS operations complete in order)
    const int c = lane;
    auto fold = [&](int w) { return (comb[wave][w][0][c] + comb[wave][w][1][c]) + (comb[wave][w][2][c] + comb[wave][w][3][c]); };
    const float ads = fold(0), aqs = fold(1), owns = fold(2);
    const float r = invstd1[c], coef = r * gamma1[c], mu = mean1[c];
    float dp = ads, dq = owns;
    if (training) {
        const float db = dbeta1[c] * invM, dg = dgamma1[c] * invM * r;
        const float deg = (float)(end - beg);
        dp -= deg * db + dg * (deg * (P[(long)j * ld + c] - mu) + aqs);
        dq -= (float)k * db + dg * (ssum1[((long)b * N + j) * C1 + c] - (float)k * mu);
    }
    float *gp = grad_pq + ((long)b * N + j) * ld;
    gp[c] = coef * dp;
    gp[C1 + c] = coef * dq;
}


// =====================================================================================================================
// Split-bf16 kernels for C2 = 64 (every two-layer EdgeConv of DGCNN-seg): the per-edge products run on
// v_mfma_f32_32x32x16_bf16 at fp32 grade -- each fp32 operand is three bf16 pieces, a product six MFMAs (see pointwise.hip
// for the error analysis), 2.7x the rate of v_mfma_f32_32x32x2_f32 -- or, NP = 1, with plain bf16 operands (the bf16 entry
// points).  What changed against the kernels above:
//   * a thread gathers 32 bytes (eight channels) of a [P | Q] row instead of one channel, transforms and splits them ONCE on
//     their way into LDS; the MFMA phases read ready operand fragments (the old bf16 mode converted on every read);
//   * the activations live in ONE row-major bf16 image per piece, [row][64 channels] with 128-byte rows, XOR-swizzled so
//     that both the row reads (ds_read_b128: eight channels of a row, contraction over channels) and the hardware
//     transposed reads (ds_read_b64_tr_b16: four rows of a channel, contraction over edge rows, the dW2 product) are free of
//     bank conflicts;
//   * the weights are loop invariants of a wave (it keeps its column tile): their fragments stay in registers;
//   * the gathers of tile t + 1 (and the neighbour indices of tile t + 2) are in flight while tile t is computed.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {   // (a, b) -> packed bf16 pair, a in the low half, RNE
    bf16x2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}

// two fp32 values -> their three bf16 pieces, packed pairwise
__device__ __forceinline__ void split2(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
    h = pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    l = pk_bf16(sa, sb);
}

template <int NP>
__device__ __forceinline__ void split8(const float (&x)[8], u32x4 (&p)[NP]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if constexpr (NP == 3) {
            unsigned h, m, l;
            split2(x[2 * q], x[2 * q + 1], h, m, l);
            p[0][q] = h;
            p[1][q] = m;
            p[2][q] = l;
        } else {
            p[0][q] = pk_bf16(x[2 * q], x[2 * q + 1]);
        }
    }
}

// byte offset of the 16-byte chunk c (eight channels) of row `row` in a [rows][64 x bf16] dual-use image.  Bank row = 256
// bytes = two image rows.  Row reads: the 16-lane groups of ds_read_b128 ({0-3,12-15,20-27}, ...) take one chunk index of 16
// rows -- 8 even, 8 odd; the XOR value 4 ((row >> 1) & 1) + ((row >> 2) & 3) is distinct over the 8 rows of either parity.
// Transposed reads: a 32-lane half takes 4 consecutive rows x 4 consecutive chunks; rows r, r + 1 fill the two halves of a
// bank row, rows r + 2, r + 3 the other chunk quad (XOR bit 2).
__device__ __forceinline__ int img_off(int row, int c) {
    return 128 * row + 16 * (c ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)));
}

template <int NP>
__device__ __forceinline__ f32x16 mfma_split(const u32x4 (&a)[NP], const u32x4 (&b)[NP], f32x16 c) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a[0]), bh = __builtin_bit_cast(bf16x8, b[0]);
    if constexpr (NP == 3) {
        const bf16x8 am = __builtin_bit_cast(bf16x8, a[1]), al = __builtin_bit_cast(bf16x8, a[2]);
        const bf16x8 bm = __builtin_bit_cast(bf16x8, b[1]), bl = __builtin_bit_cast(bf16x8, b[2]);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
}

// operand fragment for a contraction over CHANNELS: lane (ql, half) takes channels 16 ks + 8 half .. + 7 of row rt * 32 + ql
template <int NP>
__device__ __forceinline__ void row_frag(const unsigned char *img, int piece_bytes, int rt, int ks, int ql, int half,
                                         u32x4 (&f)[NP]) {
    const int off = img_off(rt * 32 + ql, 2 * ks + half);
#pragma unroll
    for (int q = 0; q < NP; ++q) f[q] = *reinterpret_cast<const u32x4 *>(img + q * piece_bytes + off);
}

// operand fragment for a contraction over ROWS: lane (ql, half) takes rows 16 ks + 8 half .. + 7 of channel ctile * 32 + ql.
// ds_read_b64_tr_b16: lane 4 q + p of a 16-lane group supplies the address of channels 4 p .. 4 p + 3 of the block's row q and
// receives channel (lane % 16) of the four rows; EXEC must be all ones (callers are wave-uniform)
template <int NP>
__device__ __forceinline__ void tr_frag(const unsigned char *img, int piece_bytes, int ctile, int ks, int lane, u32x4 (&f)[NP]) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, h = lane >> 5;
    const int row = 16 * ks + 8 * h + q, c = 4 * ctile + 2 * (g & 1) + (p >> 1);
    const int o0 = img_off(row, c) + 8 * (p & 1), o1 = img_off(row + 4, c) + 8 * (p & 1);
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) {
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4 *)(img + pc * piece_bytes + o0));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4 *)(img + pc * piece_bytes + o1));
        const u32x2 a = __builtin_bit_cast(u32x2, v0), b = __builtin_bit_cast(u32x2, v1);
        f[pc] = u32x4{a[0], a[1], b[0], b[1]};
    }
}

// 16-byte gathers with a per-lane byte offset into the [P | Q] rows of a cloud; !ok -> an offset outside the resource (0)
struct ChunkGather {
    __amdgpu_buffer_rsrc_t rs;
    unsigned oob;
    __device__ __forceinline__ ChunkGather(const float *base, long bytes) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(base);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        const int n = __builtin_amdgcn_readfirstlane((int)bytes);
        rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo), 0, n, 0x00020000);
        oob = (unsigned)n;
    }
    __device__ __forceinline__ float4 load(bool ok, unsigned byte_off) const {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? byte_off : oob, 0, 0));
    }
};

// the gather state of one tile: RT items per thread, item `it` = (row irow + 32 it, chunk) of the tile
template <int RT>
struct TileRows {
    float4 p[RT][2];
};

template <int NP, int RT>
__global__ __launch_bounds__(256) void ec2s_fwd_kernel(const float *__restrict__ pq, const int32_t *__restrict__ idx,
                                                        const float *__restrict__ w2, const float *__restrict__ gamma1,
                                                        const float *__restrict__ beta1, const float *__restrict__ mean1,
                                                        const float *__restrict__ invstd1,
                                                        const float *__restrict__ gamma2, int N, int k, int TP,
                                                        int training, float slope, float *__restrict__ ysel,
                                                        uint8_t *__restrict__ arg, float *__restrict__ ssum,
                                                        float *__restrict__ partials) {
    constexpr int C2 = 64, LD2 = C2 + 1, Rpad = 32 * RT, PB = Rpad * 128;   // PB: bytes of one piece image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Z = smem;                                            // [NP][Rpad][128]  z1 pieces
    float *Y = reinterpret_cast<float *>(Z + NP * PB);                  // [Rpad][LD2]      y2
    float *red = Y + Rpad * LD2;                                        // [3][4][C2]
    float *A1s = red + 3 * 4 * C2;                                      // [2][C1]          a1, b1 of BatchNorm 1
    float *QS = A1s + 2 * C1;                                           // [2][TP][C1]      Q rows of the tile's points (double buffer)

    const int b = blockIdx.x, G = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, half = lane >> 5;
    const int ld = 2 * C1;
    const ChunkGather rows_pq(pq + (long)b * N * ld, (long)N * ld * 4);
    const int R = TP * k;
    const int ntiles = (N + TP - 1) / TP;
    const int chunk = tid & 7, irow = tid >> 3;

    if (tid < C1) {
        const float a = gamma1[tid] * invstd1[tid];
        A1s[tid] = a;
        A1s[C1 + tid] = beta1[tid] - mean1[tid] * a;
    }
    // W2 fragments of this wave's column tile (B operand: column c2, eight consecutive c1)
    const int ct = wave & 1;
    u32x4 wf[C1 / 16][NP];
#pragma unroll
    for (int ks = 0; ks < C1 / 16; ++ks) {
        float x[8];
        const float4 *src = reinterpret_cast<const float4 *>(w2 + (ct * 32 + ql) * C1 + 16 * ks + 8 * half);
        const float4 x0 = src[0], x1 = src[1];
        x[0] = x0.x; x[1] = x0.y; x[2] = x0.z; x[3] = x0.w; x[4] = x1.x; x[5] = x1.y; x[6] = x1.z; x[7] = x1.w;
        split8<NP>(x, wf[ks]);
    }
    int prow[RT];      // point of an item's row within the tile (the same for every tile)
#pragma unroll
    for (int it = 0; it < RT; ++it) prow[it] = min((irow + 32 * it) / k, TP - 1);   // rows behind the tile: any valid row

    const float sgn = gamma2[lane] >= 0.f ? 1.f : -1.f;
    float shift = 0.f, s1 = 0.f, s2 = 0.f, cnt = 0.f;
    bool first = true;

    auto load_idx = [&](int tile, int (&j)[RT]) {
        const int i0 = tile * TP;
        const long ebase = ((long)b * N + i0) * k;
#pragma unroll
        for (int it = 0; it < RT; ++it) {
            const int row = irow + 32 * it;
            const bool ok = tile < ntiles && row < R && i0 + prow[it] < N;
            j[it] = ok ? idx[ebase + row] : -1;
        }
    };
    auto load_rows = [&](int tile, const int (&j)[RT], TileRows<RT> &t) {
    #pragma unroll
        for (int it = 0; it < RT; ++it) {
            const bool ok = j[it] >= 0;
            const unsigned po = (unsigned)j[it] * (unsigned)(ld * 4) + chunk * 32u;
            t.p[it][0] = rows_pq.load(ok, po);
            t.p[it][1] = rows_pq.load(ok, po + 16u);
        }
    };
    // Q rows of a tile's points: one 16-byte piece per thread (TP <= 16), prefetched two tiles ahead and parked in QS
    auto load_q = [&](int tile) {
        const int p = tid >> 4, i = tile * TP + p;
        const bool ok = tile < ntiles && p < TP && i < N;
        return rows_pq.load(ok, (unsigned)i * (unsigned)(ld * 4) + C1 * 4u + (tid & 15) * 16u);
    };

    int jc[RT], jn[RT];
    TileRows<RT> cur;
    load_idx(blockIdx.y, jc);
    load_rows(blockIdx.y, jc, cur);
    load_idx(blockIdx.y + G, jn);
    if (tid < TP * 16) *reinterpret_cast<float4 *>(QS + tid * 4) = load_q(blockIdx.y);
    float4 qn = load_q(blockIdx.y + G);
    __syncthreads();   // the BatchNorm table and the first tile's Q rows

    int par = 0;       // parity of the tile within this workgroup: its Q rows are in QS[par]
    for (int tile = blockIdx.y; tile < ntiles; tile += G, par ^= 1) {
        const int i0 = tile * TP;
        // ---- phase 1: z1 = LeakyReLU(BN1(P_j + Q_i)) -> pieces -> image
        const float4 *ab = reinterpret_cast<const float4 *>(A1s + chunk * 8);
        const float4 a1l = ab[0], a1h = ab[1], b1l = ab[C1 / 4], b1h = ab[C1 / 4 + 1];
        const float a1[8] = {a1l.x, a1l.y, a1l.z, a1l.w, a1h.x, a1h.y, a1h.z, a1h.w};
        const float b1[8] = {b1l.x, b1l.y, b1l.z, b1l.w, b1h.x, b1h.y, b1h.z, b1h.w};
#pragma unroll
        for (int it = 0; it < RT; ++it) {
            const bool ok = jc[it] >= 0;
            const float4 *qs = reinterpret_cast<const float4 *>(QS + (par * TP + prow[it]) * C1 + chunk * 8);
            const float4 q0 = qs[0], q1 = qs[1];
            const float y[8] = {cur.p[it][0].x + q0.x, cur.p[it][0].y + q0.y, cur.p[it][0].z + q0.z, cur.p[it][0].w + q0.w,
                                cur.p[it][1].x + q1.x, cur.p[it][1].y + q1.y, cur.p[it][1].z + q1.z, cur.p[it][1].w + q1.w};
            float z[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = ok ? lrelu(__builtin_fmaf(y[j], a1[j], b1[j]), slope) : 0.f;
            u32x4 pc[NP];
            split8<NP>(z, pc);
            const int off = img_off(irow + 32 * it, chunk);
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x4 *>(Z + q * PB + off) = pc[q];
        }
        __syncthreads();   // the image is complete; every wave is past the previous tile's selection pass (Y is free)
#pragma unroll
        for (int it = 0; it < RT; ++it) jc[it] = jn[it];
        load_rows(tile + G, jc, cur);      // behind the last tile: every index is -1, the loads touch nothing
        load_idx(tile + 2 * G, jn);
        if (tid < TP * 16) *reinterpret_cast<float4 *>(QS + ((par ^ 1) * TP * C1) + tid * 4) = qn;   // the next tile's Q rows
        qn = load_q(tile + 2 * G);

        // ---- phase 2 + 3: y2 = z1 W2^T; wave takes row tiles (wave >> 1) + 2 pi of its column tile; accumulators -> Y
#pragma unroll
        for (int pi = 0; pi < RT / 2; ++pi) {
            const int rt = (wave >> 1) + 2 * pi;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < C1 / 16; ++ks) {
                u32x4 af[NP];
                row_frag<NP>(Z, PB, rt, ks, ql, half, af);
                acc = mfma_split<NP>(af, wf[ks], acc);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                Y[row * LD2 + ct * 32 + ql] = acc[e];
            }
        }
        __syncthreads();
        // ---- phase 4: statistics + selection (lanes = layer-2 channels)
        for (int p = wave; p < TP; p += 4) {
            const int i = i0 + p;
            if (i >= N) break;
            float best = -INFINITY, tot = 0.f;
            int barg = 0;
            const float *yp = Y + p * k * LD2 + lane;
            if (first) {                     // wave-uniform, once per wave: the shift of the BatchNorm-2 sums
                shift = yp[0];
                first = false;
            }
            float t1 = 0.f, t2 = 0.f;
            // branch-free: selects instead of `if (v > best)` (the compiler turned that into a branch per row: with the row
            // bound and the first-value test the loop ran ~290 cycles per row -- half of the kernel at k = 40)
            for (int s0 = 0; s0 < k; s0 += 8) {     // eight rows per round: the LDS reads are in flight together
                float yv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) yv[u] = yp[min(s0 + u, k - 1) * LD2];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool in = s0 + u < k;       // wave-uniform
                    const float y = yv[u];
                    tot += in ? y : 0.f;
                    const float v = in ? sgn * y : -INFINITY;
                    const bool gt = v > best;
                    best = gt ? v : best;
                    barg = gt ? s0 + u : barg;
                    const float d = in ? y - shift : 0.f;
                    t1 += d;
                    t2 = __builtin_fmaf(d, d, t2);
                }
            }
            if (training) {
                s1 += t1;
                s2 += t2;
            }
            const long o = ((long)b * N + i) * C2 + lane;
            ysel[o] = sgn * best;
            arg[o] = (uint8_t)barg;
            if (ssum) ssum[o] = tot;
            cnt += (float)k;
        }
    }
    if (!training) return;
    {
        float mean = 0.f, m2 = 0.f;
        if (cnt > 0.f) {
            mean = shift + s1 / cnt;
            m2 = fmaxf(s2 - s1 * s1 / cnt, 0.f);
        }
        __syncthreads();
        red[(0 * 4 + wave) * C2 + lane] = cnt;
        red[(1 * 4 + wave) * C2 + lane] = mean;
        red[(2 * 4 + wave) * C2 + lane] = m2;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C2; c += 256) {
        float n = red[(0 * 4 + 0) * C2 + c], mu = red[(1 * 4 + 0) * C2 + c], M2 = red[(2 * 4 + 0) * C2 + c];
        for (int w = 1; w < 4; ++w) {
            const float nb = red[(0 * 4 + w) * C2 + c];
            if (nb > 0.f) {
                const float tot = n + nb, delta = red[(1 * 4 + w) * C2 + c] - mu;
                mu += delta * (nb / tot);
                M2 += red[(2 * 4 + w) * C2 + c] + delta * delta * (n * nb / tot);
                n = tot;
            }
        }
        const long rec = (long)b * gridDim.y + blockIdx.y;
        float *pr = partials + rec * 3 * C2;
        pr[c] = n;
        pr[C2 + c] = mu;
        pr[2 * C2 + c] = M2;
    }
}

// ------------------------------------------------------------------------------------------------ backward (split-bf16)
// Per tile: y2^T = W2 z1^T (A = W2 fragments in registers, B = row reads of the z1 image) -> dy2 in registers with lanes = edge
// rows -> dy2 image (row-major, 8-byte stores);  dz1 = dy2 W2 (A = row reads of the dy2 image, B = W2 fragments in registers)
// -> du1 rows to HBM + the dbeta1/dgamma1 sums;  dW2 += dy2^T z1 (both operands by transposed reads of the two images).
template <int NP, int RT>
__global__ __launch_bounds__(256, (NP == 3 && RT == 4) ? 1 : 2) void ec2s_bwd_kernel(
    const float *__restrict__ pq, const int32_t *__restrict__ idx, const float *__restrict__ w2,
    const float *__restrict__ gamma1, const float *__restrict__ beta1, const float *__restrict__ mean1,
    const float *__restrict__ invstd1, const float *__restrict__ gamma2, const float *__restrict__ mean2,
    const float *__restrict__ invstd2, const float *__restrict__ dbeta2, const float *__restrict__ dgamma2,
    const float *__restrict__ h2, const uint8_t *__restrict__ arg2, int N, int k, int TP, int training,
    float invM, float slope, float *__restrict__ du1, float *__restrict__ dw2_part, float *__restrict__ part1) {
    constexpr int C2 = 64, Rpad = 32 * RT, PB = Rpad * 128, LDY = C1 + 4;   // LDY: 16-byte aligned fp32 rows, 4 banks apart
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Z = smem;                                       // [NP][Rpad][128]  z1 pieces
    unsigned char *D = Z + NP * PB;                                // [NP][Rpad][128]  dy2 pieces
    float *Y1 = reinterpret_cast<float *>(D + NP * PB);            // [Rpad][LDY]      y1 (fp32, for f'(u1) and yhat1)
    float *BcA = Y1 + Rpad * LDY;                                  // [C2][2]          dy2 = Bc y2 + A (+ selected-edge term)
    float *A2s = BcA + 2 * C2;                                     // [C2]             gamma2 * invstd2
    float *A1s = A2s + C2;                                         // [2][C1]          a1, b1
    float *HT = A1s + 2 * C1;                                      // [TP][C2][2]      (a2 h2, slot of the selected edge)
    float *QS = HT + TP * C2 * 2;                                  // [2][TP][C1]      Q rows of the tile's points (double buffer)

    const int b = blockIdx.x, G = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, half = lane >> 5;
    const int ld = 2 * C1;
    const ChunkGather rows_pq(pq + (long)b * N * ld, (long)N * ld * 4);
    const int R = TP * k;
    const int ntiles = (N + TP - 1) / TP;
    const int chunk = tid & 7, irow = tid >> 3;

    if (tid < C2) {
        const float r2 = invstd2[tid], a2 = gamma2[tid] * r2, mu2 = mean2[tid];
        const float db2 = training ? dbeta2[tid] * invM : 0.f, dg2 = training ? dgamma2[tid] * invM : 0.f;
        BcA[2 * tid] = -(a2 * r2) * dg2;
        BcA[2 * tid + 1] = a2 * (mu2 * r2 * dg2 - db2);
        A2s[tid] = a2;
        const float a = gamma1[tid] * invstd1[tid];
        A1s[tid] = a;
        A1s[C1 + tid] = beta1[tid] - mean1[tid] * a;
    }
    // W2 fragments: wfa = rows c2 of the wave's layer-2 tile, eight consecutive c1 (A operand of y2^T = W2 z1^T);
    //               wfb = columns c1 of the wave's layer-1 tile, eight consecutive c2 (B operand of dz1 = dy2 W2)
    const int ctw = wave & 1;
    u32x4 wfa[C1 / 16][NP], wfb[C2 / 16][NP];
#pragma unroll
    for (int ks = 0; ks < C1 / 16; ++ks) {
        float x[8];
        const float4 *src = reinterpret_cast<const float4 *>(w2 + (ctw * 32 + ql) * C1 + 16 * ks + 8 * half);
        const float4 x0 = src[0], x1 = src[1];
        x[0] = x0.x; x[1] = x0.y; x[2] = x0.z; x[3] = x0.w; x[4] = x1.x; x[5] = x1.y; x[6] = x1.z; x[7] = x1.w;
        split8<NP>(x, wfa[ks]);
    }
#pragma unroll
    for (int ks = 0; ks < C2 / 16; ++ks) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = w2[(16 * ks + 8 * half + j) * C1 + ctw * 32 + ql];
        split8<NP>(x, wfb[ks]);
    }
    int prow[RT];      // point of an item's row within the tile (gather mapping), the same for every tile
#pragma unroll
    for (int it = 0; it < RT; ++it) prow[it] = min((irow + 32 * it) / k, TP - 1);   // rows behind the tile: any valid row
    int pmf[RT / 2], smf[RT / 2];   // point and slot of row rt * 32 + ql (lanes = rows in the dy2 epilogue)
#pragma unroll
    for (int pi = 0; pi < RT / 2; ++pi) {
        const int row = ((wave >> 1) + 2 * pi) * 32 + ql;
        const int p = row / k;
        smf[pi] = row - p * k;
        pmf[pi] = min(p, TP - 1);
    }

    f32x16 accw;
#pragma unroll
    for (int e = 0; e < 16; ++e) accw[e] = 0.f;
    float sb_acc = 0.f, sg_acc = 0.f;

    auto load_idx = [&](int tile, int (&j)[RT]) {
        const int i0 = tile * TP;
        const long ebase = ((long)b * N + i0) * k;
#pragma unroll
        for (int it = 0; it < RT; ++it) {
            const int row = irow + 32 * it;
            const bool ok = tile < ntiles && row < R && i0 + prow[it] < N;
            j[it] = ok ? idx[ebase + row] : -1;
        }
    };
    auto load_rows = [&](int tile, const int (&j)[RT], TileRows<RT> &t) {
    #pragma unroll
        for (int it = 0; it < RT; ++it) {
            const bool ok = j[it] >= 0;
            const unsigned po = (unsigned)j[it] * (unsigned)(ld * 4) + chunk * 32u;
            t.p[it][0] = rows_pq.load(ok, po);
            t.p[it][1] = rows_pq.load(ok, po + 16u);
        }
    };
    // Q rows of a tile's points: one 16-byte piece per thread (TP <= 16), prefetched two tiles ahead and parked in QS
    auto load_q = [&](int tile) {
        const int p = tid >> 4, i = tile * TP + p;
        const bool ok = tile < ntiles && p < TP && i < N;
        return rows_pq.load(ok, (unsigned)i * (unsigned)(ld * 4) + C1 * 4u + (tid & 15) * 16u);
    };
    // h2 / arg2 of the tile's points: one (point, channel) per thread while TP <= 4 (prefetched), a loop otherwise
    const bool ht_reg = TP <= 4;
    float hn = 0.f;
    int an = 255;
    auto load_ht = [&](int tile) {
        const int p = tid >> 6, i = tile * TP + p;
        const bool ok = tile < ntiles && p < TP && i < N;
        hn = ok ? h2[((long)b * N + i) * C2 + lane] : 0.f;
        an = ok ? (int)arg2[((long)b * N + i) * C2 + lane] : 255;
    };

    int jc[RT], jn[RT];
    TileRows<RT> cur;
    load_idx(blockIdx.y, jc);
    load_rows(blockIdx.y, jc, cur);
    load_idx(blockIdx.y + G, jn);
    if (ht_reg) load_ht(blockIdx.y);
    if (tid < TP * 16) *reinterpret_cast<float4 *>(QS + tid * 4) = load_q(blockIdx.y);
    float4 qn = load_q(blockIdx.y + G);
    __syncthreads();   // the constant tables and the first tile's Q rows

    int par = 0;       // parity of the tile within this workgroup: its Q rows are in QS[par]
    for (int tile = blockIdx.y; tile < ntiles; tile += G, par ^= 1) {
        const int i0 = tile * TP;
        const int pvalid = min(TP, N - i0);   // valid points of this tile
        const int rvalid = pvalid * k;        // valid rows
        // ---- phase 1: y1 -> Y1 (fp32), z1 = LeakyReLU(BN1(y1)) -> pieces -> Z; rows beyond the tile: zeros
        const float4 *ab = reinterpret_cast<const float4 *>(A1s + chunk * 8);
        const float4 a1l = ab[0], a1h = ab[1], b1l = ab[C1 / 4], b1h = ab[C1 / 4 + 1];
        const float a1[8] = {a1l.x, a1l.y, a1l.z, a1l.w, a1h.x, a1h.y, a1h.z, a1h.w};
        const float b1[8] = {b1l.x, b1l.y, b1l.z, b1l.w, b1h.x, b1h.y, b1h.z, b1h.w};
#pragma unroll
        for (int it = 0; it < RT; ++it) {
            const bool ok = jc[it] >= 0;
            const float4 *qs = reinterpret_cast<const float4 *>(QS + (par * TP + prow[it]) * C1 + chunk * 8);
            const float4 q0 = qs[0], q1 = qs[1];
            float y[8] = {cur.p[it][0].x + q0.x, cur.p[it][0].y + q0.y, cur.p[it][0].z + q0.z, cur.p[it][0].w + q0.w,
                          cur.p[it][1].x + q1.x, cur.p[it][1].y + q1.y, cur.p[it][1].z + q1.z, cur.p[it][1].w + q1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = ok ? y[j] : 0.f;      // rows beyond the tile: zeros
            float z[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = ok ? lrelu(__builtin_fmaf(y[j], a1[j], b1[j]), slope) : 0.f;
            u32x4 pc[NP];
            split8<NP>(z, pc);
            const int row = irow + 32 * it;
            const int off = img_off(row, chunk);
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x4 *>(Z + q * PB + off) = pc[q];
            float4 *yd = reinterpret_cast<float4 *>(Y1 + row * LDY + chunk * 8);
            yd[0] = float4{y[0], y[1], y[2], y[3]};
            yd[1] = float4{y[4], y[5], y[6], y[7]};
        }
        if (ht_reg) {
            if (tid < TP * C2) {
                HT[2 * tid] = A2s[lane] * hn;
                HT[2 * tid + 1] = __int_as_float(an);
            }
        } else {
            for (int t = tid; t < TP * C2; t += 256) {
                const int p = t >> 6, c = t & 63, i = i0 + p;
                HT[2 * t] = i < N ? A2s[c] * h2[((long)b * N + i) * C2 + c] : 0.f;
                HT[2 * t + 1] = __int_as_float(i < N ? (int)arg2[((long)b * N + i) * C2 + c] : 255);
            }
        }
        __syncthreads();   // (1) Z, Y1, HT complete
#pragma unroll
        for (int it = 0; it < RT; ++it) jc[it] = jn[it];
        load_rows(tile + G, jc, cur);      // behind the last tile: every index is -1, the loads touch nothing
        load_idx(tile + 2 * G, jn);
        if (ht_reg) load_ht(tile + G);
        if (tid < TP * 16) *reinterpret_cast<float4 *>(QS + ((par ^ 1) * TP * C1) + tid * 4) = qn;   // the next tile's Q rows
        qn = load_q(tile + 2 * G);

        // ---- phase 2 + 3: y2^T = W2 z1^T; lanes = edge rows; dy2 = Bc y2 + A (+ a2 h2 on the selected edge) -> D
#pragma unroll
        for (int pi = 0; pi < RT / 2; ++pi) {
            const int rt = (wave >> 1) + 2 * pi;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < C1 / 16; ++ks) {
                u32x4 zf[NP];
                row_frag<NP>(Z, PB, rt, ks, ql, half, zf);
                acc = mfma_split<NP>(wfa[ks], zf, acc);
            }
            const int row = rt * 32 + ql;
            const bool valid = row < rvalid;
            const float *ht = HT + (pmf[pi] * C2 + ctw * 32 + 4 * half) * 2;
            const float *bc = BcA + (ctw * 32 + 4 * half) * 2;
            const int sl = smf[pi];
#pragma unroll
            for (int g = 0; g < 4; ++g) {     // channels c2 = ctw * 32 + 8 g + 4 half + 0..3 = accumulator elements 4 g + 0..3
                const float4 c0 = *reinterpret_cast<const float4 *>(bc + 16 * g), c1 = *reinterpret_cast<const float4 *>(bc + 16 * g + 4);
                const float4 t0 = *reinterpret_cast<const float4 *>(ht + 16 * g), t1 = *reinterpret_cast<const float4 *>(ht + 16 * g + 4);
                float dy[4];
                dy[0] = __builtin_fmaf(c0.x, acc[4 * g + 0], c0.y) + (__float_as_int(t0.y) == sl ? t0.x : 0.f);
                dy[1] = __builtin_fmaf(c0.z, acc[4 * g + 1], c0.w) + (__float_as_int(t0.w) == sl ? t0.z : 0.f);
                dy[2] = __builtin_fmaf(c1.x, acc[4 * g + 2], c1.y) + (__float_as_int(t1.y) == sl ? t1.x : 0.f);
                dy[3] = __builtin_fmaf(c1.z, acc[4 * g + 3], c1.w) + (__float_as_int(t1.w) == sl ? t1.z : 0.f);
#pragma unroll
                for (int j = 0; j < 4; ++j) dy[j] = valid ? dy[j] : 0.f;
                unsigned hh[2], mm[2] = {0u, 0u}, ll[2] = {0u, 0u};
                if constexpr (NP == 3) {
                    split2(dy[0], dy[1], hh[0], mm[0], ll[0]);
                    split2(dy[2], dy[3], hh[1], mm[1], ll[1]);
                } else {
                    hh[0] = pk_bf16(dy[0], dy[1]);
                    hh[1] = pk_bf16(dy[2], dy[3]);
                }
                const int off = img_off(row, 4 * ctw + g) + 8 * half;
                *reinterpret_cast<u32x2 *>(D + off) = u32x2{hh[0], hh[1]};
                if constexpr (NP == 3) {
                    *reinterpret_cast<u32x2 *>(D + PB + off) = u32x2{mm[0], mm[1]};
                    *reinterpret_cast<u32x2 *>(D + 2 * PB + off) = u32x2{ll[0], ll[1]};
                }
            }
        }
        __syncthreads();   // (2) D complete

        // ---- phase 4: dz1 = dy2 W2 -> du1 = dz1 f'(u1); sums for dbeta1 / dgamma1; du1 rows to HBM
#pragma unroll
        for (int pi = 0; pi < RT / 2; ++pi) {
            const int rt = (wave >> 1) + 2 * pi;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < C2 / 16; ++ks) {
                u32x4 df[NP];
                row_frag<NP>(D, PB, rt, ks, ql, half, df);
                acc = mfma_split<NP>(df, wfb[ks], acc);
            }
            const int c1 = ctw * 32 + ql;
            const float a1c = A1s[c1], b1c = A1s[C1 + c1], mu1 = mean1[c1], r1 = invstd1[c1];
            float sb = 0.f, sg = 0.f;
            // the tile's du1 rows as a buffer resource of exactly rvalid rows: rows behind it (dy2 = 0, so du1 = 0 and nothing is
            // added to the sums) fall outside and are dropped by the hardware -- sixteen unconditional stores, no branches, and a
            // store count the compiler can subtract when it waits for the prefetched gathers
            const uintptr_t da = reinterpret_cast<uintptr_t>(du1 + (((long)b * N + i0) * k) * C1);
            const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)da), dhi = __builtin_amdgcn_readfirstlane((unsigned)(da >> 32));
            const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<void *>(((uintptr_t)dhi << 32) | dlo), 0, __builtin_amdgcn_readfirstlane(rvalid * C1 * 4), 0x00020000);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                const float y1 = Y1[row * LDY + c1];
                const float u1 = __builtin_fmaf(y1, a1c, b1c);
                const float du = acc[e] * (u1 > 0.f ? 1.f : slope);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, du), drs, (unsigned)(row * C1 + c1) * 4u, 0, 0);
                sb += du;
                sg = __builtin_fmaf(du, (y1 - mu1) * r1, sg);
            }
            sb_acc += sb;
            sg_acc += sg;
        }
        // ---- phase 5: dW2 += dy2^T z1 (contraction over the tile's rows, transposed reads of both images); the wave owns
        // the (c2 tile wave >> 1, c1 tile wave & 1) block across every tile of the workgroup
#pragma unroll
        for (int ks = 0; ks < Rpad / 16; ++ks) {
            u32x4 df[NP], zf[NP];
            tr_frag<NP>(D, PB, wave >> 1, ks, lane, df);
            tr_frag<NP>(Z, PB, wave & 1, ks, lane, zf);
            accw = mfma_split<NP>(df, zf, accw);
        }
        __syncthreads();   // (3) Z / D / Y1 / HT are rewritten by the next tile
    }
    // ---- epilogue: dW2 partial block and the dbeta1/dgamma1 partial sums of this workgroup
    const long rec = (long)b * gridDim.y + blockIdx.y;
    {
        const int ct2 = wave >> 1, ct1 = wave & 1;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c2 = ct2 * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            dw2_part[(rec * C2 + c2) * C1 + ct1 * 32 + ql] = accw[e];
        }
    }
    {
        float *redw = reinterpret_cast<float *>(D);   // [4 waves][2][C1]; the tiles are done (barrier 3)
        for (int t = threadIdx.x; t < 4 * 2 * C1; t += 256) redw[t] = 0.f;
        __syncthreads();
        const float sbw = sb_acc + __shfl_xor(sb_acc, 32), sgw = sg_acc + __shfl_xor(sg_acc, 32);   // the two k-halves
        if (half == 0) {
            const int c1 = ctw * 32 + ql;
            redw[(wave * 2) * C1 + c1] = sbw;
            redw[(wave * 2 + 1) * C1 + c1] = sgw;
        }
        __syncthreads();
        if (threadIdx.x < 2 * C1) {
            const int which = threadIdx.x / C1, c = threadIdx.x % C1;
            part1[rec * 2 * C1 + threadIdx.x] = redw[(0 * 2 + which) * C1 + c] + redw[(1 * 2 + which) * C1 + c] +
                                                redw[(2 * 2 + which) * C1 + c] + redw[(3 * 2 + which) * C1 + c];
        }
    }
}

}  // namespace

// shared with edgeconv.hip
int fsg_ec_stats1_launch(const float *pq, const int32_t *idx, const float *gamma, int B, int N, int k, int Co,
                         float *ysel, uint8_t *arg, float *ssum, float *partials, hipStream_t st);
int fsg_ec_finalize_launch(const float *partials, int R, int Co, float eps, float momentum, float *mean, float *invstd,
                           float *running_mean, float *running_var, hipStream_t st);
int fsg_ec_apply_launch(const float *ysel, const float *gamma, const float *beta, const float *mean, const float *invstd,
                        int B, int N, int Co, float slope, float *out, float *out_pm, hipStream_t st);
int fsg_ec_stats1_records(int B, int N);
size_t fsg_ec_finalize_stage_floats(int Co);

static int ec2_env_int(const char *name) {
    const char *e = getenv(name);
    return e ? atoi(e) : 0;
}

// C2 = 64 runs on the split-bf16 kernels (FSG_EC2_OLD=1: the fp32-MFMA kernels above, kept for C2 = 128 and as a cross-check)
static int ec2_old_env = -1;
// tests: switch the C2 = 64 path between the split-bf16 kernels (0) and the fp32-MFMA kernels (1) inside one process
// (tests/test_gpu_parity.py::test_ec2s_is_fp32_grade holds the former to the latter's error against fp64)
extern "C" void fsg_debug_ec2_use_fp32_mfma(int on) { ec2_old_env = on ? 1 : 0; }
static bool ec2_split(int C2) {
    if (ec2_old_env < 0) ec2_old_env = ec2_env_int("FSG_EC2_OLD") > 0 ? 1 : 0;
    return C2 == 64 && !ec2_old_env;
}

// tiles of the split kernels: RT * 32 edge rows = TP points; 128 rows when that cuts the padding by more than 15 % (k = 40:
// 40 of 64 rows against 120 of 128), 64 rows otherwise; `wgs` workgroups over the B clouds
static void ec2s_tiling(int k, int B, int N, bool bwd, bool bf16, int &RT, int &TP, int &G) {
    static int rt_env = -1, fwd_wgs = -1, bwd_wgs = -1;
    if (rt_env < 0) {
        rt_env = ec2_env_int("FSG_EC2S_RT");
        fwd_wgs = ec2_env_int("FSG_EC2S_FWD_WGS");
        bwd_wgs = ec2_env_int("FSG_EC2S_BWD_WGS");
    }
    const int tp2 = 64 / k, tp4 = 128 / k;
    const float u2 = tp2 * k / 64.f, u4 = tp4 * k / 128.f;
    RT = u4 > 1.15f * u2 ? 4 : 2;
    // the forward with three pieces per operand: 128-row tiles hold one workgroup per CU (82 KB of LDS) and lose more than the
    // padding costs (k = 40, 32 x 2048: 472 vs 545 us); the backward and the one-piece kernels keep the rule above
    if (!bwd && !bf16) RT = 2;
    if (rt_env == 2 || rt_env == 4) RT = rt_env;
    TP = RT == 4 ? tp4 : tp2;
    if (TP < 1) TP = 1;
    if (TP > 16) TP = 16;      // the kernels stage a tile's Q rows with one 16-byte piece per thread
    const int ntiles = (N + TP - 1) / TP;
    const int wgs = bwd ? (bwd_wgs > 0 ? bwd_wgs : 512) : (fwd_wgs > 0 ? fwd_wgs : 768);
    G = wgs / (B > 0 ? B : 1);
    if (G < 1) G = 1;
    if (G > ntiles) G = ntiles;
}

static size_t ec2s_fwd_lds(int NP, int RT, int TP) {
    return (size_t)NP * RT * 32 * 128 + sizeof(float) * ((size_t)RT * 32 * 65 + 3 * 4 * 64 + 2 * C1 + 2 * (size_t)TP * C1);
}
static size_t ec2s_bwd_lds(int NP, int RT, int TP) {
    return (size_t)2 * NP * RT * 32 * 128 +
           sizeof(float) * ((size_t)RT * 32 * (C1 + 4) + 2 * 64 + 64 + 2 * C1 + (size_t)TP * 64 * 2 + 2 * (size_t)TP * C1);
}

static void ec2_tiling(int k, int C2, int &TP, int &Rpad, int &G, int B, int N) {
    static int rmax_env = -1;
    if (rmax_env < 0) {
        const char *e = getenv("FSG_EC2_FWD_RMAX");
        rmax_env = e ? atoi(e) : 0;
    }
    const int rmax = rmax_env > 0 ? rmax_env : 64;   // measured: 64..128 rows/tile 160 us, 192 rows 191 us (C2 = 64)
    TP = rmax / k;
    if (TP < 1) TP = 1;
    Rpad = (TP * k + 31) & ~31;
    const int ntiles = (N + TP - 1) / TP;
    G = 1024 / (B > 0 ? B : 1);  // about four workgroups per CU in flight
    if (G < 1) G = 1;
    if (G > ntiles) G = ntiles;
}

extern "C" size_t fsg_edgeconv2_workspace_bytes(int B, int N, int k, int C2) {
    int TP, Rpad, G;
    ec2_tiling(k, C2, TP, Rpad, G, B, N);
    if (ec2_split(C2)) {
        int RT, TPs, Gs;
        ec2s_tiling(k, B, N, false, false, RT, TPs, Gs);   // (sizes: the larger of the fp32 / bf16 grids)
        { int RTb, TPb, Gb; ec2s_tiling(k, B, N, false, true, RTb, TPb, Gb); if (Gb > Gs) Gs = Gb; }
        if (Gs > G) G = Gs;
    }
    const size_t rec1 = (size_t)fsg_ec_stats1_records(B, N) * 3 * C1 + fsg_ec_finalize_stage_floats(C1);
    const size_t rec2 = (size_t)B * G * 3 * C2 + fsg_ec_finalize_stage_floats(C2);
    // layer-1 scratch: ysel1 (unused output of the shared stats kernel), arg1
    const size_t scratch1 = (size_t)B * N * C1 * sizeof(float) + (size_t)B * N * C1;
    return sizeof(float) * (rec1 + rec2) + scratch1 + 256;
}

static int ec2_fwd_impl(bool bf16, const float *pq, const int32_t *idx, const float *w2, const float *gamma1,
                                     const float *beta1, float *running_mean1, float *running_var1, const float *gamma2,
                                     const float *beta2, float *running_mean2, float *running_var2, int B, int N, int k,
                                     int C2, int training, float momentum1, float momentum2, float eps1, float eps2,
                                     float slope, float *out, float *out_pm, float *ssum1, float *mean1, float *invstd1,
                                     float *ysel2, uint8_t *arg2, float *ssum2, float *mean2, float *invstd2,
                                     void *workspace, fsg_stream_t stream) {
    FSG_REQUIRE(pq && idx && w2 && gamma1 && beta1 && gamma2 && beta2 && mean1 && invstd1 && ysel2 && arg2 &&
                    mean2 && invstd2 && workspace,
                "fsg_edgeconv2_fwd_f32: NULL pointer");
    FSG_REQUIRE(B > 0 && N > 0 && N <= (1 << 21) && k > 0 && k <= 64 && (C2 == 64 || C2 == 128) && B <= 65535,
                "fsg_edgeconv2_fwd_f32: bad shape B=%d N=%d k=%d C2=%d (layer widths 64 -> 64|128 only, N <= 2^21)", B, N, k, C2);
    FSG_REQUIRE(!training || (ssum1 && ssum2), "fsg_edgeconv2_fwd_f32: training needs ssum1/ssum2");
    hipStream_t st = (hipStream_t)stream;
    int TP, Rpad, G;
    ec2_tiling(k, C2, TP, Rpad, G, B, N);
    int RTs = 0, TPs = 0, Gs = 0, Gws = G;     // Gws: the record count the workspace layout is sized for
    if (ec2_split(C2)) {
        ec2s_tiling(k, B, N, false, bf16, RTs, TPs, Gs);
        if (Gs > Gws) Gws = Gs;
    }
    const int rec1 = fsg_ec_stats1_records(B, N);
    float *part1 = (float *)workspace;
    float *part2 = part1 + (size_t)rec1 * 3 * C1 + fsg_ec_finalize_stage_floats(C1);
    float *ysel1 = part2 + (size_t)B * Gws * 3 * C2 + fsg_ec_finalize_stage_floats(C2);
    uint8_t *arg1 = (uint8_t *)(ysel1 + (size_t)B * N * C1);
    int rc;
    if (training) {  // BN1 statistics over all edges of y1 = P_j + Q_i (shared kernel; its selection output is unused)
        if ((rc = fsg_ec_stats1_launch(pq, idx, gamma1, B, N, k, C1, ysel1, arg1, ssum1, part1, st)) != FSG_OK) return rc;
        if ((rc = fsg_ec_finalize_launch(part1, rec1, C1, eps1, momentum1, mean1, invstd1, running_mean1, running_var1,
                                         st)) != FSG_OK)
            return rc;
    }
    const size_t lds = sizeof(float) * ((size_t)C2 * LD1 + (size_t)Rpad * (C2 + 1) + 3 * 4 * C2);
#define FSG_EC2_FWD(CC, BFX)                                                                                             \
    do {                                                                                                                 \
        static FsgLdsGrant grant;                                                                                     \
        if (!grant.raise((const void *)ec2_fwd_kernel<CC, BFX>, 160 * 1024 - 512)) {                                  \
            fsg_set_error("fsg_edgeconv2_fwd: cannot raise dynamic LDS");                                             \
            return FSG_ERR_HIP;                                                                                       \
        }                                                                                                             \
        hipLaunchKernelGGL((ec2_fwd_kernel<CC, BFX>), dim3(B, G), dim3(256), lds, st, pq, idx, w2, gamma1, beta1, mean1,    \
                           invstd1, gamma2, N, k, TP, Rpad, training, slope, ysel2, arg2, ssum2, part2);                 \
    } while (0)
#define FSG_EC2S_FWD(NPX, RTX)                                                                                            \
    do {                                                                                                                 \
        static FsgLdsGrant grant;                                                                                     \
        if (!grant.raise((const void *)ec2s_fwd_kernel<NPX, RTX>, 160 * 1024 - 512)) {                                \
            fsg_set_error("fsg_edgeconv2_fwd: cannot raise dynamic LDS");                                             \
            return FSG_ERR_HIP;                                                                                       \
        }                                                                                                             \
        hipLaunchKernelGGL((ec2s_fwd_kernel<NPX, RTX>), dim3(B, G), dim3(256), ec2s_fwd_lds(NPX, RTX, TPs), st, pq, idx, w2,    \
                           gamma1, beta1, mean1, invstd1, gamma2, N, k, TPs, training, slope, ysel2, arg2, ssum2, part2); \
    } while (0)
    if (ec2_split(C2)) {
        G = Gs;
        if (bf16) { if (RTs == 4) FSG_EC2S_FWD(1, 4); else FSG_EC2S_FWD(1, 2); }
        else { if (RTs == 4) FSG_EC2S_FWD(3, 4); else FSG_EC2S_FWD(3, 2); }
    } else if (C2 == 64) { if (bf16) FSG_EC2_FWD(64, true); else FSG_EC2_FWD(64, false); }
    else { if (bf16) FSG_EC2_FWD(128, true); else FSG_EC2_FWD(128, false); }
#undef FSG_EC2_FWD
#undef FSG_EC2S_FWD
    FSG_CHECK_LAUNCH("fsg_edgeconv2_fwd_f32/mlp");
    if (training) {
        if ((rc = fsg_ec_finalize_launch(part2, B * G, C2, eps2, momentum2, mean2, invstd2, running_mean2, running_var2,
                                         st)) != FSG_OK)
            return rc;
    }
    if (!out) return FSG_OK;      // the caller applies BatchNorm + LeakyReLU itself (fsg_edgeconv_apply_f32)
    return fsg_ec_apply_launch(ysel2, gamma2, beta2, mean2, invstd2, B, N, C2, slope, out, out_pm, st);
}

extern "C" int fsg_edgeconv2_fwd_f32(const float *pq, const int32_t *idx, const float *w2, const float *gamma1,
                                     const float *beta1, float *running_mean1, float *running_var1, const float *gamma2,
                                     const float *beta2, float *running_mean2, float *running_var2, int B, int N, int k,
                                     int C2, int training, float momentum1, float momentum2, float eps1, float eps2,
                                     float slope, float *out, float *out_pm, float *ssum1, float *mean1, float *invstd1,
                                     float *ysel2, uint8_t *arg2, float *ssum2, float *mean2, float *invstd2,
                                     void *workspace, fsg_stream_t stream) {
    return ec2_fwd_impl(false, pq, idx, w2, gamma1, beta1, running_mean1, running_var1, gamma2, beta2, running_mean2, running_var2, B, N, k, C2, training, momentum1, momentum2, eps1, eps2, slope, out, out_pm, ssum1, mean1, invstd1, ysel2, arg2, ssum2, mean2, invstd2, workspace, stream);
}

// same arguments and tensors (all fp32); the per-edge 64 x C2 contraction runs with bf16 operands on
// v_mfma_f32_32x32x16_bf16 (fp32 accumulation): BASELINE configs 3-5 name bf16 as their compute type
extern "C" int fsg_edgeconv2_fwd_bf16(const float *pq, const int32_t *idx, const float *w2, const float *gamma1,
                                     const float *beta1, float *running_mean1, float *running_var1, const float *gamma2,
                                     const float *beta2, float *running_mean2, float *running_var2, int B, int N, int k,
                                     int C2, int training, float momentum1, float momentum2, float eps1, float eps2,
                                     float slope, float *out, float *out_pm, float *ssum1, float *mean1, float *invstd1,
                                     float *ysel2, uint8_t *arg2, float *ssum2, float *mean2, float *invstd2,
                                     void *workspace, fsg_stream_t stream) {
    return ec2_fwd_impl(true, pq, idx, w2, gamma1, beta1, running_mean1, running_var1, gamma2, beta2, running_mean2, running_var2, B, N, k, C2, training, momentum1, momentum2, eps1, eps2, slope, out, out_pm, ssum1, mean1, invstd1, ysel2, arg2, ssum2, mean2, invstd2, workspace, stream);
}

int fsg_ec_bwd_point_launch(const float *gout, const float *gout_pm, long ld_pm, const float *gout_pm2, long ld_pm2,
                            const float *ysel, const float *gamma,
                            const float *beta, const float *mean, const float *invstd, int B, int N, int Co, float slope,
                            float *h, float *partials, float *dbeta, float *dgamma, hipStream_t st);
int fsg_ec_sum_launch(const float *partials, int R, int L, int nvec, float *out0, float *out1, hipStream_t st);

static void ec2_bwd_tiling(int k, int C2, int &TP, int &Rpad, int &G, int B, int N) {
    // 64 edge rows per tile: two row tiles x two column tiles = one MFMA pair per wave in every phase (balanced), and
    // ~52 KB of LDS, i.e. three workgroups per CU (FSG_EC2_BWD_RMAX overrides for experiments)
    static int rmax_env = -1;
    if (rmax_env < 0) {
        const char *e = getenv("FSG_EC2_BWD_RMAX");
        rmax_env = e ? atoi(e) : 0;
    }
    const int rmax = rmax_env > 0 ? rmax_env : 64;
    TP = rmax / k;
    if (TP < 1) TP = 1;
    Rpad = (TP * k + 31) & ~31;
    const int ntiles = (N + TP - 1) / TP;
    G = 768 / (B > 0 ? B : 1);
    if (G < 1) G = 1;
    if (G > ntiles) G = ntiles;
}

extern "C" size_t fsg_edgeconv2_bwd_workspace_bytes(int B, int N, int k, int C2) {
    int TP, Rpad, G;
    ec2_bwd_tiling(k, C2, TP, Rpad, G, B, N);
    if (ec2_split(C2)) {
        int RT, TPs, Gs;
        ec2s_tiling(k, B, N, true, false, RT, TPs, Gs);
        if (Gs > G) G = Gs;
    }
    const size_t point_rec = (size_t)B * fsg_cdiv(N, 64) * 2 * C2;       // ec1_bwd_point partials
    const size_t dw = (size_t)B * G * C2 * C1, p1 = (size_t)B * G * 2 * C1;
    const size_t du = (size_t)B * N * k * C1;
    return sizeof(float) * (point_rec + dw + p1 + du + (size_t)B * N * C2) + 256;
}

static int ec2_bwd_impl(bool bf16, const float *grad_out, const float *grad_out_pm, int64_t ld_pm, const float *grad_out_pm2,
                                     int64_t ld_pm2, const float *pq, const int32_t *idx,
                                     const int32_t *rowptr, const int32_t *col, const float *w2, const float *gamma1,
                                     const float *beta1, const float *mean1, const float *invstd1, const float *ssum1,
                                     const float *gamma2, const float *beta2, const float *mean2, const float *invstd2,
                                     const float *ysel2, const uint8_t *arg2, int B, int N, int k, int C2, int training,
                                     float slope, float *grad_pq, float *grad_w2, float *grad_gamma1, float *grad_beta1,
                                     float *grad_gamma2, float *grad_beta2, void *workspace, fsg_stream_t stream) {
    FSG_REQUIRE((!grad_out_pm || ld_pm >= C2) && (!grad_out_pm2 || ld_pm2 >= C2),
                "fsg_edgeconv2_bwd_f32: row stride of a point-major gradient below C2=%d", C2);
    FSG_REQUIRE((grad_out || grad_out_pm || grad_out_pm2) && pq && idx && rowptr && col && w2 && gamma1 && beta1 && mean1 && invstd1 &&
                    gamma2 && beta2 && mean2 && invstd2 && ysel2 && arg2 && grad_pq && grad_w2 && grad_gamma1 &&
                    grad_beta1 && grad_gamma2 && grad_beta2 && workspace,
                "fsg_edgeconv2_bwd_f32: NULL pointer");
    FSG_REQUIRE(!training || ssum1, "fsg_edgeconv2_bwd_f32: training needs ssum1");
    FSG_REQUIRE(B > 0 && N > 0 && N <= (1 << 21) && k > 0 && k <= 64 && (C2 == 64 || C2 == 128) && B <= 65535,
                "fsg_edgeconv2_bwd_f32: bad shape B=%d N=%d k=%d C2=%d (N <= 2^21)", B, N, k, C2);
    hipStream_t st = (hipStream_t)stream;
    int TP, Rpad, G;
    ec2_bwd_tiling(k, C2, TP, Rpad, G, B, N);
    int RTs = 0, TPs = 0, Gs = 0, Gws = G;     // Gws: the record count the workspace layout is sized for
    if (ec2_split(C2)) {
        ec2s_tiling(k, B, N, true, bf16, RTs, TPs, Gs);
        if (Gs > Gws) Gws = Gs;
    }
    float *point_part = (float *)workspace;
    float *dw_part = point_part + (size_t)B * fsg_cdiv(N, 64) * 2 * C2;
    float *p1_part = dw_part + (size_t)B * Gws * C2 * C1;
    float *du1 = p1_part + (size_t)B * Gws * 2 * C1;
    float *h2 = du1 + (size_t)B * N * k * C1;
    int rc;
    // h2 = grad_out f'(u2) on the selected edge, dbeta2 / dgamma2
    if ((rc = fsg_ec_bwd_point_launch(grad_out, grad_out_pm, (long)ld_pm, grad_out_pm2, (long)ld_pm2, ysel2, gamma2, beta2,
                                      mean2, invstd2, B, N, C2, slope, h2,
                                      point_part, grad_beta2, grad_gamma2, st)) != FSG_OK)
        return rc;
    const float invM = 1.0f / ((float)B * (float)N * (float)k);
    const size_t lds = sizeof(float) * ((size_t)C2 * LD1 + (size_t)Rpad * LD1 + (size_t)Rpad * (C2 + 1) + (size_t)TP * C2 +
                                        2 * C1 + 2 * C1) + (size_t)TP * C2 + 2 * (size_t)Rpad + 16;
#define FSG_EC2_BWD(CC, BFX)                                                                                                \
    do {                                                                                                                 \
        static FsgLdsGrant grant;                                                                                     \
        if (!grant.raise((const void *)ec2_bwd_kernel<CC, BFX>, 160 * 1024 - 512)) {                                  \
            fsg_set_error("fsg_edgeconv2_bwd_f32: cannot raise dynamic LDS");                                         \
            return FSG_ERR_HIP;                                                                                       \
        }                                                                                                             \
        hipLaunchKernelGGL((ec2_bwd_kernel<CC, BFX>), dim3(B, G), dim3(256), lds, st, pq, idx, w2, gamma1, beta1, mean1, invstd1, \
                           gamma2, mean2, invstd2, grad_beta2, grad_gamma2, h2, arg2, N, k, TP, Rpad, training, invM,    \
                           slope, du1, dw_part, p1_part);                                                                \
    } while (0)
#define FSG_EC2S_BWD(NPX, RTX)                                                                                            \
    do {                                                                                                                 \
        static FsgLdsGrant grant;                                                                                     \
        if (!grant.raise((const void *)ec2s_bwd_kernel<NPX, RTX>, 160 * 1024 - 512)) {                                \
            fsg_set_error("fsg_edgeconv2_bwd_f32: cannot raise dynamic LDS");                                         \
            return FSG_ERR_HIP;                                                                                       \
        }                                                                                                             \
        hipLaunchKernelGGL((ec2s_bwd_kernel<NPX, RTX>), dim3(B, G), dim3(256), ec2s_bwd_lds(NPX, RTX, TPs), st, pq, idx,   \
                           w2, gamma1, beta1, mean1, invstd1, gamma2, mean2, invstd2, grad_beta2, grad_gamma2, h2, arg2, \
                           N, k, TPs, training, invM, slope, du1, dw_part, p1_part);                                     \
    } while (0)
    if (ec2_split(C2)) {
        G = Gs;
        FSG_REQUIRE(ec2s_bwd_lds(bf16 ? 1 : 3, RTs, TPs) <= 160 * 1024 - 512, "fsg_edgeconv2_bwd: k=%d needs too much LDS", k);
        if (bf16) { if (RTs == 4) FSG_EC2S_BWD(1, 4); else FSG_EC2S_BWD(1, 2); }
        else { if (RTs == 4) FSG_EC2S_BWD(3, 4); else FSG_EC2S_BWD(3, 2); }
    } else if (C2 == 64) { if (bf16) FSG_EC2_BWD(64, true); else FSG_EC2_BWD(64, false); }
    else { if (bf16) FSG_EC2_BWD(128, true); else FSG_EC2_BWD(128, false); }
#undef FSG_EC2_BWD
#undef FSG_EC2S_BWD
    FSG_CHECK_LAUNCH("fsg_edgeconv2_bwd_f32/mlp");
    if ((rc = fsg_ec_sum_launch(dw_part, B * G, C2 * C1, 1, grad_w2, nullptr, st)) != FSG_OK) return rc;
    if ((rc = fsg_ec_sum_launch(p1_part, B * G, C1, 2, grad_beta1, grad_gamma1, st)) != FSG_OK) return rc;
    hipLaunchKernelGGL(ec2_bwd_gather_kernel, dim3(B, fsg_cdiv(N, 4)), dim3(256), 0, st, pq, rowptr, col, du1, ssum1,
                       gamma1, mean1, invstd1, grad_beta1, grad_gamma1, N, k, training, invM, grad_pq);
    FSG_CHECK_LAUNCH("fsg_edgeconv2_bwd_f32/gather");
    return FSG_OK;
}

extern "C" int fsg_edgeconv2_bwd_f32(const float *grad_out, const float *grad_out_pm, int64_t ld_pm, const float *grad_out_pm2,
                                     int64_t ld_pm2, const float *pq, const int32_t *idx,
                                     const int32_t *rowptr, const int32_t *col, const float *w2, const float *gamma1,
                                     const float *beta1, const float *mean1, const float *invstd1, const float *ssum1,
                                     const float *gamma2, const float *beta2, const float *mean2, const float *invstd2,
                                     const float *ysel2, const uint8_t *arg2, int B, int N, int k, int C2, int training,
                                     float slope, float *grad_pq, float *grad_w2, float *grad_gamma1, float *grad_beta1,
                                     float *grad_gamma2, float *grad_beta2, void *workspace, fsg_stream_t stream) {
    return ec2_bwd_impl(false, grad_out, grad_out_pm, ld_pm, grad_out_pm2, ld_pm2, pq, idx, rowptr, col, w2, gamma1, beta1, mean1, invstd1, ssum1, gamma2, beta2, mean2, invstd2, ysel2, arg2, B, N, k, C2, training, slope, grad_pq, grad_w2, grad_gamma1, grad_beta1, grad_gamma2, grad_beta2, workspace, stream);
}

// the three per-tile products of the backward (y2 recompute, dz1 = dy2 W2, dW2 += dy2^T z1) with bf16 operands
extern "C" int fsg_edgeconv2_bwd_bf16(const float *grad_out, const float *grad_out_pm, int64_t ld_pm, const float *grad_out_pm2,
                                     int64_t ld_pm2, const float *pq, const int32_t *idx,
                                     const int32_t *rowptr, const int32_t *col, const float *w2, const float *gamma1,
                                     const float *beta1, const float *mean1, const float *invstd1, const float *ssum1,
                                     const float *gamma2, const float *beta2, const float *mean2, const float *invstd2,
                                     const float *ysel2, const uint8_t *arg2, int B, int N, int k, int C2, int training,
                                     float slope, float *grad_pq, float *grad_w2, float *grad_gamma1, float *grad_beta1,
                                     float *grad_gamma2, float *grad_beta2, void *workspace, fsg_stream_t stream) {
    return ec2_bwd_impl(true, grad_out, grad_out_pm, ld_pm, grad_out_pm2, ld_pm2, pq, idx, rowptr, col, w2, gamma1, beta1, mean1, invstd1, ssum1, gamma2, beta2, mean2, invstd2, ysel2, arg2, B, N, k, C2, training, slope, grad_pq, grad_w2, grad_gamma1, grad_beta1, grad_gamma2, grad_beta2, workspace, stream);
}

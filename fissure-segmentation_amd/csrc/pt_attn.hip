// Fused PointTransformerLayer body -- include/fsg_hip.h: fsg_pt_attn_fwd_f32 / fsg_pt_attn_bwd_f32.
// Replaces models/pointtransformer/seg_model.py:38-53 (grouping, linear_p, linear_w, softmax, shared-plane aggregate).
//
// Per edge (i,s), j = idx[i,s]  (c channels, cs = c/8):
//     d  = p_j - p_i                      a  = W1 d + b1          t  = relu(BNp(a))            (3)
//     pr = W2 t + b2                      w0 = (k_j - q_i) + pr   h1 = relu(BN1(w0))           (c)
//     u1 = Wa h1 + ba                     h2 = relu(BN2(u1))      u2 = Wb h2 + bb              (cs)
//     sm = softmax_s(u2)                  out_i[ch] = sum_s (v_j[ch] + pr[ch]) sm[s, ch mod cs]
// The reference materialises ~10 tensors of shape (n,ns,c) per layer and runs ~40 kernels; here only the cs-wide
// tensors u1 and sm (1/8 of that) are written, everything c-wide is recomputed from the per-POINT rows q, k, v that
// stay in L2.  Train-mode BatchNorm needs the statistics of all n*ns edges before anything downstream, hence one
// pass per BatchNorm:   P0 stats(a) | P1 stats(w0) | P2 u1 + stats(u1) | P3 softmax + aggregate,
// and the mirror image in the backward (each BatchNorm backward needs its d_gamma, d_beta sums first):
//     B1 dv, d softmax, d(u2) -> dWb, dz2, (dg2, db2) | B2 du1 -> dWa, (dg1, db1) | B3 dw0 -> dk, dq, dW2, dzp, (dgp, dbp)
//     | B4 da -> dW1,   with a fixed-order record reduction after each.
// Work decomposition: a workgroup owns PT = NT/c consecutive points (NT = max(256, c) threads); thread (ps, ch) walks the
// ns neighbours of its point, so every gather k[j, :], v[j, :] is one coalesced row read and dq needs no atomics;
// dk, dv scatter with hardware fp32 atomics.  The only contraction over channels, u1 = Wa h1, runs on the matrix
// cores: each wave keeps its k-slice of Wa in registers as the A operand of v_mfma_f32_16x16x4_f32 and reads h1 from
// LDS; the per-wave partial products are summed in wave order.
#include "fsg_common.h"

#ifndef FSG_PT_GMAX_SCALE
#define FSG_PT_GMAX_SCALE 1
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef fsg_pt_layer_params Prm;

constexpr int MAXNS = 16;

// workgroups per launch at most (each leaves one record per BatchNorm / parameter sum): four per CU -- a tile is a chain of
// barrier-separated phases with dependent gathers, and with one workgroup per CU (round 3) nothing hid them: 7.58 -> 6.84 ms per
// config-3 step; 8 and 16 per CU measure 6.90.  FSG_PT_GMAX_SCALE (tools only) scales it.
constexpr int pt_gmax(int c) { return (c <= 128 ? 1024 : (c == 256 ? 512 : 256)) * FSG_PT_GMAX_SCALE; }

template <int C>
struct Geo {
    static constexpr int NT = C < 256 ? 256 : C;
    static constexpr int PT = NT / C;          // points per tile
    static constexpr int CS = C / 8;
    static constexpr int WAVES = NT / 64;
    static constexpr int KSL = C / WAVES;      // channels of Wa per wave (split-K over the waves)
    static constexpr int KST = KSL / 4;
    static constexpr int OB = (CS + 15) / 16;
    static constexpr int CSP = OB * 16;
    static constexpr int HS = C + 4;           // LDS row stride of the h1 tile
    static constexpr int EMAX = PT * MAXNS;
    static constexpr int GMAX = pt_gmax(C);
};

struct Stats {
    const float *mp, *rp, *m1, *r1, *m2, *r2;
};
__host__ __device__ inline Stats split_stats(const float *s, int c) {
    const int cs = c / 8;
    return Stats{s, s + 3, s + 6, s + 6 + c, s + 6 + 2 * c, s + 6 + 2 * c + cs};
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// positional front end of one tile: neighbour ids and t = relu(BNp(W1 d + b1)); optionally a-hat and d
template <bool FULL>
__device__ __forceinline__ void stage_front(const float *__restrict__ p, const int32_t *__restrict__ idx, const Prm &P,
                                            const Stats &S, int n, int ns, int pt0, int E, int *IDX, float *T, float *AH,
                                            float *D) {
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const int ps = e / ns, pt = pt0 + ps;
        int j = 0;
        float t[3] = {0.f, 0.f, 0.f}, ah[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
        if (pt < n) {
            j = idx[(long)pt0 * ns + e];
#pragma unroll
            for (int m = 0; m < 3; ++m) d[m] = p[3L * j + m] - p[3L * pt + m];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const float a = P.lp1_b[m] + P.lp1_w[3 * m] * d[0] + P.lp1_w[3 * m + 1] * d[1] + P.lp1_w[3 * m + 2] * d[2];
                ah[m] = (a - S.mp[m]) * S.rp[m];
                t[m] = fmaxf(P.bnp_g[m] * ah[m] + P.bnp_b[m], 0.f);
            }
        }
        IDX[e] = j;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            T[4 * e + m] = t[m];
            if (FULL) { AH[4 * e + m] = ah[m]; D[4 * e + m] = d[m]; }
        }
    }
}

// ------------------------------------------------------------------------------------------------ P0: statistics of a
__global__ __launch_bounds__(256) void pt_stats_p_kernel(const float *__restrict__ p, const int32_t *__restrict__ idx,
                                                         Prm P, int n, int ns, double *__restrict__ rec) {
    __shared__ double red[4][6];
    const long M = (long)n * ns;
    double s[3] = {0, 0, 0}, ss[3] = {0, 0, 0};
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < M; e += (long)gridDim.x * 256) {
        const long pt = e / ns;
        const int j = idx[e];
        float d[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) d[m] = p[3L * j + m] - p[3L * pt + m];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const float a = P.lp1_b[m] + P.lp1_w[3 * m] * d[0] + P.lp1_w[3 * m + 1] * d[1] + P.lp1_w[3 * m + 2] * d[2];
            s[m] += a;
            ss[m] += (double)a * a;
        }
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const double a = wave_sum_d(s[m]), b = wave_sum_d(ss[m]);
        if (lane == 0) { red[wave][m] = a; red[wave][3 + m] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6) rec[(long)blockIdx.x * 6 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// records [R][2][L] (sum | sum of squares) -> mean, rstd (+ running buffers, torch's rule).  One wave per channel:
// the lanes stride over the records (fixed assignment, fixed shuffle tree -> reproducible).
__global__ __launch_bounds__(256) void pt_bn_finalize_kernel(const double *__restrict__ rec, int R, int L, double M,
                                                             float eps, float mom, float *__restrict__ mean,
                                                             float *__restrict__ rstd, float *__restrict__ rm,
                                                             float *__restrict__ rv) {
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (l >= L) return;
    double S = 0, SS = 0;
    for (int r = lane; r < R; r += 64) {
        S += rec[(long)r * 2 * L + l];
        SS += rec[(long)r * 2 * L + L + l];
    }
    S = wave_sum_d(S);
    SS = wave_sum_d(SS);
    if (lane != 0) return;
    const double m = S / M;
    double var = SS / M - m * m;
    if (var < 0) var = 0;
    mean[l] = (float)m;
    rstd[l] = (float)(1.0 / sqrt(var + (double)eps));
    if (rm) rm[l] = (float)((1.0 - mom) * rm[l] + mom * m);
    if (rv) rv[l] = (float)((1.0 - mom) * rv[l] + mom * (M > 1 ? var * M / (M - 1) : var));
}

// ------------------------------------------------------------------------------------------------ P1: statistics of w0
template <int C>
__global__ __launch_bounds__(Geo<C>::NT) void pt_stats_w_kernel(const float *__restrict__ p, const int32_t *__restrict__ idx,
                                                                const float *__restrict__ q, const float *__restrict__ k,
                                                                long ld, Prm P, const float *__restrict__ stats, int n,
                                                                int ns, double *__restrict__ rec) {
    typedef Geo<C> G;
    __shared__ int IDX[G::EMAX];
    __shared__ float T[G::EMAX * 4];
    __shared__ double RED[G::NT * 2];
    const Stats S = split_stats(stats, C);
    const int tid = threadIdx.x, ps = tid / C, ch = tid % C;
    const float w20 = P.lp2_w[3 * ch], w21 = P.lp2_w[3 * ch + 1], w22 = P.lp2_w[3 * ch + 2], b2 = P.lp2_b[ch];
    double s = 0, ss = 0;
    const int tiles = (n + G::PT - 1) / G::PT, E = G::PT * ns;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int pt0 = tile * G::PT;
        __syncthreads();
        stage_front<false>(p, idx, P, S, n, ns, pt0, E, IDX, T, nullptr, nullptr);
        __syncthreads();
        const int pt = pt0 + ps;
        if (pt < n) {
            const float qv = q[(long)pt * ld + ch];
            for (int sI = 0; sI < ns; ++sI) {
                const int e = ps * ns + sI, j = IDX[e];
                const float pr = b2 + w20 * T[4 * e] + w21 * T[4 * e + 1] + w22 * T[4 * e + 2];
                const float w0 = (k[(long)j * ld + ch] - qv) + pr;
                s += w0;
                ss += (double)w0 * w0;
            }
        }
    }
    RED[tid] = s;
    RED[G::NT + tid] = ss;
    __syncthreads();
    if (tid < C) {
        double a = 0, b = 0;
        for (int r = 0; r < G::PT; ++r) { a += RED[r * C + tid]; b += RED[G::NT + r * C + tid]; }
        rec[(long)blockIdx.x * 2 * C + tid] = a;
        rec[(long)blockIdx.x * 2 * C + C + tid] = b;
    }
}

// ------------------------------------------------------------------------------------------------ P2: u1 (+ statistics)
template <int C>
__global__ __launch_bounds__(Geo<C>::NT) void pt_u1_kernel(const float *__restrict__ p, const int32_t *__restrict__ idx,
                                                           const float *__restrict__ q, const float *__restrict__ k, long ld,
                                                           Prm P, const float *__restrict__ stats, int n, int ns,
                                                           float *__restrict__ u1, double *__restrict__ rec) {
    typedef Geo<C> G;
    constexpr int CS = G::CS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *IDX = reinterpret_cast<int *>(smem);                   // [EMAX]
    float *T = reinterpret_cast<float *>(IDX + G::EMAX);        // [EMAX][4]
    float *H = T + G::EMAX * 4;                                 // [EP][HS]; afterwards the per-wave partials [WAVES][CSP][EP]
    const Stats S = split_stats(stats, C);
    const int tid = threadIdx.x, ps = tid / C, ch = tid % C;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const float w20 = P.lp2_w[3 * ch], w21 = P.lp2_w[3 * ch + 1], w22 = P.lp2_w[3 * ch + 2], b2 = P.lp2_b[ch];
    const float m1 = S.m1[ch], r1 = S.r1[ch], g1 = P.bn1_g[ch], be1 = P.bn1_b[ch];
    // A operand: rows o of Wa, this wave's channel slice
    float areg[G::OB][G::KST];
#pragma unroll
    for (int ob = 0; ob < G::OB; ++ob)
#pragma unroll
        for (int st = 0; st < G::KST; ++st) {
            const int o = ob * 16 + l15;
            areg[ob][st] = o < CS ? P.lw1_w[(long)o * C + wave * G::KSL + 4 * st + l4] : 0.f;
        }
    double s = 0, ss = 0;
    const int tiles = (n + G::PT - 1) / G::PT, E = G::PT * ns, EP = (E + 15) & ~15, EB = EP >> 4;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int pt0 = tile * G::PT;
        __syncthreads();
        stage_front<false>(p, idx, P, S, n, ns, pt0, E, IDX, T, nullptr, nullptr);
        for (int t = tid; t < (EP - E) * C; t += G::NT) H[(E + t / C) * G::HS + t % C] = 0.f;
        __syncthreads();
        {
            const int pt = pt0 + ps;
            const float qv = pt < n ? q[(long)pt * ld + ch] : 0.f;
            for (int sI = 0; sI < ns; ++sI) {
                const int e = ps * ns + sI;
                float h = 0.f;
                if (pt < n) {
                    const int j = IDX[e];
                    const float pr = b2 + w20 * T[4 * e] + w21 * T[4 * e + 1] + w22 * T[4 * e + 2];
                    const float w0 = (k[(long)j * ld + ch] - qv) + pr;
                    h = fmaxf(g1 * ((w0 - m1) * r1) + be1, 0.f);
                }
                H[e * G::HS + ch] = h;
            }
        }
        __syncthreads();
        f32x4 acc[G::PT][G::OB];
#pragma unroll
        for (int eb = 0; eb < G::PT; ++eb) {
            if (eb < EB) {
                float bv[G::KST];
#pragma unroll
                for (int st = 0; st < G::KST; ++st) bv[st] = H[(eb * 16 + l15) * G::HS + wave * G::KSL + 4 * st + l4];
#pragma unroll
                for (int ob = 0; ob < G::OB; ++ob) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int st = 0; st < G::KST; ++st) a = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[ob][st], bv[st], a, 0, 0, 0);
                    acc[eb][ob] = a;
                }
            }
        }
        __syncthreads();  // every wave is done with H: reuse it for the partial products
        float *PU = H;
#pragma unroll
        for (int eb = 0; eb < G::PT; ++eb)
            if (eb < EB)
#pragma unroll
                for (int ob = 0; ob < G::OB; ++ob)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        PU[((wave * G::CSP) + ob * 16 + l4 * 4 + r) * EP + eb * 16 + l15] = acc[eb][ob][r];
        __syncthreads();
        for (int item = tid; item < E * CS; item += G::NT) {
            const int e = item / CS, o = item % CS;
            const int pt = pt0 + e / ns;
            if (pt < n) {
                float u = P.lw1_b[o];
#pragma unroll
                for (int w = 0; w < G::WAVES; ++w) u += PU[(w * G::CSP + o) * EP + e];
                u1[((long)pt0 * ns + e) * CS + o] = u;
                s += u;
                ss += (double)u * u;
            }
        }
    }
    if (rec) {
        __syncthreads();
        double *RED = reinterpret_cast<double *>(H);
        RED[tid] = s;
        RED[G::NT + tid] = ss;
        __syncthreads();
        if (tid < CS) {  // thread tid owns output o = tid % CS for all its items (NT % CS == 0)
            double a = 0, b = 0;
            for (int r = tid; r < G::NT; r += CS) { a += RED[r]; b += RED[G::NT + r]; }
            rec[(long)blockIdx.x * 2 * CS + tid] = a;
            rec[(long)blockIdx.x * 2 * CS + CS + tid] = b;
        }
    }
}

template <int C>
constexpr size_t u1_lds_bytes() {
    typedef Geo<C> G;
    const size_t h = (size_t)G::EMAX * G::HS, pu = (size_t)G::WAVES * G::CSP * G::EMAX, red = (size_t)G::NT * 4;
    size_t m = h > pu ? h : pu;
    m = m > red ? m : red;
    return sizeof(int) * G::EMAX + sizeof(float) * (G::EMAX * 4 + m);
}

// ------------------------------------------------------------------------------------------------ P3: softmax + aggregate
template <int C>
__global__ __launch_bounds__(Geo<C>::NT) void pt_out_kernel(const float *__restrict__ p, const int32_t *__restrict__ idx,
                                                            const float *__restrict__ v, long ld, Prm P,
                                                            const float *__restrict__ stats, int n, int ns,
                                                            const float *__restrict__ u1, float *__restrict__ sm,
                                                            float *__restrict__ out) {
    typedef Geo<C> G;
    constexpr int CS = G::CS, RS = CS + 1;
    __shared__ int IDX[G::EMAX];
    __shared__ float T[G::EMAX * 4];
    __shared__ float H2[G::EMAX * RS], U2[G::EMAX * RS], WbT[CS * RS];
    const Stats S = split_stats(stats, C);
    const int tid = threadIdx.x, ps = tid / C, ch = tid % C;
    const float w20 = P.lp2_w[3 * ch], w21 = P.lp2_w[3 * ch + 1], w22 = P.lp2_w[3 * ch + 2], b2 = P.lp2_b[ch];
    for (int t = tid; t < CS * CS; t += G::NT) WbT[(t % CS) * RS + t / CS] = P.lw2_w[t];  // WbT[o'][o] = Wb[o][o']
    const int tiles = (n + G::PT - 1) / G::PT, E = G::PT * ns;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int pt0 = tile * G::PT;
        __syncthreads();
        stage_front<false>(p, idx, P, S, n, ns, pt0, E, IDX, T, nullptr, nullptr);
        for (int item = tid; item < E * CS; item += G::NT) {
            const int e = item / CS, o = item % CS;
            float h = 0.f;
            if (pt0 + e / ns < n) {
                const float u = u1[((long)pt0 * ns + e) * CS + o];
                h = fmaxf(P.bn2_g[o] * ((u - S.m2[o]) * S.r2[o]) + P.bn2_b[o], 0.f);
            }
            H2[e * RS + o] = h;
        }
        __syncthreads();
        for (int item = tid; item < E * CS; item += G::NT) {
            const int e = item / CS, o = item % CS;
            float u = P.lw2_b[o];
            for (int o2 = 0; o2 < CS; ++o2) u += WbT[o2 * RS + o] * H2[e * RS + o2];
            U2[e * RS + o] = u;
        }
        __syncthreads();
        if (tid < G::PT * CS) {  // softmax over the neighbours, one thread per (point, shared channel)
            const int pp = tid / CS, o = tid % CS;
            float mx = -INFINITY;
            for (int sI = 0; sI < ns; ++sI) mx = fmaxf(mx, U2[(pp * ns + sI) * RS + o]);
            float den = 0.f;
            for (int sI = 0; sI < ns; ++sI) {
                const float ex = expf(U2[(pp * ns + sI) * RS + o] - mx);
                U2[(pp * ns + sI) * RS + o] = ex;
                den += ex;
            }
            const float inv = 1.0f / den;
            const bool ok = pt0 + pp < n;
            for (int sI = 0; sI < ns; ++sI) {
                const int e = pp * ns + sI;
                const float w = U2[e * RS + o] * inv;
                U2[e * RS + o] = w;
                if (ok) sm[((long)pt0 * ns + e) * CS + o] = w;
            }
        }
        __syncthreads();
        const int pt = pt0 + ps;
        if (pt < n) {
            float acc = 0.f;
            const int o = ch % CS;
            for (int sI = 0; sI < ns; ++sI) {
                const int e = ps * ns + sI, j = IDX[e];
                const float pr = b2 + w20 * T[4 * e] + w21 * T[4 * e + 1] + w22 * T[4 * e + 2];
                acc += (v[(long)j * ld + ch] + pr) * U2[e * RS + o];
            }
            out[(long)pt * C + ch] = acc;
        }
    }
}

// ================================================================================================ backward
// B1: dv, d softmax, du2 -> dz2 (kept), records [dWb | dbb | dg2 | db2]
template <int C>
__global__ __launch_bounds__(Geo<C>::NT) void pt_b1_kernel(const float *__restrict__ p, const int32_t *__restrict__ idx,
                                                           const float *__restrict__ v, long ld, Prm P,
                                                           const float *__restrict__ stats, int n, int ns,
                                                           const float *__restrict__ g, const float *__restrict__ u1,
                                                           const float *__restrict__ sm, float *__restrict__ dv, long ldg,
                                                           float *__restrict__ dz2, float *__restrict__ rec) {
    typedef Geo<C> G;
    constexpr int CS = G::CS, RS = CS + 1, NPAIR = (CS * CS + G::NT - 1) / G::NT;
    __shared__ int IDX[G::EMAX];
    __shared__ float T[G::EMAX * 4];
    __shared__ float SM[G::EMAX * RS], DS[G::EMAX * RS], H2[G::EMAX * RS], UH[G::EMAX * RS], WB[CS * RS];
    __shared__ float PROD[MAXNS * G::NT];   // g (v + p_r) of every (neighbour, point slot, channel) of the tile
    const Stats S = split_stats(stats, C);
    const int tid = threadIdx.x, ps = tid / C, ch = tid % C;
    const float w20 = P.lp2_w[3 * ch], w21 = P.lp2_w[3 * ch + 1], w22 = P.lp2_w[3 * ch + 2], b2 = P.lp2_b[ch];
    for (int t = tid; t < CS * CS; t += G::NT) WB[(t / CS) * RS + t % CS] = P.lw2_w[t];
    float dwb[NPAIR];
#pragma unroll
    for (int r = 0; r < NPAIR; ++r) dwb[r] = 0.f;
    float dbb = 0.f, dg2 = 0.f, db2 = 0.f;  // this thread's output o = tid % CS
    const int tiles = (n + G::PT - 1) / G::PT, E = G::PT * ns;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int pt0 = tile * G::PT;
        __syncthreads();
        stage_front<false>(p, idx, P, S, n, ns, pt0, E, IDX, T, nullptr, nullptr);
        for (int item = tid; item < E * CS; item += G::NT) {
            const int e = item / CS, o = item % CS;
            float w = 0.f, uh = 0.f, h = 0.f;
            if (pt0 + e / ns < n) {
                const long gi = ((long)pt0 * ns + e) * CS + o;
                w = sm[gi];
                uh = (u1[gi] - S.m2[o]) * S.r2[o];
                h = fmaxf(P.bn2_g[o] * uh + P.bn2_b[o], 0.f);
            }
            SM[e * RS + o] = w;
            UH[e * RS + o] = uh;
            H2[e * RS + o] = h;
        }
        __syncthreads();
        const int pt = pt0 + ps;
        const float gch = pt < n ? g[(long)pt * C + ch] : 0.f;
        for (int sI = 0; sI < ns; ++sI) {
            const int e = ps * ns + sI;
            float prod = 0.f;
            if (pt < n) {
                const int j = IDX[e];
                const float pr = b2 + w20 * T[4 * e] + w21 * T[4 * e + 1] + w22 * T[4 * e + 2];
                unsafeAtomicAdd(dv + (long)j * ldg + ch, gch * SM[e * RS + ch % CS]);
                prod = gch * (v[(long)j * ld + ch] + pr);
            }
            PROD[sI * G::NT + tid] = prod;
        }
        __syncthreads();   // one barrier for all neighbours (was two per neighbour)
        for (int item = tid; item < ns * G::PT * CS; item += G::NT) {   // d softmax[e][o] = sum of the 8 channels sharing o
            const int sI = item / (G::PT * CS), rem = item % (G::PT * CS), pp = rem / CS, o = rem % CS;
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) a += PROD[sI * G::NT + pp * C + r * CS + o];
            DS[(pp * ns + sI) * RS + o] = a;
        }
        __syncthreads();
        if (tid < G::PT * CS) {  // softmax backward over the neighbours
            const int pp = tid / CS, o = tid % CS;
            float dot = 0.f;
            for (int sI = 0; sI < ns; ++sI) dot += SM[(pp * ns + sI) * RS + o] * DS[(pp * ns + sI) * RS + o];
            for (int sI = 0; sI < ns; ++sI) {
                const int e = pp * ns + sI;
                DS[e * RS + o] = SM[e * RS + o] * (DS[e * RS + o] - dot);   // du2
            }
        }
        __syncthreads();
        for (int item = tid; item < E * CS; item += G::NT) {
            const int e = item / CS, o = item % CS;   // o plays o' here
            if (pt0 + e / ns < n) {
                float dh = 0.f;
                for (int o1 = 0; o1 < CS; ++o1) dh += WB[o1 * RS + o] * DS[e * RS + o1];
                const float dz = H2[e * RS + o] > 0.f ? dh : 0.f;
                dz2[((long)pt0 * ns + e) * CS + o] = dz;
                db2 += dz;
                dg2 += dz * UH[e * RS + o];
                dbb += DS[e * RS + o];
            }
        }
#pragma unroll
        for (int r = 0; r < NPAIR; ++r) {
            const int pr_ = r * G::NT + tid;
            if (pr_ < CS * CS) {
                const int o = pr_ / CS, o2 = pr_ % CS;
                float a = 0.f;
                for (int e = 0; e < E; ++e) a += DS[e * RS + o] * H2[e * RS + o2];   // rows of padded points are zero
                dwb[r] += a;
            }
        }
    }
    // records
    const int L = CS * CS + 3 * CS;
    float *my = rec + (long)blockIdx.x * L;
#pragma unroll
    for (int r = 0; r < NPAIR; ++r)
        if (r * G::NT + tid < CS * CS) my[r * G::NT + tid] = dwb[r];
    __syncthreads();
    // the three per-output sums go through PROD one after the other (NT floats)
    for (int which = 0; which < 3; ++which) {
        PROD[tid] = which == 0 ? dbb : (which == 1 ? dg2 : db2);
        __syncthreads();
        if (tid < CS) {
            float a = 0.f;
            for (int r = tid; r < G::NT; r += CS) a += PROD[r];
            my[CS * CS + which * CS + tid] = a;
        }
        __syncthreads();
    }
}

// du1 tile from dz2, u1 and the reduced (dg2, db2)
template <int C>
__device__ __forceinline__ void stage_du1(const Prm &P, const Stats &S, const float *__restrict__ u1,
                                          const float *__restrict__ dz2, const float *__restrict__ dg2,
                                          const float *__restrict__ db2, float invM, int n, int ns, int pt0, int E,
                                          float *DU1) {
    typedef Geo<C> G;
    constexpr int CS = G::CS, RS = CS + 1;
    for (int item = threadIdx.x; item < E * CS; item += G::NT) {
        const int e = item / CS, o = item % CS;
        float du = 0.f;
        if (pt0 + e / ns < n) {
            const long gi = ((long)pt0 * ns + e) * CS + o;
            const float uh = (u1[gi] - S.m2[o]) * S.r2[o];
            du = P.bn2_g[o] * S.r2[o] * (dz2[gi] - invM * db2[o] - uh * (invM * dg2[o]));
        }
        DU1[e * RS + o] = du;
    }
}

// B2: du1 -> dh1 -> dz1 sums, dWa, dba.  records per workgroup: [dWa (cs,c) | dg1 | db1 | dba]
template <int C>
__global__ __launch_bounds__(Geo<C>::NT) void pt_b2_kernel(const float *__restrict__ p, const int32_t *__restrict__ idx,
                                                           const float *__restrict__ q, const float *__restrict__ k, long ld,
                                                           Prm P, const float *__restrict__ stats, int n, int ns,
                                                           const float *__restrict__ u1, const float *__restrict__ dz2,
                                                           const float *__restrict__ dg2, const float *__restrict__ db2,
                                                           float invM, float *__restrict__ rec) {
    typedef Geo<C> G;
    constexpr int CS = G::CS, RS = CS + 1;
    __shared__ int IDX[G::EMAX];
    __shared__ float T[G::EMAX * 4];
    __shared__ float DU1[G::EMAX * RS];
    __shared__ float COMB[G::PT > 1 ? (CS + 2) * G::NT : 1];
    const Stats S = split_stats(stats, C);
    const int tid = threadIdx.x, ps = tid / C, ch = tid % C;
    const float w20 = P.lp2_w[3 * ch], w21 = P.lp2_w[3 * ch + 1], w22 = P.lp2_w[3 * ch + 2], b2 = P.lp2_b[ch];
    const float m1 = S.m1[ch], r1 = S.r1[ch], g1 = P.bn1_g[ch], be1 = P.bn1_b[ch];
    float wa[CS], dwa[CS];
#pragma unroll
    for (int o = 0; o < CS; ++o) { wa[o] = P.lw1_w[(long)o * C + ch]; dwa[o] = 0.f; }
    float dg1 = 0.f, db1 = 0.f, dba = 0.f;
    const int tiles = (n + G::PT - 1) / G::PT, E = G::PT * ns;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int pt0 = tile * G::PT;
        __syncthreads();
        stage_front<false>(p, idx, P, S, n, ns, pt0, E, IDX, T, nullptr, nullptr);
        stage_du1<C>(P, S, u1, dz2, dg2, db2, invM, n, ns, pt0, E, DU1);
        __syncthreads();
        const int pt = pt0 + ps;
        if (pt < n) {
            const float qv = q[(long)pt * ld + ch];
            for (int sI = 0; sI < ns; ++sI) {
                const int e = ps * ns + sI, j = IDX[e];
                const float pr = b2 + w20 * T[4 * e] + w21 * T[4 * e + 1] + w22 * T[4 * e + 2];
                const float w0 = (k[(long)j * ld + ch] - qv) + pr;
                const float wh = (w0 - m1) * r1, z1 = g1 * wh + be1, h1 = fmaxf(z1, 0.f);
                float dh = 0.f;
#pragma unroll
                for (int o = 0; o < CS; ++o) {
                    const float du = DU1[e * RS + o];
                    dh += wa[o] * du;
                    dwa[o] += du * h1;
                }
                const float dz = z1 > 0.f ? dh : 0.f;
                db1 += dz;
                dg1 += dz * wh;
            }
        }
        if (tid < CS)
            for (int e = 0; e < E; ++e) dba += DU1[e * RS + tid];
    }
    const int L = CS * C + 2 * C + CS;
    float *my = rec + (long)blockIdx.x * L;
    if (G::PT == 1) {
#pragma unroll
        for (int o = 0; o < CS; ++o) my[o * C + ch] = dwa[o];
        my[CS * C + ch] = dg1;
        my[CS * C + C + ch] = db1;
    } else {  // fold the point slots of the workgroup: one record per workgroup
        __syncthreads();
#pragma unroll
        for (int o = 0; o < CS; ++o) COMB[o * G::NT + tid] = dwa[o];
        COMB[CS * G::NT + tid] = dg1;
        COMB[(CS + 1) * G::NT + tid] = db1;
        __syncthreads();
        for (int i = tid; i < (CS + 2) * C; i += G::NT) {
            const int vv = i / C, c2 = i % C;
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < G::PT; ++r) a += COMB[vv * G::NT + r * C + c2];
            my[i] = a;
        }
    }
    if (tid < CS) my[CS * C + 2 * C + tid] = dba;
}

// B3: dw0 -> dk (atomics), dq, dW2, db2, dt -> dzp (kept), (dgp, dbp).
// records per workgroup: [dW2 (c,3) | db2 (c) | dgp (3) | dbp (3)]
template <int C>
__global__ __launch_bounds__(Geo<C>::NT) void pt_b3_kernel(const float *__restrict__ p, const int32_t *__restrict__ idx,
                                                           const float *__restrict__ q, const float *__restrict__ k, long ld,
                                                           Prm P, const float *__restrict__ stats, int n, int ns,
                                                           const float *__restrict__ g, const float *__restrict__ u1,
                                                           const float *__restrict__ sm, const float *__restrict__ dz2,
                                                           const float *__restrict__ dg2, const float *__restrict__ db2,
                                                           const float *__restrict__ dg1v, const float *__restrict__ db1v,
                                                           float invM, float *__restrict__ dq, float *__restrict__ dk,
                                                           long ldg, float *__restrict__ dzp, float *__restrict__ rec) {
    typedef Geo<C> G;
    constexpr int CS = G::CS, RS = CS + 1, HALVES = G::NT / 32, GRP = C / 32;
    __shared__ int IDX[G::EMAX];
    __shared__ float T[G::EMAX * 4], AH[G::EMAX * 4], D[G::EMAX * 4], DT[G::EMAX * 4];
    __shared__ float DU1[G::EMAX * RS], SM[G::EMAX * RS];
    __shared__ float REDH[MAXNS * HALVES * 4];   // per neighbour: the 32-lane partial sums of W2^T dpr
    __shared__ float REDP[G::NT / 64][8];
    __shared__ float COMB[4 * G::NT];
    const Stats S = split_stats(stats, C);
    const int tid = threadIdx.x, ps = tid / C, ch = tid % C;
    const float w20 = P.lp2_w[3 * ch], w21 = P.lp2_w[3 * ch + 1], w22 = P.lp2_w[3 * ch + 2], b2 = P.lp2_b[ch];
    const float m1 = S.m1[ch], r1 = S.r1[ch], g1 = P.bn1_g[ch], be1 = P.bn1_b[ch];
    const float cg = invM * dg1v[ch], cb = invM * db1v[ch];
    float wa[CS];
#pragma unroll
    for (int o = 0; o < CS; ++o) wa[o] = P.lw1_w[(long)o * C + ch];
    float dw2[3] = {0.f, 0.f, 0.f}, db2a = 0.f;
    float dgp[3] = {0.f, 0.f, 0.f}, dbp[3] = {0.f, 0.f, 0.f};
    const int tiles = (n + G::PT - 1) / G::PT, E = G::PT * ns;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int pt0 = tile * G::PT;
        __syncthreads();
        stage_front<true>(p, idx, P, S, n, ns, pt0, E, IDX, T, AH, D);
        stage_du1<C>(P, S, u1, dz2, dg2, db2, invM, n, ns, pt0, E, DU1);
        for (int item = tid; item < E * CS; item += G::NT) {
            const int e = item / CS, o = item % CS;
            SM[e * RS + o] = pt0 + e / ns < n ? sm[((long)pt0 * ns + e) * CS + o] : 0.f;
        }
        __syncthreads();
        const int pt = pt0 + ps;
        const bool ok = pt < n;
        const float qv = ok ? q[(long)pt * ld + ch] : 0.f;
        const float gch = ok ? g[(long)pt * C + ch] : 0.f;
        float dqa = 0.f;
        for (int sI = 0; sI < ns; ++sI) {
            const int e = ps * ns + sI;
            float pm[3] = {0.f, 0.f, 0.f};
            if (ok) {
                const int j = IDX[e];
                const float t0 = T[4 * e], t1 = T[4 * e + 1], t2 = T[4 * e + 2];
                const float pr = b2 + w20 * t0 + w21 * t1 + w22 * t2;
                const float w0 = (k[(long)j * ld + ch] - qv) + pr;
                const float wh = (w0 - m1) * r1, z1 = g1 * wh + be1;
                float dh = 0.f;
#pragma unroll
                for (int o = 0; o < CS; ++o) dh += wa[o] * DU1[e * RS + o];
                const float dz = z1 > 0.f ? dh : 0.f;
                const float dw0 = g1 * r1 * (dz - cb - wh * cg);
                unsafeAtomicAdd(dk + (long)j * ldg + ch, dw0);
                dqa -= dw0;
                const float dpr = gch * SM[e * RS + ch % CS] + dw0;
                dw2[0] += dpr * t0; dw2[1] += dpr * t1; dw2[2] += dpr * t2;
                db2a += dpr;
                pm[0] = w20 * dpr; pm[1] = w21 * dpr; pm[2] = w22 * dpr;
            }
            // dt[e][m] = sum over the channels of this point: 32-lane segments, then across the point's segments
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                float x = pm[m];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
                pm[m] = x;
            }
            if ((tid & 31) == 0) {
                float *rh = REDH + (sI * HALVES + (tid >> 5)) * 4;
                rh[0] = pm[0]; rh[1] = pm[1]; rh[2] = pm[2];
            }
        }
        __syncthreads();   // one barrier for all neighbours (was two per neighbour)
        for (int item = tid; item < ns * G::PT * 3; item += G::NT) {
            const int sI = item / (G::PT * 3), rem = item % (G::PT * 3), pp = rem / 3, m = rem % 3;
            float a = 0.f;
            for (int h = 0; h < GRP; ++h) a += REDH[(sI * HALVES + pp * GRP + h) * 4 + m];
            DT[(pp * ns + sI) * 4 + m] = a;
        }
        __syncthreads();
        if (ok) dq[(long)pt * ldg + ch] = dqa;
        if (tid < E && pt0 + tid / ns < n) {
            const int e = tid;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const float dz = T[4 * e + m] > 0.f ? DT[4 * e + m] : 0.f;
                dzp[((long)pt0 * ns + e) * 3 + m] = dz;
                dbp[m] += dz;
                dgp[m] += dz * AH[4 * e + m];
            }
        }
    }
    const int L = 4 * C + 6;
    float *my = rec + (long)blockIdx.x * L;
    __syncthreads();
    COMB[tid] = dw2[0]; COMB[G::NT + tid] = dw2[1]; COMB[2 * G::NT + tid] = dw2[2]; COMB[3 * G::NT + tid] = db2a;
    __syncthreads();
    for (int i = tid; i < 4 * C; i += G::NT) {
        const int vv = i / C, c2 = i % C;
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < G::PT; ++r) a += COMB[vv * G::NT + r * C + c2];
        my[vv < 3 ? 3 * c2 + vv : 3 * C + c2] = a;
    }
    // (dgp, dbp): block sum of the first EMAX threads' accumulators
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const float a = wave_sum_f(dgp[m]), b = wave_sum_f(dbp[m]);
        if (lane == 0) { REDP[wave][m] = a; REDP[wave][3 + m] = b; }
    }
    __syncthreads();
    if (tid < 6) {
        float a = 0.f;
        for (int w = 0; w < G::NT / 64; ++w) a += REDP[w][tid];
        my[4 * C + tid] = a;
    }
}

// B4: da -> dW1, db1.  records per block: [dW1 (3,3) | db1 (3)]
__global__ __launch_bounds__(256) void pt_b4_kernel(const float *__restrict__ p, const int32_t *__restrict__ idx, Prm P,
                                                    const float *__restrict__ stats, int n, int ns,
                                                    const float *__restrict__ dzp, const float *__restrict__ dgp,
                                                    const float *__restrict__ dbp, float invM, float *__restrict__ dp,
                                                    float *__restrict__ rec) {
    __shared__ float red[4][12];
    const long M = (long)n * ns;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < M; e += (long)gridDim.x * 256) {
        const long pt = e / ns;
        const int j = idx[e];
        float d[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) d[m] = p[3L * j + m] - p[3L * pt + m];
        float dd[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const float a = P.lp1_b[m] + P.lp1_w[3 * m] * d[0] + P.lp1_w[3 * m + 1] * d[1] + P.lp1_w[3 * m + 2] * d[2];
            const float ah = (a - stats[m]) * stats[3 + m];
            const float da = P.bnp_g[m] * stats[3 + m] * (dzp[3 * e + m] - invM * dbp[m] - ah * (invM * dgp[m]));
            acc[3 * m] += da * d[0];
            acc[3 * m + 1] += da * d[1];
            acc[3 * m + 2] += da * d[2];
            acc[9 + m] += da;
            dd[0] += P.lp1_w[3 * m] * da; dd[1] += P.lp1_w[3 * m + 1] * da; dd[2] += P.lp1_w[3 * m + 2] * da;
        }
        if (dp) {  // d = p_j - p_i
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                unsafeAtomicAdd(dp + 3L * j + m, dd[m]);
                unsafeAtomicAdd(dp + 3L * pt + m, -dd[m]);
            }
        }
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const float a = wave_sum_f(acc[i]);
        if (lane == 0) red[wave][i] = a;
    }
    __syncthreads();
    if (threadIdx.x < 12) rec[(long)blockIdx.x * 12 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// fixed-order reduction of float records [R][L] into up to four destination vectors
struct Segs {
    int off[5];      // off[i]..off[i+1] of the record goes to dst[i]
    float *dst[4];
    int count;
};
// (32 slices of 32 outputs: with 8 slices a thread walked R/8 records as one chain of dependent L2 round trips)
constexpr int RED_SL = 32;
__global__ __launch_bounds__(32 * RED_SL) void pt_reduce_kernel(const float *__restrict__ rec, int R, int L, Segs sg) {
    __shared__ double red[RED_SL][32];
    const int ll = threadIdx.x & 31, sl = threadIdx.x >> 5, l = blockIdx.x * 32 + ll;
    double a = 0;
    if (l < L) {
#pragma unroll 4
        for (int r = sl; r < R; r += RED_SL) a += rec[(long)r * L + l];
    }
    red[sl][ll] = a;
    __syncthreads();
    if (sl == 0 && l < L) {
        double t = 0;
#pragma unroll
        for (int i = 0; i < RED_SL; ++i) t += red[i][ll];
        for (int i = 0; i < sg.count; ++i)
            if (l >= sg.off[i] && l < sg.off[i + 1]) sg.dst[i][l - sg.off[i]] = (float)t;
    }
}

template <int C>
int grid_for(int n) {
    const int tiles = (n + Geo<C>::PT - 1) / Geo<C>::PT;
    return tiles < Geo<C>::GMAX ? tiles : Geo<C>::GMAX;
}
inline int edge_grid(long M) {
    const long b = (M + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

struct Work {
    double *rec_d;   // forward records (doubles) -- also the float records of the backward
    float *dz2, *dzp;
    size_t total;
};
template <int C>
size_t rec_bytes(int n) {
    typedef Geo<C> G;
    const size_t g = (size_t)grid_for<C>(n), gp = g;
    size_t fwd = sizeof(double) * 2 * C * g;
    const size_t p0 = sizeof(double) * 6 * 256;
    if (p0 > fwd) fwd = p0;
    size_t b = sizeof(float) * gp * ((size_t)G::CS * C + 2 * C + G::CS);
    const size_t b1 = sizeof(float) * g * ((size_t)G::CS * G::CS + 3 * G::CS), b3 = sizeof(float) * gp * (4 * (size_t)C + 6),
                 b4 = sizeof(float) * 12 * 256;
    if (b1 > b) b = b1;
    if (b3 > b) b = b3;
    if (b4 > b) b = b4;
    return (fwd > b ? fwd : b);
}
template <int C>
Work carve(void *ws, int n, int ns) {
    Work w;
    size_t r = (rec_bytes<C>(n) + 255) & ~(size_t)255;
    unsigned char *base = (unsigned char *)ws;
    w.rec_d = (double *)base;
    w.dz2 = (float *)(base + r);
    const size_t z2 = (sizeof(float) * (size_t)n * ns * Geo<C>::CS + 255) & ~(size_t)255;
    w.dzp = (float *)(base + r + z2);
    w.total = r + z2 + ((sizeof(float) * (size_t)n * ns * 3 + 255) & ~(size_t)255);
    return w;
}

template <int C>
int forward(const float *p, const int32_t *idx, const float *q, const float *k, const float *v, long ld, const Prm &P, int n,
            int ns, int training, float *out, float *stats, float *u1, float *sm, void *ws, hipStream_t st) {
    typedef Geo<C> G;
    const Work w = carve<C>(ws, n, ns);
    const int g = grid_for<C>(n);
    const double M = (double)n * ns;
    const Stats S = split_stats(stats, C);
    if (training) {
        const int ge = edge_grid((long)n * ns);
        hipLaunchKernelGGL(pt_stats_p_kernel, dim3(ge), dim3(256), 0, st, p, idx, P, n, ns, w.rec_d);
        hipLaunchKernelGGL(pt_bn_finalize_kernel, dim3(1), dim3(256), 0, st, w.rec_d, ge, 3, M, P.eps_p, P.mom_p,
                           (float *)S.mp, (float *)S.rp, P.bnp_rm, P.bnp_rv);
        hipLaunchKernelGGL(pt_stats_w_kernel<C>, dim3(g), dim3(G::NT), 0, st, p, idx, q, k, ld, P, stats, n, ns, w.rec_d);
        hipLaunchKernelGGL(pt_bn_finalize_kernel, dim3(C / 4), dim3(256), 0, st, w.rec_d, g, C, M, P.eps_1, P.mom_1,
                           (float *)S.m1, (float *)S.r1, P.bn1_rm, P.bn1_rv);
    }
    hipLaunchKernelGGL(pt_u1_kernel<C>, dim3(g), dim3(G::NT), u1_lds_bytes<C>(), st, p, idx, q, k, ld, P, stats, n, ns, u1,
                       training ? w.rec_d : (double *)nullptr);
    if (training)
        hipLaunchKernelGGL(pt_bn_finalize_kernel, dim3(fsg_cdiv(G::CS, 4)), dim3(256), 0, st, w.rec_d, g, G::CS, M, P.eps_2, P.mom_2,
                           (float *)S.m2, (float *)S.r2, P.bn2_rm, P.bn2_rv);
    hipLaunchKernelGGL(pt_out_kernel<C>, dim3(g), dim3(G::NT), 0, st, p, idx, v, ld, P, stats, n, ns, u1, sm, out);
    return 0;
}

template <int C>
int backward(const float *p, const int32_t *idx, const float *q, const float *k, const float *v, long ld, const Prm &P, int n,
             int ns, int training, const float *g, const float *stats, const float *u1, const float *sm, float *dq, float *dk,
             float *dv, long ldg, float *dp, const fsg_pt_layer_grads &G_, void *ws, hipStream_t st) {
    typedef Geo<C> G;
    constexpr int CS = G::CS;
    const Work w = carve<C>(ws, n, ns);
    const int gr = grid_for<C>(n);
    const float invM = training ? (float)(1.0 / ((double)n * ns)) : 0.f;
    float *rec = (float *)w.rec_d;
    // B1
    hipLaunchKernelGGL(pt_b1_kernel<C>, dim3(gr), dim3(G::NT), 0, st, p, idx, v, ld, P, stats, n, ns, g, u1, sm, dv, ldg, w.dz2,
                       rec);
    {
        const int L = CS * CS + 3 * CS;
        Segs sg = {{0, CS * CS, CS * CS + CS, CS * CS + 2 * CS, L}, {G_.lw2_w, G_.lw2_b, G_.bn2_g, G_.bn2_b}, 4};
        hipLaunchKernelGGL(pt_reduce_kernel, dim3(fsg_cdiv(L, 32)), dim3(32 * RED_SL), 0, st, rec, gr, L, sg);
    }
    // B2
    hipLaunchKernelGGL(pt_b2_kernel<C>, dim3(gr), dim3(G::NT), 0, st, p, idx, q, k, ld, P, stats, n, ns, u1, w.dz2, G_.bn2_g,
                       G_.bn2_b, invM, rec);
    {
        const int L = CS * C + 2 * C + CS;
        Segs sg = {{0, CS * C, CS * C + C, CS * C + 2 * C, L}, {G_.lw1_w, G_.bn1_g, G_.bn1_b, G_.lw1_b}, 4};
        hipLaunchKernelGGL(pt_reduce_kernel, dim3(fsg_cdiv(L, 32)), dim3(32 * RED_SL), 0, st, rec, gr, L, sg);
    }
    // B3
    hipLaunchKernelGGL(pt_b3_kernel<C>, dim3(gr), dim3(G::NT), 0, st, p, idx, q, k, ld, P, stats, n, ns, g, u1, sm, w.dz2,
                       G_.bn2_g, G_.bn2_b, G_.bn1_g, G_.bn1_b, invM, dq, dk, ldg, w.dzp, rec);
    {
        const int L = 4 * C + 6;
        Segs sg = {{0, 3 * C, 4 * C, 4 * C + 3, L}, {G_.lp2_w, G_.lp2_b, G_.bnp_g, G_.bnp_b}, 4};
        hipLaunchKernelGGL(pt_reduce_kernel, dim3(fsg_cdiv(L, 32)), dim3(32 * RED_SL), 0, st, rec, gr, L, sg);
    }
    // B4
    const int ge = edge_grid((long)n * ns);
    hipLaunchKernelGGL(pt_b4_kernel, dim3(ge), dim3(256), 0, st, p, idx, P, stats, n, ns, w.dzp, G_.bnp_g, G_.bnp_b, invM, dp, rec);
    {
        Segs sg = {{0, 9, 12, 12, 12}, {G_.lp1_w, G_.lp1_b, nullptr, nullptr}, 2};
        hipLaunchKernelGGL(pt_reduce_kernel, dim3(1), dim3(32 * RED_SL), 0, st, rec, ge, 12, sg);
    }
    return 0;
}

bool supported_c(int c) { return c == 32 || c == 64 || c == 128 || c == 256 || c == 512; }

}  // namespace

extern "C" size_t fsg_pt_attn_workspace_bytes(int n, int ns, int c) {
    if (n <= 0 || ns <= 0 || ns > MAXNS || !supported_c(c)) return 0;
    switch (c) {
        case 32: return carve<32>(nullptr, n, ns).total;
        case 64: return carve<64>(nullptr, n, ns).total;
        case 128: return carve<128>(nullptr, n, ns).total;
        case 256: return carve<256>(nullptr, n, ns).total;
        default: return carve<512>(nullptr, n, ns).total;
    }
}

extern "C" int fsg_pt_attn_fwd_f32(const float *p, const int32_t *idx, const float *q, const float *k, const float *v,
                                   int64_t ld, const fsg_pt_layer_params *params, int n, int ns, int c, int training,
                                   float *out, float *stats, float *u1, float *sm, void *workspace, fsg_stream_t stream) {
    FSG_REQUIRE(p && idx && q && k && v && params && out && stats && u1 && sm && workspace, "fsg_pt_attn_fwd_f32: NULL pointer");
    FSG_REQUIRE(n > 0 && ns > 0 && ns <= MAXNS && supported_c(c) && ld >= c,
                "fsg_pt_attn_fwd_f32: bad shape n=%d ns=%d c=%d ld=%ld (c in {32,64,128,256,512}, ns <= 16)", n, ns, c, (long)ld);
    FSG_REQUIRE((long)n * ns < (1L << 31), "fsg_pt_attn_fwd_f32: n*ns too large");
    hipStream_t st = (hipStream_t)stream;
    switch (c) {
        case 32: forward<32>(p, idx, q, k, v, ld, *params, n, ns, training, out, stats, u1, sm, workspace, st); break;
        case 64: forward<64>(p, idx, q, k, v, ld, *params, n, ns, training, out, stats, u1, sm, workspace, st); break;
        case 128: forward<128>(p, idx, q, k, v, ld, *params, n, ns, training, out, stats, u1, sm, workspace, st); break;
        case 256: forward<256>(p, idx, q, k, v, ld, *params, n, ns, training, out, stats, u1, sm, workspace, st); break;
        default: forward<512>(p, idx, q, k, v, ld, *params, n, ns, training, out, stats, u1, sm, workspace, st); break;
    }
    FSG_CHECK_LAUNCH("fsg_pt_attn_fwd_f32");
    return FSG_OK;
}

extern "C" int fsg_pt_attn_bwd_f32(const float *p, const int32_t *idx, const float *q, const float *k, const float *v,
                                   int64_t ld, const fsg_pt_layer_params *params, int n, int ns, int c, int training,
                                   const float *grad_out, const float *stats, const float *u1, const float *sm,
                                   float *grad_q, float *grad_k, float *grad_v, int64_t ldg, float *grad_p,
                                   const fsg_pt_layer_grads *grads, void *workspace, fsg_stream_t stream) {
    FSG_REQUIRE(p && idx && q && k && v && params && grad_out && stats && u1 && sm && grad_q && grad_k && grad_v && grads &&
                    workspace, "fsg_pt_attn_bwd_f32: NULL pointer");
    FSG_REQUIRE(n > 0 && ns > 0 && ns <= MAXNS && supported_c(c) && ld >= c && ldg >= c,
                "fsg_pt_attn_bwd_f32: bad shape n=%d ns=%d c=%d", n, ns, c);
    const fsg_pt_layer_grads &G_ = *grads;
    FSG_REQUIRE(G_.lp1_w && G_.lp1_b && G_.bnp_g && G_.bnp_b && G_.lp2_w && G_.lp2_b && G_.bn1_g && G_.bn1_b && G_.lw1_w &&
                    G_.lw1_b && G_.bn2_g && G_.bn2_b && G_.lw2_w && G_.lw2_b, "fsg_pt_attn_bwd_f32: NULL gradient pointer");
    hipStream_t st = (hipStream_t)stream;
    switch (c) {
        case 32: backward<32>(p, idx, q, k, v, ld, *params, n, ns, training, grad_out, stats, u1, sm, grad_q, grad_k, grad_v, ldg, grad_p, G_, workspace, st); break;
        case 64: backward<64>(p, idx, q, k, v, ld, *params, n, ns, training, grad_out, stats, u1, sm, grad_q, grad_k, grad_v, ldg, grad_p, G_, workspace, st); break;
        case 128: backward<128>(p, idx, q, k, v, ld, *params, n, ns, training, grad_out, stats, u1, sm, grad_q, grad_k, grad_v, ldg, grad_p, G_, workspace, st); break;
        case 256: backward<256>(p, idx, q, k, v, ld, *params, n, ns, training, grad_out, stats, u1, sm, grad_q, grad_k, grad_v, ldg, grad_p, G_, workspace, st); break;
        default: backward<512>(p, idx, q, k, v, ld, *params, n, ns, training, grad_out, stats, u1, sm, grad_q, grad_k, grad_v, ldg, grad_p, G_, workspace, st); break;
    }
    FSG_CHECK_LAUNCH("fsg_pt_attn_bwd_f32");
    return FSG_OK;
}

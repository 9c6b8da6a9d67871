// BatchNorm1d (+ residual) (+ ReLU) over packed point rows -- include/fsg_hip.h: fsg_bn_rows_{fwd,bwd}_f32.
// The glue between the Linears of models/pointtransformer/seg_model.py: `relu(bn(linear(x)))` (:66,:82,:138-139,
// :92-99,:168), `relu(bn3(linear3(y)) + identity)` (:140-141).  ATen runs each as collect-statistics / transform /
// running-stat update / clamp (and add) = 4-5 kernels forward, threshold / reduce / elementwise (and a copy for the
// residual branch) backward; here: statistics -> finalize -> apply, and reduce -> fold -> apply.
//   x (M,C) row-major, C in {32,64,128,256,512}; thread (row slot, channel): every access is a contiguous 4*C-byte row.
//   Train-mode statistics: per-thread fp64 sum / sum of squares, one record per workgroup, one wave per channel folds
//   them in a fixed order (reproducible; no E[x^2]-E[x]^2 cancellation at fp32 scale).
#include <stdlib.h>

#include "fsg_common.h"

namespace {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

constexpr int GMAX = 256;

__global__ void bnr_stats_kernel(const float *__restrict__ x, long M, int C, double *__restrict__ rec) {
    extern __shared__ double red[];   // [2][blockDim]
    const int NT = blockDim.x, PT = NT / C, tid = threadIdx.x, ps = tid / C, ch = tid % C;
    double s = 0, ss = 0;
#pragma unroll 4   // several rows in flight per thread (the loop is one dependent round trip per row otherwise)
    for (long r = (long)blockIdx.x * PT + ps; r < M; r += (long)gridDim.x * PT) {
        const float v = x[r * C + ch];
        s += v;
        ss += (double)v * v;
    }
    red[tid] = s;
    red[NT + tid] = ss;
    __syncthreads();
    if (tid < C) {
        double a = 0, b = 0;
        for (int p = 0; p < PT; ++p) { a += red[p * C + tid]; b += red[NT + p * C + tid]; }
        rec[(long)blockIdx.x * 2 * C + tid] = a;
        rec[(long)blockIdx.x * 2 * C + C + tid] = b;
    }
}

__global__ __launch_bounds__(256) void bnr_finalize_kernel(const double *__restrict__ rec, int R, int C, double M, float eps,
                                                           float mom, float *__restrict__ mean, float *__restrict__ rstd,
                                                           float *__restrict__ rm, float *__restrict__ rv) {
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (l >= C) return;
    double S = 0, SS = 0;
    for (int r = lane; r < R; r += 64) {
        S += rec[(long)r * 2 * C + l];
        SS += rec[(long)r * 2 * C + C + l];
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

__global__ __launch_bounds__(256) void bnr_apply_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        const float *__restrict__ mean, const float *__restrict__ rstd,
                                                        long total4, int C, int relu, float *__restrict__ out) {
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total4; t += (long)gridDim.x * 256) {
        const int c0 = (int)((t * 4) % C);
        const float4 v = reinterpret_cast<const float4 *>(x)[t];
        float4 o;
        o.x = gamma[c0] * ((v.x - mean[c0]) * rstd[c0]) + beta[c0];
        o.y = gamma[c0 + 1] * ((v.y - mean[c0 + 1]) * rstd[c0 + 1]) + beta[c0 + 1];
        o.z = gamma[c0 + 2] * ((v.z - mean[c0 + 2]) * rstd[c0 + 2]) + beta[c0 + 2];
        o.w = gamma[c0 + 3] * ((v.w - mean[c0 + 3]) * rstd[c0 + 3]) + beta[c0 + 3];
        if (res) {
            const float4 r = reinterpret_cast<const float4 *>(res)[t];
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        reinterpret_cast<float4 *>(out)[t] = o;
    }
}

// backward, pass 1: d_beta = sum g', d_gamma = sum g' xhat with g' = g [out > 0] (ReLU) -- float records per workgroup
__global__ void bnr_bwd_reduce_kernel(const float *__restrict__ g, const float *__restrict__ x, const float *__restrict__ out,
                                      const float *__restrict__ mean, const float *__restrict__ rstd, long M, int C,
                                      int relu, float *__restrict__ rec) {
    extern __shared__ double red[];
    const int NT = blockDim.x, PT = NT / C, tid = threadIdx.x, ps = tid / C, ch = tid % C;
    const float m = mean[ch], rs = rstd[ch];
    double db = 0, dg = 0;
#pragma unroll 4   // several rows in flight per thread (the loop is one dependent round trip per row otherwise)
    for (long r = (long)blockIdx.x * PT + ps; r < M; r += (long)gridDim.x * PT) {
        float gv = g[r * C + ch];
        if (relu && !(out[r * C + ch] > 0.f)) gv = 0.f;
        db += gv;
        dg += (double)gv * ((x[r * C + ch] - m) * rs);
    }
    red[tid] = dg;
    red[NT + tid] = db;
    __syncthreads();
    if (tid < C) {
        double a = 0, b = 0;
        for (int p = 0; p < PT; ++p) { a += red[p * C + tid]; b += red[NT + p * C + tid]; }
        reinterpret_cast<double *>(rec)[(long)blockIdx.x * 2 * C + tid] = a;       // d_gamma
        reinterpret_cast<double *>(rec)[(long)blockIdx.x * 2 * C + C + tid] = b;   // d_beta
    }
}

__global__ __launch_bounds__(256) void bnr_fold_kernel(const double *__restrict__ rec, int R, int C,
                                                       float *__restrict__ dgamma, float *__restrict__ dbeta) {
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (l >= 2 * C) return;
    double S = 0;
    for (int r = lane; r < R; r += 64) S += rec[(long)r * 2 * C + l];
    S = wave_sum_d(S);
    if (lane == 0) (l < C ? dgamma[l] : dbeta[l - C]) = (float)S;
}

__global__ __launch_bounds__(256) void bnr_bwd_apply_kernel(const float *__restrict__ g, const float *__restrict__ x,
                                                            const float *__restrict__ out, const float *__restrict__ gamma,
                                                            const float *__restrict__ mean, const float *__restrict__ rstd,
                                                            const float *__restrict__ dgamma, const float *__restrict__ dbeta,
                                                            long total4, int C, int relu, float invM,
                                                            float *__restrict__ gx, float *__restrict__ gres) {
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total4; t += (long)gridDim.x * 256) {
        const int c0 = (int)((t * 4) % C);
        float4 gv = reinterpret_cast<const float4 *>(g)[t];
        const float4 xv = reinterpret_cast<const float4 *>(x)[t];
        if (relu) {
            const float4 ov = reinterpret_cast<const float4 *>(out)[t];
            gv.x = ov.x > 0.f ? gv.x : 0.f; gv.y = ov.y > 0.f ? gv.y : 0.f;
            gv.z = ov.z > 0.f ? gv.z : 0.f; gv.w = ov.w > 0.f ? gv.w : 0.f;
        }
        if (gres) reinterpret_cast<float4 *>(gres)[t] = gv;
        float4 o;
        const float *gp = &gv.x, *xp = &xv.x;
        float *op = &o.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c0 + e;
            const float xh = (xp[e] - mean[c]) * rstd[c];
            op[e] = gamma[c] * rstd[c] * (gp[e] - invM * dbeta[c] - xh * (invM * dgamma[c]));
        }
        reinterpret_cast<float4 *>(gx)[t] = o;
    }
}

// ---- one launch for the whole layer when the tensor is small (M <= SMALL_M rows: the PointTransformer stages of 1024 points
// and fewer): a workgroup owns four channels, each thread the 16-byte pieces of up to four rows, which stay in registers
// between the statistics pass and the apply pass.  These kernels are chains of dependent memory round trips (~1.5-2 us each,
// the data was written by the previous kernel): every parameter is requested together with the rows, so a layer costs ONE
// round trip + one launch instead of three of each.  (Above 1024 rows a four-channel slice uses an eighth of every cache line
// it pulls and the L2 -> CU path becomes the bound: 4096 x 64 measured 13-22 us against 3 x 4.6 us for the three launches.)
constexpr int SMALL_M = 1024, SMALL_NT = 256, SMALL_RPT = SMALL_M / SMALL_NT;

__device__ __forceinline__ void block_sum8(double (&v)[8], double (*red)[8], int tid) {   // sums over the 512 threads, every thread gets them
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off, 64);
    }
    const int wave = tid >> 6;
    if ((tid & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) red[wave][q] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        double t = 0;
#pragma unroll
        for (int w = 0; w < SMALL_NT / 64; ++w) t += red[w][q];      // wave order: fixed, reproducible
        v[q] = t;
    }
}

__global__ __launch_bounds__(SMALL_NT) void bnr_small_fwd_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                                  const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                  int M, int C, float eps, float mom, int relu,
                                                                  float *__restrict__ out, float *__restrict__ mean,
                                                                  float *__restrict__ rstd, float *__restrict__ rm,
                                                                  float *__restrict__ rv) {
    __shared__ double red[SMALL_NT / 64][8];
    const int tid = threadIdx.x, c0 = blockIdx.x * 4;
    float4 v[SMALL_RPT], rr[SMALL_RPT];
    const float4 ga4 = *reinterpret_cast<const float4 *>(gamma + c0), be4 = *reinterpret_cast<const float4 *>(beta + c0);
    float rm0 = 0.f, rv0 = 0.f;
    if (tid < 4) {
        if (rm) rm0 = rm[c0 + tid];
        if (rv) rv0 = rv[c0 + tid];
    }
#pragma unroll
    for (int u = 0; u < SMALL_RPT; ++u) {
        const int r = tid + SMALL_NT * u;
        v[u] = r < M ? *reinterpret_cast<const float4 *>(x + (long)r * C + c0) : float4{0.f, 0.f, 0.f, 0.f};
        rr[u] = (res && r < M) ? *reinterpret_cast<const float4 *>(res + (long)r * C + c0) : float4{0.f, 0.f, 0.f, 0.f};
    }
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // sums and sums of squares of the four channels (rows behind M add zeros)
#pragma unroll
    for (int u = 0; u < SMALL_RPT; ++u) {
        s[0] += v[u].x; s[1] += v[u].y; s[2] += v[u].z; s[3] += v[u].w;
        s[4] += (double)v[u].x * v[u].x; s[5] += (double)v[u].y * v[u].y;
        s[6] += (double)v[u].z * v[u].z; s[7] += (double)v[u].w * v[u].w;
    }
    block_sum8(s, red, tid);
    float mu[4], rs[4], ga[4], be[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double m = s[q] / (double)M;
        double var = s[4 + q] / (double)M - m * m;
        if (var < 0) var = 0;
        mu[q] = (float)m;
        rs[q] = (float)(1.0 / sqrt(var + (double)eps));
        ga[q] = (&ga4.x)[q];
        be[q] = (&be4.x)[q];
        if (tid == q) {
            mean[c0 + q] = mu[q];
            rstd[c0 + q] = rs[q];
            if (rm) rm[c0 + q] = (float)((1.0 - mom) * rm0 + mom * m);
            if (rv) rv[c0 + q] = (float)((1.0 - mom) * rv0 + mom * (M > 1 ? var * M / (M - 1.0) : var));
        }
    }
#pragma unroll
    for (int u = 0; u < SMALL_RPT; ++u) {
        const int r = tid + SMALL_NT * u;
        if (r < M) {
            float4 o;
            o.x = ga[0] * ((v[u].x - mu[0]) * rs[0]) + be[0];
            o.y = ga[1] * ((v[u].y - mu[1]) * rs[1]) + be[1];
            o.z = ga[2] * ((v[u].z - mu[2]) * rs[2]) + be[2];
            o.w = ga[3] * ((v[u].w - mu[3]) * rs[3]) + be[3];
            o.x += rr[u].x; o.y += rr[u].y; o.z += rr[u].z; o.w += rr[u].w;      // zeros without a residual
            if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            *reinterpret_cast<float4 *>(out + (long)r * C + c0) = o;
        }
    }
}

__global__ __launch_bounds__(SMALL_NT) void bnr_small_bwd_kernel(const float *__restrict__ g, const float *__restrict__ x,
                                                                  const float *__restrict__ out, const float *__restrict__ gamma,
                                                                  const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                  int M, int C, int relu, float invM, float *__restrict__ gx,
                                                                  float *__restrict__ gres, float *__restrict__ dgamma,
                                                                  float *__restrict__ dbeta) {
    __shared__ double red[SMALL_NT / 64][8];
    const int tid = threadIdx.x, c0 = blockIdx.x * 4;
    const float4 mu = *reinterpret_cast<const float4 *>(mean + c0), rs = *reinterpret_cast<const float4 *>(rstd + c0);
    const float4 ga4 = *reinterpret_cast<const float4 *>(gamma + c0);
    float4 gv[SMALL_RPT], xh[SMALL_RPT];
#pragma unroll
    for (int u = 0; u < SMALL_RPT; ++u) {
        const int r = tid + SMALL_NT * u;
        const bool ok = r < M;
        const long o = (long)(ok ? r : 0) * C + c0;
        float4 a = *reinterpret_cast<const float4 *>(g + o);
        const float4 xv = *reinterpret_cast<const float4 *>(x + o);
        if (relu) {
            const float4 ov = *reinterpret_cast<const float4 *>(out + o);
            a.x = ov.x > 0.f ? a.x : 0.f; a.y = ov.y > 0.f ? a.y : 0.f;
            a.z = ov.z > 0.f ? a.z : 0.f; a.w = ov.w > 0.f ? a.w : 0.f;
        }
        if (!ok) a = float4{0.f, 0.f, 0.f, 0.f};
        gv[u] = a;
        xh[u] = xv;
    }
#pragma unroll
    for (int u = 0; u < SMALL_RPT; ++u)
        xh[u] = float4{(xh[u].x - mu.x) * rs.x, (xh[u].y - mu.y) * rs.y, (xh[u].z - mu.z) * rs.z, (xh[u].w - mu.w) * rs.w};
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // d_gamma (4), d_beta (4)
#pragma unroll
    for (int u = 0; u < SMALL_RPT; ++u) {
        s[0] += (double)gv[u].x * xh[u].x; s[1] += (double)gv[u].y * xh[u].y;
        s[2] += (double)gv[u].z * xh[u].z; s[3] += (double)gv[u].w * xh[u].w;
        s[4] += gv[u].x; s[5] += gv[u].y; s[6] += gv[u].z; s[7] += gv[u].w;
    }
    block_sum8(s, red, tid);
    float dg[4], db[4], co[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        dg[q] = (float)s[q];
        db[q] = (float)s[4 + q];
        co[q] = (&ga4.x)[q] * (&rs.x)[q];
        if (tid == q) {
            dgamma[c0 + q] = dg[q];
            dbeta[c0 + q] = db[q];
        }
    }
#pragma unroll
    for (int u = 0; u < SMALL_RPT; ++u) {
        const int r = tid + SMALL_NT * u;
        if (r < M) {
            const long o = (long)r * C + c0;
            if (gres) *reinterpret_cast<float4 *>(gres + o) = gv[u];
            float4 w;
            w.x = co[0] * (gv[u].x - invM * db[0] - xh[u].x * (invM * dg[0]));
            w.y = co[1] * (gv[u].y - invM * db[1] - xh[u].y * (invM * dg[1]));
            w.z = co[2] * (gv[u].z - invM * db[2] - xh[u].z * (invM * dg[2]));
            w.w = co[3] * (gv[u].w - invM * db[3] - xh[u].w * (invM * dg[3]));
            *reinterpret_cast<float4 *>(gx + o) = w;
        }
    }
}

inline bool ok_c(int C) { return C == 32 || C == 64 || C == 128 || C == 256 || C == 512; }
inline int nt_for(int C) { return C < 256 ? 256 : C; }
inline int grid_rows(long M, int C) {
    const long pt = nt_for(C) / C, g = (M + pt - 1) / pt;
    return (int)(g < 1 ? 1 : (g > GMAX ? GMAX : g));
}
inline int grid_elems(long total4) {
    const long g = (total4 + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

inline bool bnr_three_launches() {     // FSG_BNR_THREE_LAUNCHES=1: statistics / finalize / apply as separate launches at every size
    static int v = -1;
    if (v < 0) v = getenv("FSG_BNR_THREE_LAUNCHES") ? 1 : 0;
    return v == 1;
}

}  // namespace

extern "C" size_t fsg_bn_rows_workspace_bytes(long M, int C) {
    (void)M;
    return ok_c(C) ? sizeof(double) * 2 * (size_t)C * GMAX : 0;
}

extern "C" int fsg_bn_rows_fwd_f32(const float *x, const float *residual, const float *gamma, const float *beta,
                                   float *running_mean, float *running_var, long M, int C, int training, float momentum,
                                   float eps, int relu, float *out, float *mean, float *rstd, void *workspace,
                                   fsg_stream_t stream) {
    FSG_REQUIRE(x && gamma && beta && out && mean && rstd, "fsg_bn_rows_fwd_f32: NULL pointer");
    FSG_REQUIRE(M > 0 && ok_c(C), "fsg_bn_rows_fwd_f32: bad shape M=%ld C=%d (C in {32,64,128,256,512})", M, C);
    FSG_REQUIRE(!training || workspace, "fsg_bn_rows_fwd_f32: training needs the workspace");
    hipStream_t st = (hipStream_t)stream;
    if (training && M <= SMALL_M && !bnr_three_launches()) {
        hipLaunchKernelGGL(bnr_small_fwd_kernel, dim3(C / 4), dim3(SMALL_NT), 0, st, x, residual, gamma, beta, (int)M, C, eps,
                           momentum, relu, out, mean, rstd, running_mean, running_var);
        FSG_CHECK_LAUNCH("fsg_bn_rows_fwd_f32/small");
        return FSG_OK;
    }
    if (training) {
        const int G = grid_rows(M, C), NT = nt_for(C);
        hipLaunchKernelGGL(bnr_stats_kernel, dim3(G), dim3(NT), sizeof(double) * 2 * NT, st, x, M, C, (double *)workspace);
        FSG_CHECK_LAUNCH("fsg_bn_rows_fwd_f32/stats");
        hipLaunchKernelGGL(bnr_finalize_kernel, dim3(fsg_cdiv(C, 4)), dim3(256), 0, st, (const double *)workspace, G, C,
                           (double)M, eps, momentum, mean, rstd, running_mean, running_var);
        FSG_CHECK_LAUNCH("fsg_bn_rows_fwd_f32/finalize");
    }
    const long total4 = M * C / 4;
    hipLaunchKernelGGL(bnr_apply_kernel, dim3(grid_elems(total4)), dim3(256), 0, st, x, residual, gamma, beta, mean, rstd,
                       total4, C, relu, out);
    FSG_CHECK_LAUNCH("fsg_bn_rows_fwd_f32/apply");
    return FSG_OK;
}

extern "C" int fsg_bn_rows_bwd_f32(const float *grad_out, const float *x, const float *out, const float *gamma,
                                   const float *mean, const float *rstd, long M, int C, int training, int relu,
                                   float *grad_x, float *grad_residual, float *grad_gamma, float *grad_beta,
                                   void *workspace, fsg_stream_t stream) {
    FSG_REQUIRE(grad_out && x && gamma && mean && rstd && grad_x && grad_gamma && grad_beta && workspace,
                "fsg_bn_rows_bwd_f32: NULL pointer");
    FSG_REQUIRE(!relu || out, "fsg_bn_rows_bwd_f32: the ReLU mask needs the forward output");
    FSG_REQUIRE(M > 0 && ok_c(C), "fsg_bn_rows_bwd_f32: bad shape M=%ld C=%d", M, C);
    hipStream_t st = (hipStream_t)stream;
    if (M <= SMALL_M && !bnr_three_launches()) {
        hipLaunchKernelGGL(bnr_small_bwd_kernel, dim3(C / 4), dim3(SMALL_NT), 0, st, grad_out, x, out, gamma, mean, rstd, (int)M, C,
                           relu, training ? (float)(1.0 / (double)M) : 0.f, grad_x, grad_residual, grad_gamma, grad_beta);
        FSG_CHECK_LAUNCH("fsg_bn_rows_bwd_f32/small");
        return FSG_OK;
    }
    const int G = grid_rows(M, C), NT = nt_for(C);
    hipLaunchKernelGGL(bnr_bwd_reduce_kernel, dim3(G), dim3(NT), sizeof(double) * 2 * NT, st, grad_out, x, out, mean, rstd, M,
                       C, relu, (float *)workspace);
    FSG_CHECK_LAUNCH("fsg_bn_rows_bwd_f32/reduce");
    hipLaunchKernelGGL(bnr_fold_kernel, dim3(fsg_cdiv(2 * C, 4)), dim3(256), 0, st, (const double *)workspace, G, C,
                       grad_gamma, grad_beta);
    FSG_CHECK_LAUNCH("fsg_bn_rows_bwd_f32/fold");
    const long total4 = M * C / 4;
    hipLaunchKernelGGL(bnr_bwd_apply_kernel, dim3(grid_elems(total4)), dim3(256), 0, st, grad_out, x, out, gamma, mean, rstd,
                       grad_gamma, grad_beta, total4, C, relu, training ? (float)(1.0 / (double)M) : 0.f, grad_x,
                       grad_residual);
    FSG_CHECK_LAUNCH("fsg_bn_rows_bwd_f32/apply");
    return FSG_OK;
}

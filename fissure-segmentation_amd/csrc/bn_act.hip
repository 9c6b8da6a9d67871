// Fused train-mode BatchNorm + LeakyReLU on point-major rows (M, C) -- the normalisation/activation stage behind every
// 1x1 conv of the point-wise head (models/dgcnn.py:282-323 `ConvBlock`: conv -> BatchNorm -> LeakyReLU).
// include/fsg_hip.h: fsg_bn_act_{fwd,bwd}_f32.   HBM-bound: forward = 2 reads + 1 write of the (M,C) block,
// backward = 4 reads + 1 write; lanes run along channels (C % 64 == 0) so every access is a 256-byte row segment.
// Statistics: per-workgroup shifted sums -> (n, mean, M2) records merged with Chan's formula in fp64
// (bn_merge_finalize_kernel of edgeconv.hip), i.e. no E[x^2]-E[x]^2 cancellation (MIOpen's BN loses ~1e-2 there).
#include "fsg_common.h"

int fsg_ec_finalize_launch(const float *partials, int R, int Co, float eps, float momentum, float *mean, float *invstd,
                           float *running_mean, float *running_var, hipStream_t st);                 // edgeconv.hip
size_t fsg_ec_finalize_stage_floats(int Co);
int fsg_ec_sum_launch(const float *partials, int R, int L, int nvec, float *out0, float *out1, hipStream_t st);

namespace {

constexpr int ROWS = 128;  // rows per workgroup in the reduction kernels

__device__ __forceinline__ float lrelu(float u, float slope) { return u > 0.f ? u : u * slope; }

__global__ __launch_bounds__(256) void bnact_stats_kernel(const float *__restrict__ y, long M, int C,
                                                           float *__restrict__ partials) {
    __shared__ float red[3][4][64];
    const int cg = blockIdx.x, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = cg * 64 + lane;
    const long r0 = (long)blockIdx.y * ROWS;
    const long r1 = r0 + ROWS < M ? r0 + ROWS : M;
    float shift = 0.f, s1 = 0.f, s2 = 0.f, cnt = 0.f;
    if (r0 + wave < r1) shift = y[(r0 + wave) * C + c];
#pragma unroll 4
    for (long r = r0 + wave; r < r1; r += 4) {
        const float d = y[r * C + c] - shift;
        s1 += d;
        s2 = __builtin_fmaf(d, d, s2);
        cnt += 1.f;
    }
    float mean = 0.f, m2 = 0.f;
    if (cnt > 0.f) {
        mean = shift + s1 / cnt;
        m2 = fmaxf(s2 - s1 * s1 / cnt, 0.f);
    }
    red[0][wave][lane] = cnt;
    red[1][wave][lane] = mean;
    red[2][wave][lane] = m2;
    __syncthreads();
    if (wave == 0) {
        float n = red[0][0][lane], mu = red[1][0][lane], M2 = red[2][0][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float nb = red[0][w][lane];
            if (nb > 0.f) {
                const float tot = n + nb, delta = red[1][w][lane] - mu;
                mu += delta * (nb / tot);
                M2 += red[2][w][lane] + delta * delta * (n * nb / tot);
                n = tot;
            }
        }
        float *pr = partials + (long)blockIdx.y * 3 * C;
        pr[c] = n;
        pr[C + c] = mu;
        pr[2 * C + c] = M2;
    }
}

__global__ __launch_bounds__(256) void bnact_apply_kernel(const float *__restrict__ y, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, const float *__restrict__ mean,
                                                           const float *__restrict__ invstd, long total4, int C,
                                                           float slope, float *__restrict__ out) {
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total4; t += (long)gridDim.x * 256) {
        const int c = (int)((t * 4) % C);
        const float4 v = reinterpret_cast<const float4 *>(y)[t];
        const float4 g = *reinterpret_cast<const float4 *>(gamma + c), be = *reinterpret_cast<const float4 *>(beta + c);
        const float4 mu = *reinterpret_cast<const float4 *>(mean + c), r = *reinterpret_cast<const float4 *>(invstd + c);
        float4 o;
        o.x = lrelu(__builtin_fmaf(v.x, g.x * r.x, be.x - mu.x * g.x * r.x), slope);
        o.y = lrelu(__builtin_fmaf(v.y, g.y * r.y, be.y - mu.y * g.y * r.y), slope);
        o.z = lrelu(__builtin_fmaf(v.z, g.z * r.z, be.z - mu.z * g.z * r.z), slope);
        o.w = lrelu(__builtin_fmaf(v.w, g.w * r.w, be.w - mu.w * g.w * r.w), slope);
        reinterpret_cast<float4 *>(out)[t] = o;
    }
}

// partial sums of h = g f'(u) and h*yhat per channel
__global__ __launch_bounds__(256) void bnact_bwd_reduce_kernel(const float *__restrict__ gout, const float *__restrict__ y,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ beta,
                                                                const float *__restrict__ mean,
                                                                const float *__restrict__ invstd, long M, int C,
                                                                float slope, float *__restrict__ partials) {
    __shared__ float red[2][4][64];
    const int cg = blockIdx.x, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = cg * 64 + lane;
    const long r0 = (long)blockIdx.y * ROWS;
    const long r1 = r0 + ROWS < M ? r0 + ROWS : M;
    const float r = invstd[c], mu = mean[c], a = gamma[c] * r, b = beta[c] - mu * a;
    float sb = 0.f, sg = 0.f;
#pragma unroll 4
    for (long rr = r0 + wave; rr < r1; rr += 4) {
        const float yv = y[rr * C + c];
        const float u = __builtin_fmaf(yv, a, b);
        const float h = gout[rr * C + c] * (u > 0.f ? 1.f : slope);
        sb += h;
        sg = __builtin_fmaf(h, (yv - mu) * r, sg);
    }
    red[0][wave][lane] = sb;
    red[1][wave][lane] = sg;
    __syncthreads();
    if (wave == 0) {
        float *pr = partials + (long)blockIdx.y * 2 * C;
        pr[c] = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
        pr[C + c] = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
    }
}

__global__ __launch_bounds__(256) void bnact_bwd_apply_kernel(const float *__restrict__ gout, const float *__restrict__ y,
                                                               const float *__restrict__ gamma,
                                                               const float *__restrict__ beta,
                                                               const float *__restrict__ mean,
                                                               const float *__restrict__ invstd,
                                                               const float *__restrict__ dbeta,
                                                               const float *__restrict__ dgamma, long total, int C,
                                                               int training, float invM, float slope,
                                                               float *__restrict__ gy) {
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int c = (int)(t % C);
        const float r = invstd[c], mu = mean[c], a = gamma[c] * r, b = beta[c] - mu * a;
        const float yv = y[t];
        const float u = __builtin_fmaf(yv, a, b);
        float h = gout[t] * (u > 0.f ? 1.f : slope);
        if (training) h -= dbeta[c] * invM + (yv - mu) * r * dgamma[c] * invM;
        gy[t] = a * h;
    }
}

// ---- BatchNorm + LeakyReLU + max over the points of each cloud (the global feature, models/dgcnn.py:134-137,156):
// f(BN(.)) is monotone per channel, so only the per-cloud max (gamma >= 0) or min (gamma < 0) of the PRE-norm rows
// is needed; the (B*N, C) activation is never written.
__global__ __launch_bounds__(256) void bnmax_stats_kernel(const float *__restrict__ y, const float *__restrict__ gamma,
                                                           int N, int C, int training, float *__restrict__ partials,
                                                           float *__restrict__ sel_val, int *__restrict__ sel_arg) {
    __shared__ float red[3][4][64];
    __shared__ float bestv[4][64];
    __shared__ int besta[4][64];
    const int cg = blockIdx.x, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = cg * 64 + lane;
    const int tiles = (N + ROWS - 1) / ROWS;
    const int b = blockIdx.y / tiles, tile = blockIdx.y - b * tiles;
    const int n0 = tile * ROWS, n1 = min(N, n0 + ROWS);
    const float *yb = y + (long)b * N * C;
    const float sgn = gamma[c] >= 0.f ? 1.f : -1.f;
    float shift = 0.f, s1 = 0.f, s2 = 0.f, cnt = 0.f, best = -INFINITY;
    int barg = 0;
    if (n0 + wave < n1) shift = yb[(long)(n0 + wave) * C + c];
#pragma unroll 4
    for (int n = n0 + wave; n < n1; n += 4) {
        const float v = yb[(long)n * C + c];
        const float d = v - shift;
        s1 += d;
        s2 = __builtin_fmaf(d, d, s2);
        cnt += 1.f;
        if (sgn * v > best) { best = sgn * v; barg = n; }
    }
    float mean = 0.f, m2 = 0.f;
    if (cnt > 0.f) {
        mean = shift + s1 / cnt;
        m2 = fmaxf(s2 - s1 * s1 / cnt, 0.f);
    }
    red[0][wave][lane] = cnt;
    red[1][wave][lane] = mean;
    red[2][wave][lane] = m2;
    bestv[wave][lane] = best;
    besta[wave][lane] = barg;
    __syncthreads();
    if (wave == 0) {
        float n = red[0][0][lane], mu = red[1][0][lane], M2 = red[2][0][lane];
        float bv = bestv[0][lane];
        int ba = besta[0][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float nb = red[0][w][lane];
            if (nb > 0.f) {
                const float tot = n + nb, delta = red[1][w][lane] - mu;
                mu += delta * (nb / tot);
                M2 += red[2][w][lane] + delta * delta * (n * nb / tot);
                n = tot;
            }
            if (bestv[w][lane] > bv || (bestv[w][lane] == bv && besta[w][lane] < ba)) { bv = bestv[w][lane]; ba = besta[w][lane]; }
        }
        if (training) {
            float *pr = partials + (long)blockIdx.y * 3 * C;
            pr[c] = n;
            pr[C + c] = mu;
            pr[2 * C + c] = M2;
        }
        sel_val[(long)blockIdx.y * C + c] = bv;   // signed: sgn * y
        sel_arg[(long)blockIdx.y * C + c] = ba;
    }
}

__global__ __launch_bounds__(256) void bnmax_finish_kernel(const float *__restrict__ sel_val, const int *__restrict__ sel_arg,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            const float *__restrict__ mean, const float *__restrict__ invstd,
                                                            int tiles, int C, float slope, float *__restrict__ out,
                                                            float *__restrict__ ysel, int *__restrict__ arg) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float bv = -INFINITY;
    int ba = 0;
    for (int t = 0; t < tiles; ++t) {
        const float v = sel_val[((long)b * tiles + t) * C + c];
        const int a = sel_arg[((long)b * tiles + t) * C + c];
        if (v > bv || (v == bv && a < ba)) { bv = v; ba = a; }
    }
    const float sgn = gamma[c] >= 0.f ? 1.f : -1.f;
    const float yv = sgn * bv;
    const float a1 = gamma[c] * invstd[c];
    ysel[(long)b * C + c] = yv;
    arg[(long)b * C + c] = ba;
    out[(long)b * C + c] = lrelu(__builtin_fmaf(yv, a1, beta[c] - mean[c] * a1), slope);
}

// grad_out (B,C) -> dy (B,N,C) = a ( h [n = arg] - dbeta/M - yhat dgamma/M ); dbeta/dgamma are sums over the B clouds
__global__ __launch_bounds__(256) void bnmax_bwd_kernel(const float *__restrict__ gout, const float *__restrict__ y,
                                                         const float *__restrict__ ysel, const int *__restrict__ arg,
                                                         const float *__restrict__ gamma, const float *__restrict__ beta,
                                                         const float *__restrict__ mean, const float *__restrict__ invstd,
                                                         int B, int N, int C, int training, float slope,
                                                         float *__restrict__ gy, float *__restrict__ dgamma,
                                                         float *__restrict__ dbeta) {
    const int cg = blockIdx.x, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = cg * 64 + lane;
    const int tiles = (N + ROWS - 1) / ROWS;
    const int b = blockIdx.y / tiles, tile = blockIdx.y - b * tiles;
    const float r = invstd[c], mu = mean[c], a = gamma[c] * r, bb = beta[c] - mu * a;
    float sb = 0.f, sg = 0.f, hb = 0.f;
    for (int q = 0; q < B; ++q) {   // B is small: every workgroup recomputes the two channel sums
        const float ys = ysel[(long)q * C + c];
        const float u = __builtin_fmaf(ys, a, bb);
        const float h = gout[(long)q * C + c] * (u > 0.f ? 1.f : slope);
        sb += h;
        sg = __builtin_fmaf(h, (ys - mu) * r, sg);
        if (q == b) hb = h;
    }
    if (blockIdx.y == 0 && wave == 0) {
        dbeta[c] = sb;
        dgamma[c] = sg;
    }
    const float invM = 1.0f / ((float)B * (float)N);
    const float db = training ? sb * invM : 0.f, dg = training ? sg * invM * r : 0.f;
    const int an = arg[(long)b * C + c];
    const int n0 = tile * ROWS, n1 = min(N, n0 + ROWS);
    const float *yb = y + (long)b * N * C;
    float *gb = gy + (long)b * N * C;
#pragma unroll 4
    for (int n = n0 + wave; n < n1; n += 4) {
        const float yv = yb[(long)n * C + c];
        gb[(long)n * C + c] = a * ((n == an ? hb : 0.f) - db - (yv - mu) * dg);
    }
}

}  // namespace

extern "C" size_t fsg_bn_act_workspace_bytes(long M, int C) {
    return sizeof(float) * ((size_t)fsg_cdiv(M, ROWS) * 3 * C + fsg_ec_finalize_stage_floats(C));
}

extern "C" int fsg_bn_act_fwd_f32(const float *y, const float *gamma, const float *beta, float *running_mean,
                                  float *running_var, long M, int C, int training, float momentum, float eps,
                                  float slope, float *out, float *mean, float *invstd, float *workspace,
                                  fsg_stream_t stream) {
    FSG_REQUIRE(y && gamma && beta && out && mean && invstd, "fsg_bn_act_fwd_f32: NULL pointer");
    FSG_REQUIRE(M > 0 && C > 0 && C % 64 == 0, "fsg_bn_act_fwd_f32: bad shape M=%ld C=%d (C must be a multiple of 64)", M, C);
    FSG_REQUIRE(!training || workspace, "fsg_bn_act_fwd_f32: training needs the workspace");
    hipStream_t st = (hipStream_t)stream;
    if (training) {
        const int R = fsg_cdiv(M, ROWS);
        hipLaunchKernelGGL(bnact_stats_kernel, dim3(C / 64, R), dim3(256), 0, st, y, M, C, workspace);
        FSG_CHECK_LAUNCH("fsg_bn_act_fwd_f32/stats");
        const int rc = fsg_ec_finalize_launch(workspace, R, C, eps, momentum, mean, invstd, running_mean, running_var, st);
        if (rc != FSG_OK) return rc;
    }
    const long total4 = M * C / 4;
    const int grid = (int)((total4 + 255) / 256 > 8192 ? 8192 : (total4 + 255) / 256);
    hipLaunchKernelGGL(bnact_apply_kernel, dim3(grid), dim3(256), 0, st, y, gamma, beta, mean, invstd, total4, C, slope, out);
    FSG_CHECK_LAUNCH("fsg_bn_act_fwd_f32/apply");
    return FSG_OK;
}

extern "C" int fsg_bn_act_bwd_f32(const float *grad_out, const float *y, const float *gamma, const float *beta,
                                  const float *mean, const float *invstd, long M, int C, int training, float slope,
                                  float *grad_y, float *grad_gamma, float *grad_beta, float *workspace,
                                  fsg_stream_t stream) {
    FSG_REQUIRE(grad_out && y && gamma && beta && mean && invstd && grad_y && grad_gamma && grad_beta && workspace,
                "fsg_bn_act_bwd_f32: NULL pointer");
    FSG_REQUIRE(M > 0 && C > 0 && C % 64 == 0, "fsg_bn_act_bwd_f32: bad shape M=%ld C=%d", M, C);
    hipStream_t st = (hipStream_t)stream;
    const int R = fsg_cdiv(M, ROWS);
    hipLaunchKernelGGL(bnact_bwd_reduce_kernel, dim3(C / 64, R), dim3(256), 0, st, grad_out, y, gamma, beta, mean, invstd,
                       M, C, slope, workspace);
    FSG_CHECK_LAUNCH("fsg_bn_act_bwd_f32/reduce");
    const int rc = fsg_ec_sum_launch(workspace, R, C, 2, grad_beta, grad_gamma, st);
    if (rc != FSG_OK) return rc;
    const long total = M * C;
    const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(bnact_bwd_apply_kernel, dim3(grid), dim3(256), 0, st, grad_out, y, gamma, beta, mean, invstd,
                       grad_beta, grad_gamma, total, C, training, 1.0f / (float)M, slope, grad_y);
    FSG_CHECK_LAUNCH("fsg_bn_act_bwd_f32/apply");
    return FSG_OK;
}

extern "C" size_t fsg_bn_act_max_workspace_bytes(int B, int N, int C) {
    const size_t rec = (size_t)B * fsg_cdiv(N, ROWS);
    return sizeof(float) * (rec * 5 * (size_t)C + fsg_ec_finalize_stage_floats(C));  // records + stage + per-tile selection
}

extern "C" int fsg_bn_act_max_fwd_f32(const float *y, const float *gamma, const float *beta, float *running_mean,
                                      float *running_var, int B, int N, int C, int training, float momentum, float eps,
                                      float slope, float *out, float *ysel, int32_t *arg, float *mean, float *invstd,
                                      float *workspace, fsg_stream_t stream) {
    FSG_REQUIRE(y && gamma && beta && out && ysel && arg && mean && invstd && workspace, "fsg_bn_act_max_fwd_f32: NULL pointer");
    FSG_REQUIRE(B > 0 && N > 0 && C > 0 && C % 64 == 0, "fsg_bn_act_max_fwd_f32: bad shape B=%d N=%d C=%d", B, N, C);
    hipStream_t st = (hipStream_t)stream;
    const int tiles = fsg_cdiv(N, ROWS), R = B * tiles;
    float *partials = workspace;
    float *sel_val = partials + (size_t)R * 3 * C + fsg_ec_finalize_stage_floats(C);
    int *sel_arg = (int *)(sel_val + (size_t)R * C);
    hipLaunchKernelGGL(bnmax_stats_kernel, dim3(C / 64, R), dim3(256), 0, st, y, gamma, N, C, training, partials, sel_val,
                       sel_arg);
    FSG_CHECK_LAUNCH("fsg_bn_act_max_fwd_f32/stats");
    if (training) {
        const int rc = fsg_ec_finalize_launch(partials, R, C, eps, momentum, mean, invstd, running_mean, running_var, st);
        if (rc != FSG_OK) return rc;
    }
    hipLaunchKernelGGL(bnmax_finish_kernel, dim3(fsg_cdiv(C, 256), B), dim3(256), 0, st, sel_val, sel_arg, gamma, beta, mean,
                       invstd, tiles, C, slope, out, ysel, arg);
    FSG_CHECK_LAUNCH("fsg_bn_act_max_fwd_f32/finish");
    return FSG_OK;
}

extern "C" int fsg_bn_act_max_bwd_f32(const float *grad_out, const float *y, const float *ysel, const int32_t *arg,
                                      const float *gamma, const float *beta, const float *mean, const float *invstd, int B,
                                      int N, int C, int training, float slope, float *grad_y, float *grad_gamma,
                                      float *grad_beta, fsg_stream_t stream) {
    FSG_REQUIRE(grad_out && y && ysel && arg && gamma && beta && mean && invstd && grad_y && grad_gamma && grad_beta,
                "fsg_bn_act_max_bwd_f32: NULL pointer");
    FSG_REQUIRE(B > 0 && N > 0 && C > 0 && C % 64 == 0, "fsg_bn_act_max_bwd_f32: bad shape B=%d N=%d C=%d", B, N, C);
    const int tiles = fsg_cdiv(N, ROWS);
    hipLaunchKernelGGL(bnmax_bwd_kernel, dim3(C / 64, B * tiles), dim3(256), 0, (hipStream_t)stream, grad_out, y, ysel, arg,
                       gamma, beta, mean, invstd, B, N, C, training, slope, grad_y, grad_gamma, grad_beta);
    FSG_CHECK_LAUNCH("fsg_bn_act_max_bwd_f32");
    return FSG_OK;
}

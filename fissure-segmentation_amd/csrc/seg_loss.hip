// Segmentation loss, value and gradient in two launches -- include/fsg_hip.h: fsg_nnu_loss_f32.
// Replaces losses/nnu_loss.py:6-19: CrossEntropyLoss(class_weights) + GDL(softmax, batch_dice=True)
// (losses/dice_loss.py:24-96).  The loss is the last node of the training step, so the same pass that evaluates it can
// hand back d loss / d logits; the autograd wrapper only scales it by the incoming gradient.
//
//   pass 1  one lane per point: softmax over the C classes in registers, per-class sums  tp_c = sum p_c [y=c],
//           sp_c = sum p_c, cnt_c = sum [y=c], and the cross-entropy numerator / denominator; wave shuffles, then a fixed
//           order over the block's waves -> one fp64 record per block (no atomics: the result is reproducible).
//   pass 2  every block folds the (<= 64) block records in the same order, derives the scalars
//           tp, fp, fn, dice, ce, and writes the gradient of its points:
//              d(-dice)/dp_c(i) = num / (vol_c den^2) - [y_i=c] 2 / (vol_c den),   num = 2tp+s, den = 2tp+fp+fn+s
//              dL/dz_c(i) = p_c (g_c - sum_c' p_c' g_c') + w[y_i]/sum_w (p_c - [y_i=c])
// HBM-bound and tiny (C2: 16 384 points x 4 classes): two reads of the logits, one write of the gradient.
#include "fsg_common.h"

namespace {

constexpr int BLOCK = 256;
constexpr int MAX_BLOCKS = 64;

template <int MAXC>
struct PointSoftmax {
    float p[MAXC];
    float logp_y;
    int y;
};

template <int MAXC>
__device__ __forceinline__ void load_softmax(const float *__restrict__ logits, long base, long sc, int C,
                                             const int64_t *__restrict__ labels, long pt, PointSoftmax<MAXC> &o) {
    float z[MAXC];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        z[c] = c < C ? logits[base + c * sc] : -INFINITY;
        m = fmaxf(m, z[c]);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        o.p[c] = c < C ? expf(z[c] - m) : 0.f;
        s += o.p[c];
    }
    const float inv = 1.0f / s;
    const float lse = m + logf(s);
    const long yl = labels[pt];
    o.y = (yl >= 0 && yl < C) ? (int)yl : -1;
    o.logp_y = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        o.p[c] *= inv;
        if (c == o.y) o.logp_y = z[c] - lse;
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// record layout per block: [tp_c | sp_c | cnt_c] (3*C) then ce_num, ce_den, n_bad
template <int MAXC>
__global__ __launch_bounds__(BLOCK) void nnu_partial_kernel(const float *__restrict__ logits, long sb, long sc, long sn,
                                                            const int64_t *__restrict__ labels,
                                                            const float *__restrict__ cw, int C, int N, long P,
                                                            double *__restrict__ rec) {
    __shared__ double part[BLOCK / 64][3 * MAXC + 3];
    float tp[MAXC], sp[MAXC], cnt[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) tp[c] = sp[c] = cnt[c] = 0.f;
    float ce_num = 0.f, ce_den = 0.f, bad = 0.f;
    for (long pt = (long)blockIdx.x * BLOCK + threadIdx.x; pt < P; pt += (long)gridDim.x * BLOCK) {
        const long b = pt / N, n = pt - b * N;
        PointSoftmax<MAXC> s;
        load_softmax<MAXC>(logits, b * sb + n * sn, sc, C, labels, pt, s);
        if (s.y < 0) { bad += 1.f; continue; }
        const float w = cw ? cw[s.y] : 1.0f;
        ce_num -= w * s.logp_y;
        ce_den += w;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            sp[c] += s.p[c];
            if (c == s.y) { tp[c] += s.p[c]; cnt[c] += 1.f; }
        }
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const double a = wave_sum((double)tp[c]), s2 = wave_sum((double)sp[c]), n2 = wave_sum((double)cnt[c]);
        if (lane == 0 && c < C) { part[wave][c] = a; part[wave][C + c] = s2; part[wave][2 * C + c] = n2; }
    }
    {
        const double a = wave_sum((double)ce_num), d = wave_sum((double)ce_den), e = wave_sum((double)bad);
        if (lane == 0) { part[wave][3 * C] = a; part[wave][3 * C + 1] = d; part[wave][3 * C + 2] = e; }
    }
    __syncthreads();
    const int R = 3 * C + 3;
    if ((int)threadIdx.x < R) {
        double v = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) v += part[w][threadIdx.x];
        rec[(long)blockIdx.x * R + threadIdx.x] = v;
    }
}

template <int MAXC>
__global__ __launch_bounds__(BLOCK) void nnu_grad_kernel(const float *__restrict__ logits, long sb, long sc, long sn,
                                                         const int64_t *__restrict__ labels,
                                                         const float *__restrict__ cw, int C, int N, long P,
                                                         const double *__restrict__ rec, int nrec, float w_ce,
                                                         float w_dice, float smooth, float *__restrict__ loss_out,
                                                         float *__restrict__ grad, long gb, long gc, long gn) {
    __shared__ double tot[3 * MAXC + 3];
    __shared__ float coefA[MAXC], coefE[MAXC];
    __shared__ float ce_scale;
    const int R = 3 * C + 3;
    {
        // fold the per-block records: the workgroup as (group, component) with Rp = 2^m >= R components per group; every
        // group walks its share of the records (one thread per component walking ALL of them was a chain of nrec dependent
        // L2 round trips: 13 of the kernel's 19 us at 64 records), then the groups are merged in a fixed tree
        __shared__ double part[BLOCK];
        int Rp = 16;
        while (Rp < R) Rp <<= 1;                      // R <= 99 -> Rp <= 128 <= BLOCK
        const int comp = threadIdx.x & (Rp - 1), grp = threadIdx.x / Rp, ngrp = BLOCK / Rp;
        double v = 0.0;
        if (comp < R)
            for (int r = grp; r < nrec; r += ngrp) v += rec[(long)r * R + comp];
        part[threadIdx.x] = v;
        __syncthreads();
        for (int sft = ngrp >> 1; sft >= 1; sft >>= 1) {
            if (grp < sft) part[threadIdx.x] += part[threadIdx.x + sft * Rp];
            __syncthreads();
        }
        if ((int)threadIdx.x < R) tot[threadIdx.x] = part[threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tp = 0.0, fp = 0.0, fn = 0.0;
        for (int c = 0; c < C; ++c) {
            const double vol = tot[2 * C + c] + 1e-6;
            tp += tot[c] / vol;
            fp += (tot[C + c] - tot[c]) / vol;
            fn += (tot[2 * C + c] - tot[c]) / vol;
        }
        const double num = 2.0 * tp + smooth, den = 2.0 * tp + fp + fn + smooth;
        const double gdl = -num / den;
        const double ce = tot[3 * C] / tot[3 * C + 1];
        for (int c = 0; c < C; ++c) {
            const double vol = tot[2 * C + c] + 1e-6;
            coefA[c] = (float)(w_dice * num / (vol * den * den));
            coefE[c] = (float)(-w_dice * 2.0 / (vol * den));
        }
        ce_scale = (float)(w_ce / tot[3 * C + 1]);
        if (blockIdx.x == 0) {
            loss_out[0] = (float)(w_ce * ce + w_dice * gdl);
            loss_out[1] = (float)ce;
            loss_out[2] = (float)gdl;
            loss_out[3] = (float)tot[3 * C + 2];
        }
    }
    __syncthreads();
    if (!grad) return;
    for (long pt = (long)blockIdx.x * BLOCK + threadIdx.x; pt < P; pt += (long)gridDim.x * BLOCK) {
        const long b = pt / N, n = pt - b * N;
        PointSoftmax<MAXC> s;
        load_softmax<MAXC>(logits, b * sb + n * sn, sc, C, labels, pt, s);
        const long go = b * gb + n * gn;
        if (s.y < 0) {
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) grad[go + c * gc] = 0.f;
            continue;
        }
        float g[MAXC];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            g[c] = c < C ? coefA[c] + (c == s.y ? coefE[c] : 0.f) : 0.f;
            dot += s.p[c] * g[c];
        }
        const float wsc = (cw ? cw[s.y] : 1.0f) * ce_scale;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) grad[go + c * gc] = s.p[c] * (g[c] - dot) + wsc * (s.p[c] - (c == s.y ? 1.f : 0.f));
    }
}

inline int n_blocks(long P) {
    const int nb = fsg_cdiv(P, BLOCK);
    return nb < 1 ? 1 : (nb > MAX_BLOCKS ? MAX_BLOCKS : nb);
}

template <int MAXC>
void launch(const float *logits, long sb, long sc, long sn, const int64_t *labels, const float *cw, int C, int N, long P,
            float w_ce, float w_dice, float smooth, float *loss_out, float *grad, long gb, long gc, long gn,
            double *rec, hipStream_t st) {
    const int nb = n_blocks(P);
    hipLaunchKernelGGL(nnu_partial_kernel<MAXC>, dim3(nb), dim3(BLOCK), 0, st, logits, sb, sc, sn, labels, cw, C, N, P,
                       rec);
    hipLaunchKernelGGL(nnu_grad_kernel<MAXC>, dim3(grad ? nb : 1), dim3(BLOCK), 0, st, logits, sb, sc, sn, labels, cw, C,
                       N, P, rec, nb, w_ce, w_dice, smooth, loss_out, grad, gb, gc, gn);
}

}  // namespace

extern "C" size_t fsg_nnu_loss_workspace_bytes(int C) {
    return (size_t)MAX_BLOCKS * (3 * (size_t)(C > 0 ? C : 0) + 3) * sizeof(double);
}

extern "C" int fsg_nnu_loss_f32(const float *logits, int64_t stride_b, int64_t stride_c, int64_t stride_n,
                                const int64_t *labels, const float *class_weights, int B, int C, int N, float w_ce,
                                float w_dice, float smooth, float *loss_out, float *grad, int64_t gstride_b,
                                int64_t gstride_c, int64_t gstride_n, void *workspace, void *stream) {
    FSG_REQUIRE(logits && labels && loss_out && workspace, "fsg_nnu_loss_f32: NULL pointer");
    FSG_REQUIRE(B > 0 && N > 0 && C >= 2 && C <= 32, "fsg_nnu_loss_f32: bad shape B=%d C=%d N=%d (2 <= C <= 32)", B, C, N);
    FSG_REQUIRE(((uintptr_t)workspace & 7) == 0, "fsg_nnu_loss_f32: workspace must be 8-byte aligned");
    const long P = (long)B * N;
    hipStream_t st = (hipStream_t)stream;
    double *rec = (double *)workspace;
    if (C <= 4)
        launch<4>(logits, stride_b, stride_c, stride_n, labels, class_weights, C, N, P, w_ce, w_dice, smooth, loss_out, grad,
                  gstride_b, gstride_c, gstride_n, rec, st);
    else if (C <= 8)
        launch<8>(logits, stride_b, stride_c, stride_n, labels, class_weights, C, N, P, w_ce, w_dice, smooth, loss_out, grad,
                  gstride_b, gstride_c, gstride_n, rec, st);
    else if (C <= 16)
        launch<16>(logits, stride_b, stride_c, stride_n, labels, class_weights, C, N, P, w_ce, w_dice, smooth, loss_out,
                   grad, gstride_b, gstride_c, gstride_n, rec, st);
    else
        launch<32>(logits, stride_b, stride_c, stride_n, labels, class_weights, C, N, P, w_ce, w_dice, smooth, loss_out,
                   grad, gstride_b, gstride_c, gstride_n, rec, st);
    FSG_CHECK_LAUNCH("fsg_nnu_loss_f32");
    return FSG_OK;
}

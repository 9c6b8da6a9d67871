/*
 * fsg_oracle.c -- CPU restatement (plain C) of the point-cloud primitives on the hot path of
 * kaftanski/fissure-segmentation.  TEST INFRASTRUCTURE ONLY: nothing in the product package may
 * link, load or call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg do (as the checker / the timed CPU baseline, never as the thing shipped).
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * Pinning: the dense kNN / edge-feature functions are checked against golden vectors produced by
 * importing the reference itself (oracle/make_golden.py -> tests/golden/).  The packed-segment kNN,
 * FPS and Chamfer functions restate third-party code that is absent from the reference tree
 * (pointops_cuda of POSTECH-CVLab/point-transformer, pytorch3d; both unpinned there): for those the
 * parity is UNPINNED at the third-party boundary and only the reference's own call sites pin the
 * wrapper semantics.
 *
 * Arithmetic contract shared bit-for-bit with the HIP kernels (build with -ffp-contract=off):
 *   dot(i,j) = fmaf chain over channels c = 0..C-1 starting from +0  (what v_mfma_f32_32x32x2_f32
 *              computes, k-ordered, one rounding per product)
 *   xx(i)    = the same chain with both operands = x_i
 *   d(i,j)   = (xx(i) - 2*dot(i,j)) + xx(j)          [utils/general_utils.py:49-51, same
 *              association: `xx - 2.0 * xTx + xx.transpose(2, 1)`]
 *   selection order = ascending (d, j): ties go to the lower index (torch.topk leaves tie order
 *              unspecified; this is the build's own rule).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FSG_KNN_FIX_DIAG 1   /* force d(i,i) = 0           (general_utils.py:52)            */
#define FSG_KNN_DROP_FIRST 2 /* take k+1, drop column 0    (general_utils.py:317,320-322)   */

static inline int key_less(float da, int ia, float db, int ib) {
    return (da < db) || (da == db && ia < ib);
}

/* bounded sorted insertion of (d, j) into best[0..m) (ascending), m <= cap */
static inline void topk_push(float *bd, int *bi, int *m, int cap, float d, int j) {
    int n = *m;
    if (n == cap && !key_less(d, j, bd[n - 1], bi[n - 1])) return;
    int p = (n < cap) ? n : n - 1;
    while (p > 0 && key_less(d, j, bd[p - 1], bi[p - 1])) {
        bd[p] = bd[p - 1];
        bi[p] = bi[p - 1];
        --p;
    }
    bd[p] = d;
    bi[p] = j;
    if (n < cap) *m = n + 1;
}

/*
 * Dense kNN graph.  Restates pairwise_dist + knn (utils/general_utils.py:43-53, 315-327) and, with
 * flags = 0, dgcnn_opensrc.knn (models/dgcnn_opensrc.py:34-40: topk-largest of the negated
 * distance == ascending order of the same d, no diagonal fix, self included).
 *   x: (B, C, N) with element strides (sb, sc, 1); only channels [0, c_knn) enter the distance.
 *   idx_out: (B, N, k) int32 ; dist_out: (B, N, k) float or NULL.
 * Returns 0, or -1 for bad arguments (k + drop > N).
 */
int orc_knn_dense_f32(const float *x, int B, int N, long sb, long sc, int c_knn, int k, int flags,
                      int32_t *idx_out, float *dist_out) {
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    const int kk = k + drop;
    if (kk > N || k <= 0) return -1;
    float *xx = (float *)malloc(sizeof(float) * (size_t)N);
    if (!xx) return -1;
    int failed = 0;
    for (int b = 0; b < B; ++b) {
        const float *xb = x + (long)b * sb;
        for (int i = 0; i < N; ++i) {
            float a = 0.f;
            for (int c = 0; c < c_knn; ++c) a = fmaf(xb[c * sc + i], xb[c * sc + i], a);
            xx[i] = a;
        }
        /* the queries are independent (own running top-k list each): spread over the host cores -- the arithmetic and the
         * candidate order per query are what they were, so the result is the same whatever the thread count */
#pragma omp parallel
        {
            float *bd = (float *)malloc(sizeof(float) * (size_t)kk);
            int *bi = (int *)malloc(sizeof(int) * (size_t)kk);
            if (!bd || !bi) {
#pragma omp atomic write
                failed = 1;
            }
#pragma omp for schedule(static)
            for (int i = 0; i < N; ++i) {
                if (!bd || !bi) continue;
                int m = 0;
                for (int j = 0; j < N; ++j) {
                    float dot = 0.f;
                    for (int c = 0; c < c_knn; ++c) dot = fmaf(xb[c * sc + i], xb[c * sc + j], dot);
                    float t = xx[i] - 2.0f * dot;
                    float d = t + xx[j];
                    if ((flags & FSG_KNN_FIX_DIAG) && i == j) d = 0.f;
                    topk_push(bd, bi, &m, kk, d, j);
                }
                for (int s = 0; s < k; ++s) {
                    idx_out[((long)b * N + i) * k + s] = bi[s + drop];
                    if (dist_out) dist_out[((long)b * N + i) * k + s] = bd[s + drop];
                }
            }
            free(bd);
            free(bi);
        }
    }
    free(xx);
    return failed ? -1 : 0;
}

/*
 * Edge features.  Restates create_neighbor_features (models/dgcnn.py:31-36) and get_graph_feature
 * (models/dgcnn_opensrc.py:43-66): edge[b, c, i, s] = x[b, c, idx[b,i,s]] - x[b, c, i] for c < C and
 * edge[b, C + c, i, s] = x[b, c, i].   x: (B, C, N) contiguous, idx: (B, N, k), edge: (B, 2C, N, k).
 */
void orc_edge_features_f32(const float *x, const int32_t *idx, int B, int C, int N, int k,
                           float *edge) {
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int i = 0; i < N; ++i) {
                const float xi = x[((long)b * C + c) * N + i];
                for (int s = 0; s < k; ++s) {
                    const int j = idx[((long)b * N + i) * k + s];
                    edge[(((long)b * 2 * C + c) * N + i) * k + s] = x[((long)b * C + c) * N + j] - xi;
                    edge[(((long)b * 2 * C + C + c) * N + i) * k + s] = xi;
                }
            }
}

/* Backward of the above w.r.t. x (what autograd derives for take_along_dim / repeat / cat). */
void orc_edge_features_bwd_f32(const float *grad_edge, const int32_t *idx, int B, int C, int N,
                               int k, float *grad_x) {
    memset(grad_x, 0, sizeof(float) * (size_t)B * C * N);
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            float *gx = grad_x + ((long)b * C + c) * N;
            for (int i = 0; i < N; ++i)
                for (int s = 0; s < k; ++s) {
                    const int j = idx[((long)b * N + i) * k + s];
                    const float gr = grad_edge[(((long)b * 2 * C + c) * N + i) * k + s];
                    const float gc = grad_edge[(((long)b * 2 * C + C + c) * N + i) * k + s];
                    gx[j] += gr;
                    gx[i] += gc - gr;
                }
        }
}

/*
 * Chamfer nearest neighbours.  Restates what losses/chamfer_loss.py:19 obtains from
 * pytorch3d.loss.chamfer_distance (third party, unpinned): for every x_i the squared L2 distance to
 * its nearest y_j, and the argmin (ties -> lowest j); direct difference form
 *   d = fma(dz, dz, fma(dy, dy, dx*dx)).
 * x: (B, N, 3), y: (B, M, 3) contiguous; dist: (B, N); arg: (B, N) int32.
 */
void orc_chamfer_nn_f32(const float *x, const float *y, int B, int N, int M, float *dist,
                        int32_t *arg) {
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N; ++i) {
            const float *p = x + ((long)b * N + i) * 3;
            float best = INFINITY;
            int bj = 0;
            for (int j = 0; j < M; ++j) {
                const float *q = y + ((long)b * M + j) * 3;
                const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
                const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                if (d < best) {
                    best = d;
                    bj = j;
                }
            }
            dist[(long)b * N + i] = best;
            arg[(long)b * N + i] = bj;
        }
}

/*
 * Packed-segment kNN query.  Restates pointops.KNNQuery (models/pointtransformer/pointops.py:42-62;
 * arithmetic in the absent pointops_cuda): for every new_xyz row, the nsample nearest rows of xyz
 * inside the same batch segment, ascending (d2, index); d2 in the direct form above.  Segments
 * shorter than nsample are padded with (index = first row of the segment, d2 = 1e10) -- the
 * build's own rule, see DESIGN.md.   offset/new_offset: cumulative segment ends, (b) int32.
 * idx: (m, nsample) int32 global row numbers; dist2: (m, nsample) squared distances.
 */
void orc_knn_segment_f32(const float *xyz, const float *new_xyz, const int32_t *offset,
                         const int32_t *new_offset, int b, int nsample, int32_t *idx,
                         float *dist2) {
    float *bd = (float *)malloc(sizeof(float) * (size_t)nsample);
    int *bi = (int *)malloc(sizeof(int) * (size_t)nsample);
    for (int s = 0; s < b; ++s) {
        const int st = s ? offset[s - 1] : 0, en = offset[s];
        const int qs = s ? new_offset[s - 1] : 0, qe = new_offset[s];
        for (int q = qs; q < qe; ++q) {
            const float *p = new_xyz + (long)q * 3;
            int m = 0;
            for (int j = st; j < en; ++j) {
                const float *r = xyz + (long)j * 3;
                const float dx = p[0] - r[0], dy = p[1] - r[1], dz = p[2] - r[2];
                const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                topk_push(bd, bi, &m, nsample, d, j);
            }
            for (int t = 0; t < nsample; ++t) {
                idx[(long)q * nsample + t] = t < m ? bi[t] : st;
                dist2[(long)q * nsample + t] = t < m ? bd[t] : 1e10f;
            }
        }
    }
    free(bd);
    free(bi);
}

/*
 * Farthest point sampling per segment.  Restates pointops.FurthestSampling
 * (models/pointtransformer/pointops.py:16-39; arithmetic in the absent pointops_cuda): the first
 * sample of a segment is its first row, every further sample is the row with the largest distance
 * to the already selected set (running min of squared distances, initialised to 1e10 as at
 * pointops.py:32); ties -> lowest index.   idx: (new_offset[b-1]) int32 global row numbers.
 */
void orc_fps_f32(const float *xyz, const int32_t *offset, const int32_t *new_offset, int b,
                 int32_t *idx) {
    for (int s = 0; s < b; ++s) {
        const int st = s ? offset[s - 1] : 0, en = offset[s];
        const int qs = s ? new_offset[s - 1] : 0, qe = new_offset[s];
        const int n = en - st;
        if (qe <= qs || n <= 0) continue;
        float *md = (float *)malloc(sizeof(float) * (size_t)n);
        for (int i = 0; i < n; ++i) md[i] = 1e10f;
        int cur = st;
        idx[qs] = cur;
        for (int t = qs + 1; t < qe; ++t) {
            const float *p = xyz + (long)cur * 3;
            float best = -1.f;
            int bj = st;
            for (int j = st; j < en; ++j) {
                const float *r = xyz + (long)j * 3;
                const float dx = r[0] - p[0], dy = r[1] - p[1], dz = r[2] - p[2];
                const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                const float v = d < md[j - st] ? d : md[j - st];
                md[j - st] = v;
                if (v > best) {
                    best = v;
                    bj = j;
                }
            }
            cur = bj;
            idx[t] = cur;
        }
        free(md);
    }
}

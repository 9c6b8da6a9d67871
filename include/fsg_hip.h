/*
 * fsg_hip.h -- C ABI of libfsg_hip.so: the MI355X (gfx950) hot path of kaftanski/fissure-segmentation.
 *
 * The reference has no FFI of its own: its seam is the Python nn.Module API.  This
 * header is the boundary that sits UNDER that API; each entry point names the reference code it
 * replaces (paths relative to the reference checkout).  Conventions:
 *   - plain C, no torch types; every pointer is a DEVICE pointer owned by the caller (hipMalloc'd /
 *     torch storage) unless marked "host"; nothing is allocated or freed inside the library;
 *   - row-major contiguous tensors unless explicit element strides are passed;
 *   - `stream` is a hipStream_t (NULL = default stream); calls are asynchronous on that stream and
 *     keep no global state, so the library is re-entrant per stream (one process per GPU under DDP);
 *   - return value: FSG_OK or an FSG_ERR_* code; fsg_last_error() gives the text for the calling
 *     thread.  Nothing throws across the boundary.
 */
#ifndef FSG_HIP_H
#define FSG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSG_OK 0
#define FSG_ERR_ARG 1         /* bad shape / NULL pointer / unsupported size */
#define FSG_ERR_HIP 2         /* a HIP runtime call or launch failed          */
#define FSG_ERR_UNSUPPORTED 3 /* valid request this build cannot serve        */

#define FSG_KNN_FIX_DIAG 1   /* force d(i,i) = 0          -- utils/general_utils.py:52            */
#define FSG_KNN_DROP_FIRST 2 /* select k+1, drop column 0 -- utils/general_utils.py:317,320-322   */

#define FSG_KNN_FORCE_ROWS 4 /* use the general "rows in LDS" kernel even where the MFMA kernel applies (tests) */

#define FSG_KNN_FORCE_MFMA 8 /* (experimental kernels: libfsg_hip_experiments.so only; rejected by libfsg_hip.so)     */

#define FSG_KNN_MAX_K 64

typedef void *fsg_stream_t;

int fsg_version(void);
const char *fsg_last_error(void);

/*
 * Dense kNN graph build: replaces utils/general_utils.py:43-53 (pairwise_dist), :315-327 (knn) and
 * models/dgcnn_opensrc.py:34-40 (knn; flags = 0).  The (B,N,N) distance matrix is never materialised.
 *   x        (B, C, N) fp32, element strides (stride_b, stride_c, 1) -- a channel slice x[:, :3]
 *            is passed without a copy; only channels [0, c_knn) enter the distance
 *   idx_out  (B, N, k) int32, ascending (distance, index)
 *   dist_out (B, N, k) fp32 or NULL
 *   xx_scratch (B, N) fp32 scratch for the squared norms, or NULL (then only the general kernel is used)
 * Arithmetic: d = (xx_i - 2 * dot_ij) + xx_j, dot/xx as channel-ordered fp32 fma chains.
 * Limits: 1 <= k, k + drop <= min(N, FSG_KNN_MAX_K); N <= 32768.
 */
int fsg_knn_dense_f32(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn,
                      int k, int flags, int32_t *idx_out, float *dist_out, float *xx_scratch,
                      fsg_stream_t stream);

/*
 * The same graph build with a caller-owned workspace (what the nn.Module layer calls).  Inside 1024 <= N <= 8192 with
 * c_knn <= 64 (N <= 4096 for 64 < c_knn <= 128) and k + drop <= 64 the graph comes from the coarse-sweep + exact-refine kernel
 * (csrc/knn_split.hip): coarse products on the matrix cores -- ONE fp16 image of the points centred on a sampled mean and
 * scaled by a power of two above 4 channels (v_mfma_f32_32x32x16_f16), two bf16 pieces / three products up to 4 channels
 * (v_mfma_f32_32x32x16_bf16) -- only NOMINATE candidates under a rigorous error bound (~25 per query at k = 20), and only the
 * nominees get the arithmetic above: indices AND distance bits equal fsg_knn_dense_f32's.  Other shapes: fsg_knn_dense_f32 with
 * the workspace as xx_scratch.  workspace may be NULL (then as fsg_knn_dense_f32 with xx_scratch = NULL).
 */
size_t fsg_knn_dense_workspace_bytes(int B, int N, int c_knn);
int fsg_knn_dense_ws_f32(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn,
                         int k, int flags, int32_t *idx_out, float *dist_out, void *workspace,
                         size_t workspace_bytes, fsg_stream_t stream);

/*
 * fsg_knn_dense_ws_f32 over points of up to FOUR channels, plus their per-point product with a small weight:
 * pq_out (B, N, rows_pq) = x^T w_pq^T with w_pq (rows_pq, c_knn) row-major -- the "one plain GEMM" of the FIRST EdgeConv's
 * contract (models/dgcnn.py:212-243: its first 1x1 conv, decomposed per point; fsg_edge_weights_many_f32 makes the weight), a
 * K <= 4 product no matrix unit is needed for.  Where the graph build reads the (B, C, N) points itself (N = 2048, rows
 * addressable in 16-byte pieces) the rows come out of its first launch; otherwise a small launch follows the build.
 * 1 <= c_knn <= 4, rows_pq divides 256.
 */
int fsg_knn_dense_ws_pq_f32(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn, int k, int flags,
                            int32_t *idx_out, float *dist_out, void *workspace, size_t workspace_bytes, const float *w_pq,
                            int rows_pq, float *pq_out, fsg_stream_t stream);

/*
 * The same graph build when the PRODUCER of the points has already prepared it: fsg_edgeconv_apply_f32 (below) with a
 * knn_workspace emits the squared norms, the centred norms and the fp16 operand image of its output on the way, so the build
 * starts at its main kernel (one launch and ~11 us less per feature-space graph of DGCNN-seg).  x_pm = the point-major
 * (B, N, c_knn) copy of the points (the producer's out_pm), workspace = the one handed to the producer
 * (fsg_knn_dense_workspace_bytes(B, N, c_knn) bytes).  Same result bits as fsg_knn_dense_f32 on the same points.
 * c_knn in {16, 32, 64}, N % 64 == 0, 1024 <= N <= 8192, k + drop <= 64; FSG_ERR_UNSUPPORTED otherwise.
 */
int fsg_knn_dense_prepared_f32(const float *x_pm, int B, int N, int c_knn, int k, int flags, int32_t *idx_out, float *dist_out,
                               void *workspace, size_t workspace_bytes, fsg_stream_t stream);

/*
 * Edge features: replaces models/dgcnn.py:31-36 (create_neighbor_features: take_along_dim, repeat,
 * cat) and models/dgcnn_opensrc.py:43-66 (get_graph_feature).
 *   x (B,C,N) fp32, idx (B,N,k) int32 -> edge (B,2C,N,k): [x_j - x_i ; x_i]
 * _bwd is the autograd transpose: grad_edge (B,2C,N,k) -> grad_x (B,C,N), overwritten.
 */
int fsg_edge_gather_fwd_f32(const float *x, const int32_t *idx, float *edge, int B, int C, int N,
                            int k, fsg_stream_t stream);
int fsg_edge_gather_bwd_f32(const float *grad_edge, const int32_t *idx, float *grad_x, int B, int C,
                            int N, int k, fsg_stream_t stream);
/* create_neighbor_features (models/dgcnn.py:15-36) in one call: dynamic kNN graph over channels [0, c_knn) with the point
 * itself as neighbour 0 (:26-27) + the edge tensor.  x (B,C,N) contiguous; idx_out (B,N,k) int32; edge (B,2C,N,k);
 * xx_scratch (B,N) fp32.  (Reference-semantic op: the training path uses the fused fsg_edgeconv* entries instead.) */
int fsg_knn_gather_fused_f32(const float *x, int B, int C, int N, int k, int c_knn, int32_t *idx_out, float *edge,
                             float *xx_scratch, fsg_stream_t stream);
/* ... with the workspace of fsg_knn_dense_workspace_bytes(B, N, c_knn): the graph then comes from fsg_knn_dense_ws_f32 */
int fsg_knn_gather_fused_ws_f32(const float *x, int B, int C, int N, int k, int c_knn, int32_t *idx_out, float *edge,
                                void *workspace, size_t workspace_bytes, fsg_stream_t stream);
/* bf16 storage of the same op (BASELINE configs 3-5; SURVEY 8d counts the edge tensor at 2 bytes per element):
 * x, edge and grad_edge are bf16 (raw 16-bit patterns), the difference is formed in fp32 and rounded once;
 * grad_x is accumulated and returned in FP32 (B,C,N). */
int fsg_edge_gather_fwd_bf16(const void *x, const int32_t *idx, void *edge, int B, int C, int N, int k,
                             fsg_stream_t stream);
int fsg_edge_gather_bwd_bf16(const void *grad_edge, const int32_t *idx, float *grad_x, int B, int C, int N,
                             int k, fsg_stream_t stream);

/*
 * Reverse graph (CSR by destination) of a kNN graph -- needed by the gather-style backward below.
 *   idx (B,N,k) int32 -> rowptr (B,N+1) int32, col (B,N*k) int32 with col = (source point << 6) | slot.
 *   The in-edges of a destination are sorted ascending by (source, slot) -- the backward that walks them is then
 *   reproducible from run to run.  Destinations with more than 1024 in-edges (hub points) are sorted through a copy in
 *   the workspace; without a workspace they keep the order they were filled in.
 *   workspace: fsg_graph_reverse_csr_workspace_bytes(B,N,k) bytes, or NULL (then one workgroup per cloud builds the
 *   graph; with the workspace 16 workgroups per cloud share the edge list: count / scan / fill).
 */
size_t fsg_graph_reverse_csr_workspace_bytes(int B, int N, int k);
int fsg_graph_reverse_csr(const int32_t *idx, int B, int N, int k, int32_t *rowptr, int32_t *col,
                          void *workspace, fsg_stream_t stream);

/*
 * Weight of the per-point GEMM behind the fused EdgeConv: the first 1x1 conv over [x_j - x_i ; x_i] with
 * W = [W_rel | W_ctr] (Co, 2C) (models/dgcnn.py:234-241, :282-323) equals P_j + Q_i with [P | Q] = x [W_rel ; W_ctr - W_rel]^T.
 *   _fwd: W (Co,2C) -> Wt (2Co,C);   _bwd: grad_Wt (2Co,C) -> grad_W (Co,2C) (overwritten)
 */
int fsg_edge_weights_fwd_f32(const float *W, int Co, int C, float *Wt, fsg_stream_t stream);
int fsg_edge_weights_bwd_f32(const float *grad_Wt, int Co, int C, float *grad_W, fsg_stream_t stream);
/* the same for up to 8 layers in ONE launch (the transforms depend on the weights only): src/dst per layer are W -> Wt
 * (backward == 0) or grad_Wt -> grad_W (backward != 0); `jobs` is read on the host during the call */
#define FSG_EDGE_WEIGHT_MAX_JOBS 8
typedef struct fsg_edge_weight_jobs {
    const float *src[FSG_EDGE_WEIGHT_MAX_JOBS];
    float *dst[FSG_EDGE_WEIGHT_MAX_JOBS];
    int Co[FSG_EDGE_WEIGHT_MAX_JOBS];
    int C[FSG_EDGE_WEIGHT_MAX_JOBS];
    int n;
} fsg_edge_weight_jobs;
int fsg_edge_weights_many_f32(const fsg_edge_weight_jobs *jobs, int backward, fsg_stream_t stream);

/*
 * Last pass of the fused EdgeConv forward on its own: out (B,Co,N) [+ out_pm (B,N,Co)] = LeakyReLU(BatchNorm(ysel)) from the
 * selected pre-norm values, for callers that ran fsg_edgeconv{1,2}_fwd_* with out == NULL (those then stop after the
 * statistics).  knn_workspace != NULL (Co == 64, N % 64 == 0, both layouts): the pass also prepares the feature-space graph
 * build of the NEXT layer over its own output -- see fsg_knn_dense_prepared_f32.
 */
int fsg_edgeconv_apply_f32(const float *ysel, const float *gamma, const float *beta, const float *mean, const float *invstd,
                           int B, int N, int Co, float slope, float *out, float *out_pm, void *knn_workspace,
                           size_t knn_workspace_bytes, fsg_stream_t stream);

/*
 * The same pass (knn_workspace required) that ALSO emits the per-point rows of the next fused EdgeConv over this block's
 * output: pq_next (B, N, rows_next) = out_pm w_next^T with w_next (rows_next, Co) row-major = the [W_rel ; W_ctr - W_rel] weight
 * of the next block's first conv (fsg_edge_weights_many_f32) -- the "one plain GEMM" of fsg_edgeconv{1,2}_fwd_f32's contract
 * (models/dgcnn.py:212-243: the next EdgeConv's first 1x1 conv, decomposed per point), computed on the tile the pass holds in
 * LDS by the exact fp32 matrix instruction instead of by a library launch.  Co == 64, rows_next == 128, N % 64 == 0.
 */
int fsg_edgeconv_apply_pq_f32(const float *ysel, const float *gamma, const float *beta, const float *mean, const float *invstd,
                              int B, int N, int Co, float slope, float *out, float *out_pm, void *knn_workspace,
                              size_t knn_workspace_bytes, const float *w_next, int rows_next, float *pq_next,
                              fsg_stream_t stream);

/*
 * Fused EdgeConv with ONE shared-MLP layer: replaces models/dgcnn.py:234-241 (gather, 1x1 Conv2d, BatchNorm2d,
 * LeakyReLU, max over k) and the get_graph_feature -> conv -> max blocks of models/folding_net.py:120-133.
 * The caller supplies the per-point rows of the decomposed conv (W = [W_rel | W_ctr]):
 *   pq (B,N,2*Co) = [ x^T W_rel^T | x^T (W_ctr - W_rel)^T ]      (one plain GEMM), Co % 64 == 0, k <= 64.
 * Forward outputs: out (B,Co,N) and, if non-NULL, out_pm (B,N,Co) (the same values point-major, the layout the
 *   point-wise head consumes); saved for backward: ysel (B,N,Co) selected pre-BN value, arg (B,N,Co) uint8
 *   selected slot, ssum (B,N,Co) = sum_s y (training only), mean/invstd (Co) (outputs when training, inputs
 *   -- running statistics -- otherwise).  running_mean/var (nullable) are updated in place when training.
 *   workspace: fsg_edgeconv1_workspace_bytes() floats-as-bytes (training only).
 * Backward: grad_out (B,Co,N) and/or up to two point-major gradients grad_out_pm / grad_out_pm2 (B,N,Co) with row strides
 *   ld_pm / ld_pm2 >= Co in elements (any of the three may be NULL; they are summed) -> grad_pq (B,N,2*Co), grad_gamma, grad_beta (Co); h_scratch (B,N,Co) and
 *   workspace (2*Co*B*ceil(N/64) floats) are caller-provided scratch.
 */
size_t fsg_edgeconv1_workspace_bytes(int B, int N, int Co);
int fsg_edgeconv1_fwd_f32(const float *pq, const int32_t *idx, const float *gamma, const float *beta,
                          float *running_mean, float *running_var, int B, int N, int k, int Co, int training,
                          float momentum, float eps, float slope, float *out, float *out_pm, float *ysel,
                          uint8_t *arg, float *ssum, float *mean, float *invstd, float *workspace,
                          fsg_stream_t stream);
int fsg_edgeconv1_bwd_f32(const float *grad_out, const float *grad_out_pm, int64_t ld_pm, const float *grad_out_pm2,
                          int64_t ld_pm2, const float *pq, const int32_t *rowptr, const int32_t *col,
                          const float *gamma, const float *beta, const float *mean, const float *invstd,
                          const float *ysel, const uint8_t *arg, const float *ssum, int B, int N, int k, int Co,
                          int training, float slope, float *grad_pq, float *grad_gamma, float *grad_beta,
                          float *h_scratch, float *workspace, fsg_stream_t stream);

/*
 * Fused EdgeConv with TWO shared-MLP layers (2C -> 64 -> C2, C2 = 64 or 128): replaces models/dgcnn.py:234-241 for
 * len(shared_mlp) == 2 (ec1 of DGCNNSeg, the EdgeConv of the SpatialTransformer).  Layer 1 arrives decomposed as
 * pq (B,N,128) = [P | Q] like fsg_edgeconv1; w2 (C2,64) is the second 1x1 conv; *1 / *2 are the two BatchNorms.
 * Forward saves ssum1 (B,N,64), mean1/invstd1, ysel2/arg2/ssum2 (B,N,C2), mean2/invstd2 for the backward
 * (mean/invstd are INPUTS -- running statistics -- when training == 0).  Backward (output gradients as for
 * fsg_edgeconv1_bwd_f32) returns grad_pq (B,N,128),
 * grad_w2 (C2,64) and the four BatchNorm parameter gradients.  Workspaces: fsg_edgeconv2_workspace_bytes /
 * fsg_edgeconv2_bwd_workspace_bytes (the backward one holds the only per-edge tensor, du1 (B*N*k, 64)).
 */
size_t fsg_edgeconv2_workspace_bytes(int B, int N, int k, int C2);
size_t fsg_edgeconv2_bwd_workspace_bytes(int B, int N, int k, int C2);
int fsg_edgeconv2_fwd_f32(const float *pq, const int32_t *idx, const float *w2, const float *gamma1,
                          const float *beta1, float *running_mean1, float *running_var1, const float *gamma2,
                          const float *beta2, float *running_mean2, float *running_var2, int B, int N, int k, int C2,
                          int training, float momentum1, float momentum2, float eps1, float eps2, float slope,
                          float *out, float *out_pm, float *ssum1, float *mean1, float *invstd1, float *ysel2,
                          uint8_t *arg2, float *ssum2, float *mean2, float *invstd2, void *workspace,
                          fsg_stream_t stream);
int fsg_edgeconv2_bwd_f32(const float *grad_out, const float *grad_out_pm, int64_t ld_pm, const float *grad_out_pm2,
                          int64_t ld_pm2, const float *pq, const int32_t *idx,
                          const int32_t *rowptr, const int32_t *col, const float *w2, const float *gamma1,
                          const float *beta1, const float *mean1, const float *invstd1, const float *ssum1,
                          const float *gamma2, const float *beta2, const float *mean2, const float *invstd2,
                          const float *ysel2, const uint8_t *arg2, int B, int N, int k, int C2, int training,
                          float slope, float *grad_pq, float *grad_w2, float *grad_gamma1, float *grad_beta1,
                          float *grad_gamma2, float *grad_beta2, void *workspace, fsg_stream_t stream);
/* bf16 operand mode of the same two entry points (same arguments, every tensor fp32): the per-edge 64 x C2 contraction
 * of the forward and the three per-tile products of the backward (y2 recompute, dz1 = dy2 W2, dW2 += dy2^T z1) run on
 * v_mfma_f32_32x32x16_bf16 -- operands rounded to bf16 (once, on their way INTO the LDS operand image at C2 = 64; at every
 * fragment read at C2 = 128), fp32 accumulation; gathers, BatchNorm statistics, selection and all stored tensors are
 * unchanged.  The graph (idx) is always built in fp32.  (The fp32 entry points run the same kernels at C2 = 64 with three
 * bf16 pieces per operand and six products -- fp32-grade, see fsg_pw_* below; v_mfma_f32_32x32x2_f32 at C2 = 128.) */
int fsg_edgeconv2_fwd_bf16(const float *pq, const int32_t *idx, const float *w2, const float *gamma1,
                           const float *beta1, float *running_mean1, float *running_var1, const float *gamma2,
                           const float *beta2, float *running_mean2, float *running_var2, int B, int N, int k, int C2,
                           int training, float momentum1, float momentum2, float eps1, float eps2, float slope,
                           float *out, float *out_pm, float *ssum1, float *mean1, float *invstd1, float *ysel2,
                           uint8_t *arg2, float *ssum2, float *mean2, float *invstd2, void *workspace,
                           fsg_stream_t stream);
int fsg_edgeconv2_bwd_bf16(const float *grad_out, const float *grad_out_pm, int64_t ld_pm, const float *grad_out_pm2,
                           int64_t ld_pm2, const float *pq, const int32_t *idx,
                           const int32_t *rowptr, const int32_t *col, const float *w2, const float *gamma1,
                           const float *beta1, const float *mean1, const float *invstd1, const float *ssum1,
                           const float *gamma2, const float *beta2, const float *mean2, const float *invstd2,
                           const float *ysel2, const uint8_t *arg2, int B, int N, int k, int C2, int training,
                           float slope, float *grad_pq, float *grad_w2, float *grad_gamma1, float *grad_beta1,
                           float *grad_gamma2, float *grad_beta2, void *workspace, fsg_stream_t stream);

/*
 * Fused BatchNorm + LeakyReLU on point-major rows (M, C), C % 64 == 0: the stage behind every 1x1 conv of the
 * point-wise head (models/dgcnn.py:282-323 ConvBlock: conv -> BatchNorm -> LeakyReLU; slope 1 = no activation,
 * slope 0 = ReLU).  Forward: y -> out, plus mean/invstd (outputs when training, inputs otherwise) and the in-place
 * running-statistics update; backward: grad_out, y -> grad_y, grad_gamma, grad_beta.
 * workspace: fsg_bn_act_workspace_bytes(M, C) bytes.
 */
size_t fsg_bn_act_workspace_bytes(long M, int C);
int fsg_bn_act_fwd_f32(const float *y, const float *gamma, const float *beta, float *running_mean, float *running_var,
                       long M, int C, int training, float momentum, float eps, float slope, float *out, float *mean,
                       float *invstd, float *workspace, fsg_stream_t stream);
int fsg_bn_act_bwd_f32(const float *grad_out, const float *y, const float *gamma, const float *beta, const float *mean,
                       const float *invstd, long M, int C, int training, float slope, float *grad_y, float *grad_gamma,
                       float *grad_beta, float *workspace, fsg_stream_t stream);

/*
 * BatchNorm + LeakyReLU + max over the N points of each cloud in one stage: the global feature of DGCNNSeg
 * (models/dgcnn.py:134-137: SharedFullyConnected(192,1024) -> AdaptiveMaxPool1d(1)) without materialising the
 * (B*N, C) activation.  y (B,N,C) pre-norm rows -> out (B,C); saved: ysel (B,C) selected pre-norm value, arg (B,C)
 * its point index.  Backward: grad_out (B,C) -> grad_y (B,N,C) (dense: BN statistics terms), grad_gamma, grad_beta.
 */
size_t fsg_bn_act_max_workspace_bytes(int B, int N, int C);
int fsg_bn_act_max_fwd_f32(const float *y, const float *gamma, const float *beta, float *running_mean,
                           float *running_var, int B, int N, int C, int training, float momentum, float eps,
                           float slope, float *out, float *ysel, int32_t *arg, float *mean, float *invstd,
                           float *workspace, fsg_stream_t stream);
int fsg_bn_act_max_bwd_f32(const float *grad_out, const float *y, const float *ysel, const int32_t *arg,
                           const float *gamma, const float *beta, const float *mean, const float *invstd, int B, int N,
                           int C, int training, float slope, float *grad_y, float *grad_gamma, float *grad_beta,
                           fsg_stream_t stream);

/*
 * Chamfer nearest neighbour, one direction: replaces the pytorch3d.loss.chamfer_distance call of
 * losses/chamfer_loss.py:19 (and losses/mesh_loss.py:29-31, train_pc_ae.py:88).
 *   x (B,N,3), y (B,M,3) fp32 -> dist (B,N) = min_j |x_i - y_j|^2, arg (B,N) int32 (lowest j on ties)
 * _bwd: given g = dLoss/d dist (B,N) and arg, ACCUMULATES into grad_x (B,N,3) and grad_y (B,M,3)
 *   grad_x[i] += 2 g_i (x_i - y_arg) ; grad_y[arg] -= 2 g_i (x_i - y_arg)      (caller zero-fills)
 *   workspace (fsg_chamfer_nn_bwd_workspace_bytes, 4-byte aligned): grad_y is summed per target in ascending query order
 *   through the reverse graph of arg (reproducible); NULL: scattered with hardware fp32 atomics.
 */
size_t fsg_chamfer_nn_bwd_workspace_bytes(int B, int N, int M);
int fsg_chamfer_nn_f32(const float *x, const float *y, int B, int N, int M, float *dist,
                       int32_t *arg, fsg_stream_t stream);
int fsg_chamfer_nn_bwd_f32(const float *x, const float *y, const int32_t *arg, const float *g_dist,
                           int B, int N, int M, float *grad_x, float *grad_y, void *workspace,
                           fsg_stream_t stream);

/*
 * Segmentation loss, value and gradient: replaces losses/nnu_loss.py:6-19, i.e.
 * nn.CrossEntropyLoss(class_weights) + GDL(softmax over dim 1, batch_dice=True, do_bg=True, smooth=1, weights 1/volume)
 * of losses/dice_loss.py:24-96 (the criterion train.py:38 builds for the default --loss nnunet).
 *   logits (B,C,N) fp32 with element strides (stride_b, stride_c, stride_n) -- class-major or point-major alike;
 *   labels (B,N) int64 contiguous, values in [0,C); class_weights (C) fp32 or NULL (= all ones); 2 <= C <= 32
 *   loss_out[4] = { w_ce*ce + w_dice*gdl, ce, gdl, number of labels outside [0,C) (those points are skipped) }
 *   grad (nullable) = d loss_out[0] / d logits, written with its own element strides
 *   workspace: fsg_nnu_loss_workspace_bytes(C) bytes, 8-byte aligned.  Reproducible: no atomics, fixed reduction order.
 */
size_t fsg_nnu_loss_workspace_bytes(int C);
int fsg_nnu_loss_f32(const float *logits, int64_t stride_b, int64_t stride_c, int64_t stride_n,
                     const int64_t *labels, const float *class_weights, int B, int C, int N, float w_ce,
                     float w_dice, float smooth, float *loss_out, float *grad, int64_t gstride_b,
                     int64_t gstride_c, int64_t gstride_n, void *workspace, fsg_stream_t stream);

/*
 * Packed-segment kNN query: replaces pointops_cuda.knnquery_cuda behind
 * models/pointtransformer/pointops.py:42-62.
 *   xyz (n,3), new_xyz (m,3) fp32; offset / new_offset (b) int32 cumulative segment ends
 *   idx (m,nsample) int32 global row numbers, dist2 (m,nsample) SQUARED distances, ascending
 *   (the Python wrapper returns sqrt, as pointops.py:60 does).  Segments shorter than nsample are
 *   padded with (first row of the segment, 1e10).   nsample <= 64.
 */
int fsg_knn_segment_f32(const float *xyz, const float *new_xyz, const int32_t *offset,
                        const int32_t *new_offset, int b, int n, int m, int nsample, int32_t *idx,
                        float *dist2, fsg_stream_t stream);

/*
 * Farthest point sampling per segment: replaces pointops_cuda.furthestsampling_cuda behind
 * models/pointtransformer/pointops.py:16-39.   tmp: (n) fp32 scratch (the reference's `tmp`),
 * idx: (new_offset[b-1]) int32.  First sample of a segment = its first row; ties -> lowest index.
 */
int fsg_fps_f32(const float *xyz, const int32_t *offset, const int32_t *new_offset, int b, int n,
                float *tmp, int32_t *idx, fsg_stream_t stream);

/*
 * Row gather / scatter-add by neighbour index: replaces the fancy-index grouping of
 * models/pointtransformer/pointops.py:114-118 (and pointops_cuda.grouping_*, :78,:94).
 *   feat (n,c), idx (m,ns) -> out (m,ns,c) = feat[idx]       ; bwd ACCUMULATES into grad_feat (n,c)
 */
int fsg_group_gather_fwd_f32(const float *feat, const int32_t *idx, float *out, int n, int c, int m,
                             int ns, fsg_stream_t stream);
int fsg_group_gather_bwd_f32(const float *grad_out, const int32_t *idx, float *grad_feat, int n,
                             int c, int m, int ns, fsg_stream_t stream);

/* pointops.queryandgroup(use_xyz=True) (reference: models/pointtransformer/pointops.py:100-123) as one launch: out (m, ns, 3 + c)
 * = [ xyz[idx] - new_xyz | feat[idx] ].  Backward: the feature columns of grad_out (m, ns, 3 + c) are scattered into grad_feat
 * (n, c), which the caller ZEROES (atomics); coordinates carry no gradient here. */
int fsg_group_xyz_feat_fwd_f32(const float *xyz, const float *new_xyz, const float *feat, const int32_t *idx, float *out, int n,
                               int c, int m, int ns, fsg_stream_t stream);
int fsg_group_xyz_feat_bwd_f32(const float *grad_out, const int32_t *idx, float *grad_feat, int n, int c, int m, int ns,
                               fsg_stream_t stream);

/* max over the ns neighbour rows of x (m, ns, c) with its arg-max (TransitionDown's MaxPool1d, reference: models/pointtransformer/
 * seg_model.py:77-83); the backward writes the whole (m, ns, c) gradient in one launch. */
int fsg_rows_max_fwd_f32(const float *x, float *out, int32_t *arg, int m, int ns, int c, fsg_stream_t stream);
int fsg_rows_max_bwd_f32(const float *grad_out, const int32_t *arg, float *grad_x, int m, int ns, int c, fsg_stream_t stream);

/* pointops.interpolation (reference: models/pointtransformer/pointops.py:198-215) as one launch each way: out (m, c) =
 * sum_j feat[idx[i, j]] w_ij with w_ij = 1 / (sqrt(dist2[i, j]) + 1e-8) normalised over the k neighbours (k <= 8); idx / dist2
 * (m, k) from fsg_knn_segment_f32.  Backward: grad_feat (n, c) must be ZEROED by the caller (atomic accumulation). */
int fsg_interp_fwd_f32(const float *feat, const int32_t *idx, const float *dist2, float *out, int n, int c, int m, int k,
                       fsg_stream_t stream);
int fsg_interp_bwd_f32(const float *grad_out, const int32_t *idx, const float *dist2, float *grad_feat, int n, int c, int m, int k,
                       fsg_stream_t stream);


/*
 * Vector-attention aggregate: replaces models/pointtransformer/seg_model.py:50-52 (and
 * pointops_cuda.aggregation_*, pointops.py:161-195), fused with the value gather:
 *   out[i, ch] = sum_j (v[idx[i,j], ch] + pos[i,j,ch]) * w[i, j, ch mod cw]        cw = c / share_planes
 *   v (n,c), pos (n,ns,c), w (n,ns,cw), idx (n,ns) -> out (n,c)
 * _bwd: grad_out (n,c) -> grad_v (n,c) ACCUMULATED (caller zero-fills), grad_pos (n,ns,c) and
 *   grad_w (n,ns,cw) overwritten.
 */
int fsg_vec_attn_fwd_f32(const float *v, const float *pos, const float *w, const int32_t *idx,
                         float *out, int n, int ns, int c, int cw, fsg_stream_t stream);
int fsg_vec_attn_bwd_f32(const float *v, const float *pos, const float *w, const int32_t *idx,
                         const float *grad_out, float *grad_v, float *grad_pos, float *grad_w, int n,
                         int ns, int c, int cw, fsg_stream_t stream);

/*
 * BatchNorm1d (+ residual) (+ ReLU) over packed point rows: the `relu(bn(linear(x)))` / `relu(bn(linear(y)) + identity)`
 * glue of models/pointtransformer/seg_model.py (:66, :82, :92-99, :138-141, :168).
 *   x, residual (nullable), out: (M,C) fp32 row-major, C in {32,64,128,256,512};  out = [relu]( bn(x) [+ residual] )
 *   training: batch statistics (fp64 sums, fixed order), running buffers updated with torch's rule (nullable);
 *   eval: the caller provides mean / rstd.  workspace: fsg_bn_rows_workspace_bytes(M,C), 8-byte aligned.
 * _bwd: grad_out (M,C) -> grad_x, grad_residual (nullable; = grad_out masked by the ReLU), grad_gamma, grad_beta,
 *   all overwritten; `out` is the forward result (the ReLU mask), needed when relu != 0.
 */
size_t fsg_bn_rows_workspace_bytes(long M, int C);
int fsg_bn_rows_fwd_f32(const float *x, const float *residual, const float *gamma, const float *beta,
                        float *running_mean, float *running_var, long M, int C, int training, float momentum,
                        float eps, int relu, float *out, float *mean, float *rstd, void *workspace,
                        fsg_stream_t stream);
int fsg_bn_rows_bwd_f32(const float *grad_out, const float *x, const float *out, const float *gamma,
                        const float *mean, const float *rstd, long M, int C, int training, int relu,
                        float *grad_x, float *grad_residual, float *grad_gamma, float *grad_beta,
                        void *workspace, fsg_stream_t stream);

/*
 * Small / skinny fp32 GEMM:  C[i,j] = sum_k A(i,k) * B(k,j) (+ bias[j]),  A(i,k) = A[i*sa_i + k*sa_k],
 * B(k,j) = B[k*sb_k + j*sb_j] (element strides: any transposition), C row-major with row stride ldc.
 * Carries the point-wise Linears of models/pointtransformer/seg_model.py (:25-33, :64-69, :92-99, :128-134, :168-169)
 * and their dX / dW products, for which the vendor GEMM launches a single workgroup (see csrc/small_gemm.hip).
 * Reproducible (split reductions are summed in a fixed order).  workspace: fsg_gemm_small_workspace_bytes(I,J,K)
 * bytes (0 when the shape needs no split), 4-byte aligned.
 */
size_t fsg_gemm_small_workspace_bytes(int I, int J, int K);
int fsg_gemm_small_f32(const float *A, int64_t sa_i, int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_j,
                       const float *bias, float *C, int64_t ldc, int I, int J, int K, void *workspace,
                       fsg_stream_t stream);
/* The same product with a by-product: rowsum[i] = sum_k A(i, k) (NULL: none).  With A = dY^T (the weight gradient
 * dW = dY^T X of a Linear, models/pointtransformer/seg_model.py) that is the layer's bias gradient -- one launch instead of the
 * product plus a column reduction.  Same workspace, same fixed summation order. */
int fsg_gemm_small_rowsum_f32(const float *A, int64_t sa_i, int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_j,
                              const float *bias, float *C, int64_t ldc, int I, int J, int K, float *rowsum, void *workspace,
                              fsg_stream_t stream);

/* Deferred split reduction (PointTransformer backward: 50 weight gradients per step, each a tiny output behind a reduction over
 * all points).  fsg_gemm_small_deferred_f32 runs the product only: *splits = S > 1 when the S partial products (and row-sum
 * partials) were left in `workspace`, which the caller keeps alive; 0 when C / rowsum are already final.
 * fsg_gemm_small_reduce_many_f32 then sums up to FSG_GEMM_REDUCE_MAX_JOBS such products in ONE launch, in split order
 * (reproducible).  `blocks` is filled by the callee. */
#define FSG_GEMM_REDUCE_MAX_JOBS 48
typedef struct fsg_gemm_reduce_jobs {
    const float *part[FSG_GEMM_REDUCE_MAX_JOBS];
    float *C[FSG_GEMM_REDUCE_MAX_JOBS];
    float *rowsum[FSG_GEMM_REDUCE_MAX_JOBS];
    int64_t ldc[FSG_GEMM_REDUCE_MAX_JOBS];
    int32_t S[FSG_GEMM_REDUCE_MAX_JOBS], I[FSG_GEMM_REDUCE_MAX_JOBS], J[FSG_GEMM_REDUCE_MAX_JOBS], blocks[FSG_GEMM_REDUCE_MAX_JOBS];
    int32_t n;
} fsg_gemm_reduce_jobs;
int fsg_gemm_small_deferred_f32(const float *A, int64_t sa_i, int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_j, float *C,
                                int64_t ldc, int I, int J, int K, float *rowsum, void *workspace, int *splits,
                                fsg_stream_t stream);
int fsg_gemm_small_reduce_many_f32(const fsg_gemm_reduce_jobs *jobs, fsg_stream_t stream);
/* bf16 operand mode of fsg_gemm_small_rowsum_f32 (splits == NULL) / fsg_gemm_small_deferred_f32 (splits != NULL, bias NULL): both
 * operands are rounded to bf16 (nearest even) on their way into LDS, products on v_mfma_f32_32x32x16_bf16, fp32 accumulation,
 * fp32 output.  The nn.Linear products of the PointTransformer in bf16 mode (reference: autocast, model_trainer.py:75-76,157). */
int fsg_gemm_small_bf16(const float *A, int64_t sa_i, int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_j, const float *bias,
                        float *C, int64_t ldc, int I, int J, int K, float *rowsum, void *workspace, int *splits,
                        fsg_stream_t stream);


/*
 * Fused PointTransformerLayer body: replaces models/pointtransformer/seg_model.py:38-53 after the three
 * q/k/v Linears (:37), i.e. neighbour grouping of keys, values and coordinates, linear_p (Linear(3,3) -> BatchNorm1d(3)
 * -> ReLU -> Linear(3,c)), w = k_j - q_i + p_r, linear_w (BatchNorm1d(c) -> ReLU -> Linear(c,c/8) -> BatchNorm1d(c/8) ->
 * ReLU -> Linear(c/8,c/8)), softmax over the nsample neighbours and the shared-plane aggregate
 *     out[i,ch] = sum_s (v[idx[i,s],ch] + p_r[i,s,ch]) * softmax_s(w)[i,s,ch mod c/8]
 * without any (n,nsample,c) tensor in HBM.  Train-mode BatchNorm statistics run over all n*nsample edges (one pass per
 * BatchNorm, fp64 sums, fixed order) and update the running buffers with torch's rule (unbiased variance).
 *   p (n,3) fp32; idx (n,ns) int32 rows of p/q/k/v (every entry valid -- fsg_knn_segment_f32 pads short segments);
 *   q, k, v (n,c) fp32 with a common row stride ld (in elements: 3c when they are slices of one (n,3c) GEMM output);
 *   c in {32,64,128,256,512}, 1 <= ns <= 16, share_planes = 8.
 *   stats [2*(3+c+c/8)] = mean|rstd of the three BatchNorms (written in training, read in eval: the caller fills them
 *   from the running buffers); u1 (n,ns,c/8) pre-BatchNorm output of the first linear_w Linear and sm (n,ns,c/8) the
 *   softmax weights are kept for the backward; workspace: fsg_pt_attn_workspace_bytes(n, ns, c), 8-byte aligned.
 * _bwd: grad_out (n,c) -> grad_q (n,c, row stride ldg) overwritten; grad_k, grad_v ACCUMULATED (caller zero-fills);
 *   every pointer of `grads` overwritten (same shapes as the parameters); grad_p (n,3) nullable, ACCUMULATED (the
 *   coordinates are network inputs, seg_model.py:202-231: pass NULL unless the input itself needs a gradient).
 */
typedef struct fsg_pt_layer_params {
    const float *lp1_w, *lp1_b;               /* linear_p.0: (3,3), (3) */
    const float *bnp_g, *bnp_b;               /* linear_p.1: (3) */
    float *bnp_rm, *bnp_rv;                   /*   running mean / var, nullable */
    const float *lp2_w, *lp2_b;               /* linear_p.3: (c,3), (c) */
    const float *bn1_g, *bn1_b;               /* linear_w.0: (c) */
    float *bn1_rm, *bn1_rv;
    const float *lw1_w, *lw1_b;               /* linear_w.2: (c/8,c), (c/8) */
    const float *bn2_g, *bn2_b;               /* linear_w.3: (c/8) */
    float *bn2_rm, *bn2_rv;
    const float *lw2_w, *lw2_b;               /* linear_w.5: (c/8,c/8), (c/8) */
    float eps_p, eps_1, eps_2;
    float mom_p, mom_1, mom_2;                /* momentum of this call for the running buffers */
} fsg_pt_layer_params;

typedef struct fsg_pt_layer_grads {
    float *lp1_w, *lp1_b, *bnp_g, *bnp_b, *lp2_w, *lp2_b, *bn1_g, *bn1_b, *lw1_w, *lw1_b, *bn2_g, *bn2_b, *lw2_w,
        *lw2_b;
} fsg_pt_layer_grads;

size_t fsg_pt_attn_workspace_bytes(int n, int ns, int c);
int fsg_pt_attn_fwd_f32(const float *p, const int32_t *idx, const float *q, const float *k, const float *v,
                        int64_t ld, const fsg_pt_layer_params *params, int n, int ns, int c, int training,
                        float *out, float *stats, float *u1, float *sm, void *workspace, fsg_stream_t stream);
int fsg_pt_attn_bwd_f32(const float *p, const int32_t *idx, const float *q, const float *k, const float *v,
                        int64_t ld, const fsg_pt_layer_params *params, int n, int ns, int c, int training,
                        const float *grad_out, const float *stats, const float *u1, const float *sm,
                        float *grad_q, float *grad_k, float *grad_v, int64_t ldg, float *grad_p,
                        const fsg_pt_layer_grads *grads, void *workspace, fsg_stream_t stream);

/*
 * Adam over one flat fp32 buffer -- replaces torch.optim.Adam(model.parameters(), lr, weight_decay) of
 * model_trainer.py:57 once the parameters live in one contiguous buffer (optim.FlatAdam).  Update rule of
 * torch/optim/adam.py (L2 weight decay, no amsgrad, no maximize).  All four arrays (n) 16-byte aligned, updated in place.
 *   state: 8 bytes, 8-byte aligned, zero-initialised by the caller once: { float step; uint32 ticket } -- the step count
 *          lives on the device and is incremented by the call itself (the call is replayable inside a hipGraph).
 *   lr_dev: nullable; when given, the learning rate is read from device memory (a scheduler can change it between
 *          replays), otherwise `lr` is used.
 */
int fsg_adam_flat_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, void *state, int64_t n,
                      float lr, const float *lr_dev, float beta1, float beta2, float eps, float weight_decay,
                      fsg_stream_t stream);

/*
 * Accumulation step of the test-time ensembling `predict_full_pointcloud` (models/point_seg_net.py:21-48; the loop body
 * `softmax_accumulation[..., perm] += softmax(net(pc[..., perm]))` of :27-29 and :40-44) for R runs at once:
 *   logits (R,B,cls,S) fp32 contiguous -- the net's output for the R subsets, run as one batch of R*B clouds
 *   pts    (R,S) int64 -- the point indices of every run (indices outside [0,n_points) are ignored)
 *   acc    (B,cls,n_points) fp32, updated in place:  acc[b,:,pts[r,s]] += softmax_c(logits[r,b,:,s]),  runs applied in
 *          the order r = 0..R-1 per point (same association as the reference's loop; no float atomics).  When a run
 *          names a point more than once, the highest slot s is the one added (the reference's indexed += also adds one).
 *   cls <= 32.  workspace: fsg_ensemble_accumulate_workspace_bytes(R, n_points), 4-byte aligned.
 */
size_t fsg_ensemble_accumulate_workspace_bytes(int R, int64_t n_points);
int fsg_ensemble_accumulate_f32(const float *logits, int R, int B, int cls, int S, const int64_t *pts, int64_t n_points,
                                float *acc, void *workspace, fsg_stream_t stream);

/*
 * On-device sampling + augmentation, the step in front of the path: data.py:435-460 (random `sample_points` subset of an
 * item) after augmentations.py:52-113 (random rotation / scale / translation of the coordinate rows), one pass:
 *   x (B,C,N) fp32;  sample (B,S) int64 column indices in [0,N), or NULL (then S == N: identity subset);
 *   affine (B,12) fp32 = row-major [A | t] (3 x 4, column-vector form  x' = A x + t) applied to rows 0..2, or NULL;
 *   out (B,C,S):  out[b,0:3,i] = A_b x[b,0:3,sample[b,i]] + t_b,  out[b,3:,i] = x[b,3:,sample[b,i]].
 */
int fsg_sample_transform_f32(const float *x, int B, int C, int64_t N, const int64_t *sample, int S, const float *affine,
                             float *out, fsg_stream_t stream);

/*
 * Column sums of a narrow row-major matrix: out[c] = sum_m x[m*C + c], C a power of two <= 32 -- the bias gradient of the
 * last point-wise layer (models/dgcnn.py:146, Conv1d(128, num_classes) with bias) over B*N rows.  One workgroup, fixed
 * summation order.
 */
int fsg_colsum_narrow_f32(const float *x, int64_t M, int C, float *out, fsg_stream_t stream);

/*
 * First layer of a folding MLP (models/folding_net.py:205-221): Conv1d(E + cp, Cout, 1) over cat([code repeated over the m
 * points, pts]) (+ ReLU) with the code part already reduced to per_cloud (B,Cout) = code W[:, :E]^T + bias by the caller:
 *   out[b,i,:] = [relu]( per_cloud[b,:] + sum_{j<cp} pts[b,i,j] * w[:,j] ),   pts (B,m,cp), cp in 1..3, w (Cout,cp) with row
 *   stride ldw (the W[:, E:] slice of the conv weight), out (B,m,Cout) written once.  Cout % 4 == 0, 16-byte aligned rows.
 */
int fsg_fold_layer1_f32(const float *pts, int cp, const float *w, int64_t ldw, const float *per_cloud, int B, int m, int Cout,
                        int relu, float *out, fsg_stream_t stream);

/*
 * Point-wise layers of the DGCNN head on the bf16 matrix pipe with fp32-grade results (csrc/pointwise.hip): replaces the
 * Conv1d(kernel 1) products of models/dgcnn.py:123-137,156-160,282-323.  Every fp32 operand is split into three bf16 pieces
 * (x = h + m + l, exact to 2^-27 |x|) and a product is six v_mfma_f32_32x32x16_bf16 products with fp32 accumulation: as
 * close to real arithmetic as an fp32 fma chain, at 2.7x the rate of the fp32 matrix instruction.
 *
 * fsg_pw_weight_image_f32: the (N, K) matrix W(n, k) = W[n*stride_n + k*stride_k] * scale -> the register image of the MFMA's
 *   B operand (fsg_pw_weight_image_bytes(N, K) bytes; rows and columns padded with zeros).  An image may concatenate several
 *   matrices along k: this call fills k-steps [ks0, ks0 + ceil(K/16)) of an image with KS k-steps per 32-row block.
 * fsg_pw_linear_f32: C (M, N) = A (M, K) W^T (+ bias), A fp32 rows with stride lda (multiple of 4, 16-byte aligned),
 *   K % 32 == 0, W given as its image.  tile: 0 = chosen by shape, 1 = 128x128, 2 = 64x128, 3 = 64x64, 4 = 128x64.
 * fsg_pw_weight_images_f32: up to FSG_PW_MAX_IMAGE_JOBS images in one launch (the weights of a whole head, both orientations).
 */
#define FSG_PW_MAX_IMAGE_JOBS 10
typedef struct fsg_pw_image_jobs {
    const float *W[FSG_PW_MAX_IMAGE_JOBS];
    int64_t stride_n[FSG_PW_MAX_IMAGE_JOBS], stride_k[FSG_PW_MAX_IMAGE_JOBS];
    int N[FSG_PW_MAX_IMAGE_JOBS], K[FSG_PW_MAX_IMAGE_JOBS], ks0[FSG_PW_MAX_IMAGE_JOBS], KS[FSG_PW_MAX_IMAGE_JOBS];
    float scale[FSG_PW_MAX_IMAGE_JOBS];
    void *image[FSG_PW_MAX_IMAGE_JOBS];
    int n;
} fsg_pw_image_jobs;
int fsg_pw_weight_images_f32(const fsg_pw_image_jobs *jobs, fsg_stream_t stream);
size_t fsg_pw_weight_image_bytes(int N, int K);
int fsg_pw_weight_image_f32(const float *W, int64_t stride_n, int64_t stride_k, int N, int K, float scale, int ks0, int KS,
                            void *image, fsg_stream_t stream);
int fsg_pw_linear_f32(const float *A, int64_t lda, const void *image, const float *bias, float *C, int64_t ldc, int M, int N,
                      int K, int tile, fsg_stream_t stream);

/*
 * bf16 operand mode of the same kernels (BASELINE configs 3-5 name bf16; the reference trains under autocast,
 * model_trainer.py:75-76,157): ONE bf16 piece per fp32 operand (round-to-nearest-even), one v_mfma_f32_32x32x16_bf16 product,
 * fp32 accumulation, fp32 storage.  Carries the nn.Linear products of models/pointtransformer/seg_model.py (:25-33, :64-69,
 * :92-99, :128-134, :168-169) when functional.set_mfma_operands("bf16") / an ambient autocast(bfloat16) is on.
 *   fsg_pw_weight_image_bf16: (N, K) weight -> one-piece image of fsg_pw_weight_image_bytes(N, K) / 3 bytes
 *   fsg_pw_linear_bf16:       C (M, N) = bf16(A) bf16(W)^T (+ bias), K % 32 == 0; tile 0 = by shape, 2 = 64 x 128, 3 = 64 x 64
 *   fsg_pw_tn_bf16:           C1 (N1a, N2) = bf16(L1)^T bf16(R) over the M rows (plain single-segment operands of fsg_pw_tn_args),
 *                             tile 2 = 64 x 128, 3 = 64 x 64
 */
int fsg_pw_weight_image_bf16(const float *W, int64_t stride_n, int64_t stride_k, int N, int K, void *image, fsg_stream_t stream);
int fsg_pw_linear_bf16(const float *A, int64_t lda, const void *image, const float *bias, float *C, int64_t ldc, int M, int N,
                       int K, int tile, fsg_stream_t stream);

/*
 * The members of the family behind the fused DGCNN head (`functional.seg_head`; models/dgcnn.py:123-162 of the reference).
 * Tile codes: 1 = 128 x 128, 2 = 64 x 128, 3 = 64 x 64, 4 = 128 x 64, 5 = 64 x 192 (rows x columns of C; 5 only with the BatchNorm-backward prologue + bias epilogue); fsg_pw_tile_rows(tile) = rows.
 *
 * fsg_pw_rowgemm_f32:  C (M, N) = pro(A) (M, K1 + K2) . B,  B given as a weight image with (K1 + K2) / 16 k-steps.
 *   prologue `pro` on segment 1 of A (segment 2 is always plain):
 *     0 none
 *     1 BatchNorm + LeakyReLU of the producing layer:  a = lrelu(alpha[k] A1[m,k] + delta[cloud(m)][k])
 *     2 BatchNorm backward:  a = alpha[k] A1[m,k] f'(alpha[k] Y1[m,k] + delta[cloud][k]) - P[cloud][k] - Q[k] Y1[m,k]
 *       (A1 = gradient w.r.t. the layer's activation, Y1 = its pre-BatchNorm values; P, Q from fsg_pw_bnbwd_finalize_f32)
 *     per-cloud tables have `tstride` floats between clouds (0: one row for all), cloud(m) = m / rows_per_cloud
 *   epilogue `epi` = OR of
 *     1  STORE     columns >= store_n0 go to C[m * ldc + n - store_n0]
 *     2  STATS     rec (M / rows, 3, N): (n, mean, M2) of every column over the tile's rows (train-mode BatchNorm statistics)
 *     4  SEL       columns < sel_n: sel_val (M / rows, sel_n) = max over the tile's rows of sgn[n] * c, sel_arg = its row inside
 *                  the cloud, lowest row on ties (the global max-pool through the monotone BatchNorm + LeakyReLU)
 *     8  BWDSTATS  rec2 (M / rows, 2, N): sums over the tile's rows of h = c f'(ealpha Yp + edelta[cloud]) and
 *                  h (Yp - emu[cloud]) er  (BatchNorm backward sums of the layer whose activation gradient C is)
 *     16 BIAS      bias[n] added on the way out (with STORE)
 *   K1, K2 multiples of 32; A rows 16-byte aligned with strides that are multiples of 4; reducing epilogues need M and
 *   rows_per_cloud to be multiples of the tile's rows.  Only the combinations the head uses are instantiated
 *   (FSG_ERR_UNSUPPORTED otherwise).
 */
typedef struct fsg_pw_rowgemm_args {
    const float *A1, *Y1, *A2;
    int64_t lda1, lda2;
    int K1, K2;
    const void *Bimg;
    int M, N, rows_per_cloud;
    const float *alpha, *delta, *P, *Q;
    int tstride;
    float slope;
    float *C;
    int64_t ldc;
    int store_n0;
    const float *bias;
    float *rec;
    const float *sgn;
    float *sel_val;
    int32_t *sel_arg;
    int sel_n;
    const float *Yp;
    int64_t ldyp;
    const float *ealpha, *edelta, *emu, *er;
    int etstride;
    float *rec2;
} fsg_pw_rowgemm_args;
int fsg_pw_tile_rows(int tile);
int fsg_pw_rowgemm_f32(const fsg_pw_rowgemm_args *args, int pro, int epi, int tile, fsg_stream_t stream);

/*
 * fsg_pw_tn_f32:  [C1 ; C2] (N1a + N1b, N2) = sum_m L'(m, :)^T R'(m, :)  -- weight gradients dW = dy^T a and the Gram matrix
 *   of the global-feature backward; contraction over the M rows, split into slices of rows_per_slice rows (multiple of 32)
 *   whose partial products are summed in slice order (reproducible).  Left operand = [segment 1 | segment 2]: segment 1
 *   (N1a columns) with prologue lpro = 0 or 2 (as above: L1 = gradient, LY1 = pre-BatchNorm values), segment 2 (N1b columns,
 *   N1a % 64 == 0 then) plain; right operand (N2 columns) with rpro = 0 or 1.  Rows of the result below N1a go to C1, the
 *   others to C2.  tile: 1 = 128 x 128, 2 = 64 x 128, 3 = 64 x 64, 5 = 128 x 192 (N1 x N2).  workspace:
 *   fsg_pw_tn_workspace_bytes(N1a + N1b, N2, M, rows_per_slice) bytes.
 *   C1 == NULL: the slices are left in the workspace, to be folded later by fsg_pw_tn_reduce_f32 -- the reductions of several
 *   products in ONE launch (job j: S = ceil(M / rows_per_slice) slices of (N1, N2) in workspace[j]; rows < N1a -> C1, others -> C2).
 */
typedef struct fsg_pw_tn_args {
    const float *L1, *LY1, *L2;
    int64_t ldl1, ldl2;
    int N1a, N1b, lpro;
    const float *lalpha, *ldelta, *lP, *lQ;
    int lts;
    const float *R;
    int64_t ldr;
    int N2, rpro;
    const float *ralpha, *rdelta;
    int rts;
    float slope;
    int M, rows_per_cloud, rows_per_slice;
    int ones;      /* 1: one more left column (behind segment 2) that is all ones: result row N1a + N1b = column sums of R' */
} fsg_pw_tn_args;
#define FSG_PW_MAX_REDUCE_JOBS 6
typedef struct fsg_pw_tn_reduce_jobs {
    const void *workspace[FSG_PW_MAX_REDUCE_JOBS];
    float *C1[FSG_PW_MAX_REDUCE_JOBS], *C2[FSG_PW_MAX_REDUCE_JOBS];
    int64_t ldc1[FSG_PW_MAX_REDUCE_JOBS], ldc2[FSG_PW_MAX_REDUCE_JOBS];
    int S[FSG_PW_MAX_REDUCE_JOBS], N1[FSG_PW_MAX_REDUCE_JOBS], N2[FSG_PW_MAX_REDUCE_JOBS], N1a[FSG_PW_MAX_REDUCE_JOBS];
    int n;
} fsg_pw_tn_reduce_jobs;
int fsg_pw_tn_reduce_f32(const fsg_pw_tn_reduce_jobs *jobs, fsg_stream_t stream);
size_t fsg_pw_tn_workspace_bytes(int N1, int N2, int M, int rows_per_slice);
int fsg_pw_tn_f32(const fsg_pw_tn_args *args, int tile, void *workspace, size_t workspace_bytes, float *C1, int64_t ldc1,
                  float *C2, int64_t ldc2, fsg_stream_t stream);
int fsg_pw_tn_bf16(const fsg_pw_tn_args *args, int tile, void *workspace, size_t workspace_bytes, float *C1, int64_t ldc1,
                   fsg_stream_t stream);

/*
 * fsg_pw_bn_finalize_f32: STATS records (R, 3, ldn), columns [c0, c0 + C) -> train-mode BatchNorm statistics (Chan's merge in
 *   fp64; `shift` (B, C) or NULL is added to the mean of every record of its cloud: y = y0 + shift[cloud]), torch's running
 *   update (unbiased variance), and the tables of the consumers: alpha = gamma invstd, delta (B or 1, C) = alpha (shift - mean)
 *   + beta, emu (B or 1, C) = mean - shift (nullable), cloud_mean (B, C) = unshifted per-cloud mean (nullable).
 *   training == 0: mean / invstd are inputs (running statistics), only the tables are written.
 *   gfeat (B, CG) != NULL: the shift is computed in the kernel first, shift[b][c] = sum_j gfeat[b][j] Wglob[c * ldwg + j] (the
 *   per-cloud constant of the first head layer, models/dgcnn.py:159-160), and written to shift_out == shift (B <= 64).
 * fsg_pw_cloud_linear_f32: out (B, C0) = x (B, CG) W^T, W (C0, CG) with row stride ldw -- the per-cloud constant of the first head
 *   layer (its global-feature block times the max-pooled feature, models/dgcnn.py:159-160); one wave per output.
 * fsg_pw_max_finish_f32: SEL records (B * tiles, C) -> out (B, C) = lrelu(alpha ysel + delta), ysel, arg (row inside the cloud).
 * fsg_pw_bnbwd_finalize_f32: BWDSTATS records (R, 2, C) -> dbeta, dgamma, P (B or 1, C), Q (C) of prologue 2, and (first head
 *   layer, dc != NULL) dc (B, C) = per-cloud column sums of dy = the gradient of the per-cloud constant.
 * fsg_pw_logits_bwd_f32: last layer (Conv1d(C, classes) + bias): da (M, C) = g (M, classes) W3, stored, + BWDSTATS records
 *   (ceil(M / 32), 2, C) of the BatchNorm in front of it.
 */
int fsg_pw_bn_finalize_f32(const float *rec, int R, int ldn, int c0, int C, const float *shift, int B, int training,
                           const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                           float *running_var, float *mean, float *invstd, float *alpha, float *delta, float *emu,
                           float *cloud_mean, const float *gfeat, const float *Wglob, int64_t ldwg, int CG, float *shift_out,
                           fsg_stream_t stream);
int fsg_pw_cloud_linear_f32(const float *x, const float *W, int64_t ldw, int B, int C0, int CG, float *out, fsg_stream_t stream);
int fsg_pw_max_finish_f32(const float *sel_val, const int32_t *sel_arg, const float *sgn, const float *alpha, const float *delta,
                          int B, int tiles, int C, float slope, float *out, float *ysel, int32_t *arg, fsg_stream_t stream);
/* fsg_pw_bn_finalize_f32 (no per-cloud shift) and fsg_pw_max_finish_f32 as ONE launch: the statistics of the BatchNorm in front
 * of the global max-pool and the pool's finish from the SEL records of the same product (models/dgcnn.py:134-137,156); B <= 64. */
int fsg_pw_bn_finalize_max_f32(const float *rec, int R, int ldn, int c0, int C, int B, int training, const float *gamma,
                               const float *beta, float eps, float momentum, float *running_mean, float *running_var, float *mean,
                               float *invstd, float *alpha, float *delta, const float *sel_val, const int32_t *sel_arg,
                               const float *sgn, int tiles, float slope, float *out, float *ysel, int32_t *arg, fsg_stream_t stream);

int fsg_pw_bnbwd_finalize_f32(const float *rec2, int R, int C, int B, int64_t M, int training, const float *alpha,
                              const float *invstd, const float *emu, int emu_per_cloud, const float *cloud_mean, float *dbeta,
                              float *dgamma, float *P, float *Q, float *dc, fsg_stream_t stream);
int fsg_pw_logits_bwd_f32(const float *g, int classes, const float *W3, const float *y, const float *alpha, const float *delta,
                          const float *mean, const float *invstd, int64_t M, int C, float slope, float *da, float *rec2,
                          fsg_stream_t stream);

/*
 * Global-feature layer, backward in its Gram form (the (M, 1024) activation gradient is never formed): after the max over the
 * points only B * C entries of dY are "selected", the BatchNorm terms are affine in y = X W^T, so with G = X^T X, s = sum_m X_m:
 *   dX = selected rows - 1 (W^T P)^T - X (W^T diag(Q) W),   dW = selected rows - P s^T - diag(Q) W G.
 * fsg_pw_gf_prep_f32: per channel dbeta, dgamma, P, Q and coef (B, C) = weight of the selected row in dy.  The gradient of the
 *   global feature is either given (dg (B, C), dc == NULL) or formed here from the gradient dc (B, C0) of the first head layer's
 *   per-cloud constant: dg = dc W0g (W0g (C0, C), row stride ldw0), together with dW0g (C0, C) = dc^T gfeat (B <= 32).
 *   Wq != NULL: also the rows [Q[c] W[c, :] | -P[c]] (C, K + 1) with row stride ldwq -- the left operand of the row contraction
 *   [M1 ; npvec] = [Q o W | -P]^T W (fsg_pw_tn_f32 over the C channel rows).
 * fsg_pw_scatter_rows_f32: dX[b Npts + arg[b,c], :] += coef[b,c] W[c, :], summed per destination row in channel order
 *   (C <= 4096; workspace fsg_pw_scatter_rows_workspace_bytes(B, C) bytes for the sorted selection keys).
 * fsg_pw_gf_dw_f32: dW[c, :] = sum_b coef[b,c] X[b Npts + arg[b,c], :] - P[c] s - Q[c] (W G)[c, :]  (G (K, K) contiguous).
 *   (s = column sums of X and G come out of ONE row contraction: fsg_pw_tn_f32 with `ones` = 1.)
 */
int fsg_pw_gf_prep_f32(const float *dc, const float *W0g, int64_t ldw0, int C0, const float *gfeat, float *dW0g, int64_t lddw0,
                       const float *dg, const float *ysel, const float *alpha, const float *delta, const float *mean,
                       const float *invstd, int B, int C, int64_t M, int training, float slope, float *dbeta, float *dgamma,
                       float *P, float *Q, float *coef, const float *W, int64_t ldw, int K, float *Wq, int64_t ldwq,
                       fsg_stream_t stream);
size_t fsg_pw_scatter_rows_workspace_bytes(int B, int C);
int fsg_pw_scatter_rows_f32(const float *coef, const int32_t *arg, const float *W, int64_t ldw, int B, int C, int K, int Npts,
                            float *dX, int64_t ldx, void *workspace, fsg_stream_t stream);
int fsg_pw_gf_dw_f32(const float *coef, const int32_t *arg, const float *X, int64_t ldx, const float *s, const float *W,
                     int64_t ldw, const float *G, const float *P, const float *Q, int B, int C, int K, int Npts, float *dW,
                     int64_t lddw, fsg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FSG_HIP_H */

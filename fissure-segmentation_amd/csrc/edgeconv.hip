// Fused EdgeConv (one shared-MLP layer): kNN-neighbour gather + 1x1 conv + train-mode BatchNorm +
// LeakyReLU + max over k, forward and backward, without any per-edge tensor in HBM.
// include/fsg_hip.h: fsg_graph_reverse_csr, fsg_edgeconv1_{fwd,bwd}_f32.
// Replaces models/dgcnn.py:234-241 (EdgeConv.forward for len(shared_mlp) == 1) and the
// get_graph_feature -> conv -> max blocks of models/folding_net.py:120-133 of the reference.
//
// Algebra.  The conv over edge features [x_j - x_i ; x_i] with W = [W_rel | W_ctr] is
//     y(i,s) = W_rel x_j + (W_ctr - W_rel) x_i = P_j + Q_i,
// so the caller computes the per-POINT rows P, Q with one small GEMM (k times fewer flops than the
// per-edge conv) and this file only gathers rows of P.  BatchNorm+LeakyReLU is monotone per channel
// (increasing for gamma >= 0, decreasing otherwise), hence  max_s f(BN(y(i,s))) = f(BN(max_s/min_s y(i,s)))
// and the (B,Cout,N,k) activation never exists either.  Train-mode BN needs the statistics of ALL edges:
// they come from per-workgroup shifted sums merged with Chan's formula in fp64 (no E[x^2]-E[x]^2 cancellation).
//
// Backward.  With h_i = dL/dout_i * f'(u_i) on the selected edge, dbeta = sum_i h_i, dgamma = sum_i h_i yhat_i and
//     dy(i,s) = r*gamma * ( h_i [s = arg_i] - dbeta/M - yhat(i,s) dgamma/M ),      M = B N k, r = invstd
// the per-point gradients are dQ_i = sum_s dy(i,s) and dP_j = sum over in-edges of j of dy -- evaluated per
// DESTINATION through the reverse graph (CSR by destination) instead of float atomics:
//     dQ_i = r g ( h_i - k db/M - (dg/M) r (S_i - k mu) ),                 S_i = sum_s y(i,s) (saved by forward)
//     dP_j = r g ( sum_{(i,s)->j, s=arg_i} h_i - deg_j db/M - (dg/M) r (deg_j (P_j - mu) + sum_{i->j} Q_i) ).
// Layouts: pq (B,N,2*Co) rows [P | Q]; ysel/ssum/h (B,N,Co); arg (B,N,Co) uint8; out/grad_out (B,Co,N).
// Lanes run along channels (Co % 64 == 0): every P-row gather is one coalesced 256-byte access served by L2;
// blockIdx.x = cloud, so the workgroups of a cloud share an XCD (round-robin placement) and its L2 keeps that
// cloud's P rows.
#include <type_traits>

#include "fsg_common.h"

size_t fsg_ec_finalize_stage_floats(int Co);
int fsg_ec_finalize_launch(const float *partials, int R, int Co, float eps, float momentum, float *mean, float *invstd,
                           float *running_mean, float *running_var, hipStream_t st);

namespace {

constexpr int TP = 16;  // points per tile in the gather kernels (4 waves x 4 points)
constexpr int FSG_CSR_SPLIT = 16;  // workgroups per cloud of the multi-workgroup reverse-graph build
constexpr int TPW = 2;  // tiles per workgroup of the statistics kernel: 8 waves, one BN partial record per workgroup

__device__ __forceinline__ float lrelu(float u, float slope) { return u > 0.f ? u : u * slope; }

// ------------------------------------------------------------------ reverse graph
__global__ __launch_bounds__(1024) void csr_build_kernel(const int32_t *__restrict__ idx, int N, int k,
                                                          int32_t *__restrict__ rowptr, int32_t *__restrict__ col) {
    extern __shared__ int sh[];
    int *hist = sh;        // [N] counts, later cursors
    int *part = sh + N;    // [1024]
    const int b = blockIdx.x, tid = threadIdx.x;
    const long NK = (long)N * k;
    const int32_t *ib = idx + b * NK;
    for (int j = tid; j < N; j += 1024) hist[j] = 0;
    __syncthreads();
    for (long e0 = tid; e0 < NK; e0 += 8 * 1024) {
        int dst[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) dst[u] = (e0 + u * 1024 < NK) ? ib[e0 + u * 1024] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (dst[u] >= 0) atomicAdd(&hist[dst[u]], 1);
    }
    __syncthreads();
    const int per = (N + 1023) / 1024;
    const int j0 = tid * per, j1 = min(N, j0 + per);
    int local = 0;
    for (int j = j0; j < j1; ++j) local += hist[j];
    part[tid] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - local;  // exclusive prefix of this thread's bins
    for (int j = j0; j < j1; ++j) {
        const int cnt = hist[j];
        rowptr[(long)b * (N + 1) + j] = run;
        hist[j] = run;
        run += cnt;
    }
    if (tid == 0) rowptr[(long)b * (N + 1) + N] = (int)NK;
    __syncthreads();
    for (long e0 = tid; e0 < NK; e0 += 8 * 1024) {
        int dst[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) dst[u] = (e0 + u * 1024 < NK) ? ib[e0 + u * 1024] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (dst[u] >= 0) {
                const long e = e0 + u * 1024;
                const int pos = atomicAdd(&hist[dst[u]], 1);
                const int i = (int)(e / k), s = (int)(e - (long)i * k);
                col[b * NK + pos] = (i << 6) | s;
            }
    }
}

// Multi-workgroup variant (the single-workgroup kernel above leaves 248 of 256 CUs idle for ~32 us per graph):
// G workgroups per cloud, each owning a contiguous slice of the edge list.
//   csr_count: LDS histogram of the slice -> cnt[b][g][N]
//   csr_scan : per cloud, rowptr = exclusive scan of sum_g cnt, and cnt[b][g][j] <- rowptr[j] + sum_{g' < g} cnt[b][g'][j]
//   csr_fill : LDS cursors start at cnt[b][g][:], every edge of the slice takes the next slot of its destination
// (bipartite form: NS source rows with k slots each point into N destinations; the kNN graph has NS == N)
__global__ __launch_bounds__(1024) void csr_count_kernel(const int32_t *__restrict__ idx, int NS, int N, int k, int G,
                                                          int32_t *__restrict__ cnt) {
    extern __shared__ int sh[];
    const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const long NK = (long)NS * k;
    const long e0 = NK * g / G, e1 = NK * (g + 1) / G;
    const int32_t *ib = idx + b * NK;
    for (int j = tid; j < N; j += 1024) sh[j] = 0;
    __syncthreads();
    for (long e = e0 + tid; e < e1; e += 1024) atomicAdd(&sh[ib[e]], 1);
    __syncthreads();
    int32_t *out = cnt + ((long)b * G + g) * N;
    for (int j = tid; j < N; j += 1024) out[j] = sh[j];
}

__global__ __launch_bounds__(1024) void csr_scan_kernel(int NS, int N, int k, int G, int32_t *__restrict__ cnt,
                                                         int32_t *__restrict__ rowptr) {
    __shared__ int part[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    int32_t *cb = cnt + (long)b * G * N;
    const int per = (N + 1023) / 1024;
    const int j0 = tid * per, j1 = min(N, j0 + per);
    int local = 0;
    for (int j = j0; j < j1; ++j)
        for (int g = 0; g < G; ++g) local += cb[(long)g * N + j];
    part[tid] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - local;
    for (int j = j0; j < j1; ++j) {
        rowptr[(long)b * (N + 1) + j] = run;
        for (int g = 0; g < G; ++g) {
            const int c = cb[(long)g * N + j];
            cb[(long)g * N + j] = run;
            run += c;
        }
    }
    if (tid == 0) rowptr[(long)b * (N + 1) + N] = NS * k;
}

__global__ __launch_bounds__(1024) void csr_fill_kernel(const int32_t *__restrict__ idx, int NS, int N, int k, int G,
                                                         const int32_t *__restrict__ cnt, int32_t *__restrict__ col) {
    extern __shared__ int sh[];
    const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const long NK = (long)NS * k;
    const long e0 = NK * g / G, e1 = NK * (g + 1) / G;
    const int32_t *ib = idx + b * NK;
    const int32_t *start = cnt + ((long)b * G + g) * N;
    for (int j = tid; j < N; j += 1024) sh[j] = start[j];
    __syncthreads();
    for (long e = e0 + tid; e < e1; e += 1024) {
        const int pos = atomicAdd(&sh[ib[e]], 1);
        const int i = (int)(e / k), sl = (int)(e - (long)i * k);
        col[b * NK + pos] = (i << 6) | sl;
    }
}

// csr_scan folded into csr_fill (N <= 8192: two LDS arrays of N ints): every workgroup of a cloud repeats the cloud's
// small scan itself -- 16 x 128 KB of L2 reads instead of a 17 us single-workgroup launch between count and fill.
// cnt is only read; the g == 0 workgroup writes rowptr.
__global__ __launch_bounds__(1024) void csr_scan_fill_kernel(const int32_t *__restrict__ idx, int NS, int N, int k, int G,
                                                              const int32_t *__restrict__ cnt, int32_t *__restrict__ rowptr,
                                                              int32_t *__restrict__ col) {
    extern __shared__ int sh[];   // [N] totals -> cursors, [N] counts of the lower slices
    __shared__ int wsum[16];
    int *lower = sh + N;
    const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const long NK = (long)NS * k;
    const int32_t *cb = cnt + (long)b * G * N;
    for (int j = tid; j < N; j += 1024) {
        int cv[FSG_CSR_SPLIT];   // G == FSG_CSR_SPLIT (launcher): all slice counts of a destination in flight at once -- with
#pragma unroll                   // a run-time bound the loop is one dependent L2 round trip per slice
        for (int gg = 0; gg < FSG_CSR_SPLIT; ++gg) cv[gg] = cb[(long)gg * N + j];
        int tot = 0, low = 0;
#pragma unroll
        for (int gg = 0; gg < FSG_CSR_SPLIT; ++gg) {
            tot += cv[gg];
            low += gg < g ? cv[gg] : 0;
        }
        sh[j] = tot;
        lower[j] = low;
    }
    __syncthreads();
    // exclusive scan of sh[0..N): contiguous chunk per thread, wave scan, 16 wave totals
    const int per = (N + 1023) / 1024;
    const int j0 = min(N, tid * per), j1 = min(N, j0 + per);
    int local = 0;
    for (int j = j0; j < j1; ++j) local += sh[j];
    int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = incl - local;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    int32_t *rp = rowptr + (long)b * (N + 1);
    for (int j = j0; j < j1; ++j) {
        const int c = sh[j];
        sh[j] = run + lower[j];
        if (g == 0) rp[j] = run;
        run += c;
    }
    if (g == 0 && tid == 0) rp[N] = NS * k;
    __syncthreads();
    const long e0 = NK * g / G, e1 = NK * (g + 1) / G;
    const int32_t *ib = idx + b * NK;
    for (long e = e0 + tid; e < e1; e += 1024) {
        const int pos = atomicAdd(&sh[ib[e]], 1);
        const int i = (int)(e / k), sl = (int)(e - (long)i * k);
        col[b * NK + pos] = (i << 6) | sl;
    }
}

// In-edge order: the slots of a destination are handed out by LDS atomics, i.e. in no particular order, and the backward
// sums its in-edges in slot order -- the gradients would differ in the last bits from run to run.  One wave per
// destination sorts its slice ascending by (source, slot) (rank by counting through an LDS copy), which makes the
// reverse graph, and with it the whole EdgeConv backward, reproducible.  Rows above CSR_SORT_CAP in-edges (hub points; a
// collapsed reconstruction in the Chamfer backward) do not fit the LDS copy: they are copied to `tmp` (NK ints per cloud, same
// offsets) and ranked from there -- O(deg^2 / 64) global loads for that one row; without `tmp` they stay as filled.
constexpr int CSR_SORT_CAP = 1024;
__global__ __launch_bounds__(256) void csr_sort_rows_kernel(const int32_t *__restrict__ rowptr, int32_t *__restrict__ col, int N,
                                                            int NK, int32_t *__restrict__ tmp) {
    __shared__ int32_t buf[4][CSR_SORT_CAP];
    const int b = blockIdx.y, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wave;
    if (j >= N) return;
    const int32_t *rp = rowptr + (long)b * (N + 1);
    const int beg = rp[j], deg = rp[j + 1] - beg;
    if (deg < 2) return;
    int32_t *row = col + (long)b * NK + beg;
    if (deg > CSR_SORT_CAP) {
        if (!tmp) return;
        int32_t *cp = tmp + (long)b * NK + beg;
        for (int t = lane; t < deg; t += 64) cp[t] = row[t];
        __threadfence_block();                       // the wave's own stores, re-read below by other lanes of the same wave
        __builtin_amdgcn_wave_barrier();
        for (int t0 = 0; t0 < deg; t0 += 64) {
            const int t = t0 + lane;
            const int32_t e = t < deg ? __builtin_nontemporal_load(cp + t) : 0x7fffffff;
            int rank = 0;
            for (int u = 0; u < deg; ++u) rank += __builtin_nontemporal_load(cp + u) < e ? 1 : 0;
            if (t < deg) row[rank] = e;
        }
        return;
    }
    int32_t *mine = buf[wave];
    for (int t = lane; t < deg; t += 64) mine[t] = row[t];
    __builtin_amdgcn_wave_barrier();
    for (int t0 = 0; t0 < deg; t0 += 64) {
        const int t = t0 + lane;
        const int32_t e = t < deg ? mine[t] : 0x7fffffff;
        int rank = 0;
#pragma unroll 8
        for (int u = 0; u < deg; ++u) rank += mine[u] < e ? 1 : 0;   // entries are distinct: (source << 6 | slot)
        if (t < deg) row[rank] = e;
    }
}

// ------------------------------------------------------------------ forward
// Round 4.  The SQ counters of round 3's kernel (lanes = channels, one 4-byte load per lane and row) show it VALU-bound: 4.7 M
// vector instructions per launch = 14 per gathered value, 59 % of the SIMD cycles at 4 cycles per instruction, no matrix work
// to hide behind.  This version has a lane own FOUR channels of ONE point: group g = lane / 16 works on point ibase + g, lane
// c16 = lane % 16 on channels 4 c16 .. 4 c16 + 3 of its rows -- a row is one 16-byte load per lane (a wave instruction reads
// four whole 256-byte rows, one per point), addresses and neighbour ids are computed once per FOUR values, sums and the
// BatchNorm moments run on packed fp32 instructions (v_pk_add_f32 / v_pk_fma_f32: two channels per instruction).  ~5.5
// instructions per value with the selection, ~2.5 without (SELECT = false: the statistics pass of a two-layer EdgeConv, whose
// selection belongs to its second layer).  Values and association are those of the first version: yy = y + q, tot summed in
// slot order, the first maximum of sgn yy wins.
template <bool SELECT>
__global__ __launch_bounds__(512) void ec1_stats_select_kernel(const float *__restrict__ pq,
                                                                const int32_t *__restrict__ idx,
                                                                const float *__restrict__ gamma, int N, int k, int Co,
                                                                int training, float *__restrict__ ysel,
                                                                uint8_t *__restrict__ arg, float *__restrict__ ssum,
                                                                float *__restrict__ partials) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    constexpr int NW = 4 * TPW;                        // waves
    __shared__ float red[2][4 * NW][64];               // (mean, M2) of every 16-lane group (= point) of the workgroup
    const int b = blockIdx.x, tile = blockIdx.y, cg = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c16 = lane & 15;
    const int c0 = cg * 64 + 4 * c16;                 // first of this lane's four channels
    const int ld = 2 * Co;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(pq + (long)b * N * ld), 0, __builtin_amdgcn_readfirstlane((int)((long)N * ld * 4)), 0x00020000);
    f2 sgn[2] = {{1.f, 1.f}, {1.f, 1.f}};
    if (SELECT) {
        const f4 gm = *reinterpret_cast<const f4 *>(gamma + c0);
        sgn[0] = f2{gm[0] >= 0.f ? 1.f : -1.f, gm[1] >= 0.f ? 1.f : -1.f};
        sgn[1] = f2{gm[2] >= 0.f ? 1.f : -1.f, gm[3] >= 0.f ? 1.f : -1.f};
    }
    const int i = tile * TP * TPW + wave * (TP / 4) + g;    // this group's point
    const bool live = i < N;
    const int ic = live ? i : N - 1;
    // the point's neighbour ids (lane c16 holds slots c16, c16 + 16, ...) and its Q row
    int ids[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (16 * r < k) ids[r] = (16 * r + c16 < k) ? idx[((long)b * N + ic) * k + 16 * r + c16] : 0;
    const f4 q4 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(
        rs, (unsigned)ic * (unsigned)(ld * 4) + (unsigned)((Co + c0) * 4), 0, 0));
    const f2 q[2] = {{q4[0], q4[1]}, {q4[2], q4[3]}};
    f2 tot[2] = {{0.f, 0.f}, {0.f, 0.f}}, shift[2] = {{0.f, 0.f}, {0.f, 0.f}}, s1[2] = {{0.f, 0.f}, {0.f, 0.f}},
       s2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int barg[4] = {0, 0, 0, 0};
    // rounds of RB neighbours: a round's loads are all requested before its first value is used (RB = 8: two register sets of
    // 32 keep the kernel at <= 128 VGPRs = two 512-thread workgroups per CU)
    constexpr int RB = SELECT ? 8 : 10;        // (without the selection's registers ten fit: k = 20 is ONE round trip of twenty loads)
    auto request = [&](int s0, u4 (&y)[RB]) {
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int s = min(s0 + u, k - 1);                         // uniform
            const int r = s >> 4;
            const int held = r == 0 ? ids[0] : (r == 1 ? ids[1] : (r == 2 ? ids[2] : ids[3]));
            const int j = __builtin_amdgcn_ds_bpermute(4 * ((lane & 48) | (s & 15)), held);   // lane (g, s % 16) holds slot s
            y[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)j * (unsigned)(ld * 4) + (unsigned)(c0 * 4), 0, 0);
        }
    };
    auto consume = [&](int s0, const u4 (&y)[RB]) {
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int s = s0 + u;
            if (s < k) {                                             // uniform
                const f4 yv = __builtin_bit_cast(f4, y[u]);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const f2 yy = f2{yv[2 * hf], yv[2 * hf + 1]} + q[hf];
                    tot[hf] += yy;
                    if (training) {
                        if (s == 0) shift[hf] = yy;
                        const f2 d = yy - shift[hf];
                        s1[hf] += d;
                        s2[hf] = __builtin_elementwise_fma(d, d, s2[hf]);
                    }
                    if (SELECT) {
                        const f2 v = sgn[hf] * yy;
#pragma unroll
                        for (int e = 0; e < 2; ++e)
                            if (v[e] > best[2 * hf + e]) { best[2 * hf + e] = v[e]; barg[2 * hf + e] = s; }
                    }
                }
            }
        }
    };
    u4 ya[RB], yb[RB];
    request(0, ya);
    for (int s0 = 0; s0 < k; s0 += 2 * RB) {
        if (s0 + RB < k) request(s0 + RB, yb);
        consume(s0, ya);
        if (s0 + 2 * RB < k) request(s0 + 2 * RB, ya);
        if (s0 + RB < k) consume(s0 + RB, yb);
    }
    if (live) {
        const long o = ((long)b * N + i) * Co + c0;
        if (SELECT) {
            *reinterpret_cast<f4 *>(ysel + o) = f4{sgn[0][0] * best[0], sgn[0][1] * best[1], sgn[1][0] * best[2], sgn[1][1] * best[3]};
            *reinterpret_cast<unsigned *>(arg + o) =
                (unsigned)barg[0] | ((unsigned)barg[1] << 8) | ((unsigned)barg[2] << 16) | ((unsigned)barg[3] << 24);
        }
        if (ssum) *reinterpret_cast<f4 *>(ssum + o) = f4{tot[0][0], tot[0][1], tot[1][0], tot[1][1]};
    }
    if (!training) return;
    // per-group (mean, M2) of the point's k values -> LDS; wave 0 merges the workgroup's 4 NW groups.  Every live group holds
    // exactly k values, so Chan's merge collapses to  mean = average of the group means,  M2 = sum M2_g + k sum (mean_g - mean)^2:
    // two passes of independent adds in a fixed order, ONE division (a chain of 31 pairwise merges with a division each was
    // ~3 us of one wave at the end of every workgroup).
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float s1e = s1[e >> 1][e & 1], s2e = s2[e >> 1][e & 1], she = shift[e >> 1][e & 1];
        red[0][wave * 4 + g][4 * c16 + e] = she + s1e / (float)k;
        red[1][wave * 4 + g][4 * c16 + e] = fmaxf(s2e - s1e * s1e / (float)k, 0.f);
    }
    __syncthreads();
    if (wave == 0) {
        const int c = cg * 64 + lane;
        const int live_groups = min(4 * NW, max(0, N - tile * TP * TPW));    // points of this workgroup inside the cloud
        float sm = 0.f, sM = 0.f;
#pragma unroll 8
        for (int w = 0; w < 4 * NW; ++w) {
            const bool on = w < live_groups;
            sm += on ? red[0][w][lane] : 0.f;
            sM += on ? red[1][w][lane] : 0.f;
        }
        const float mu = live_groups > 0 ? sm / (float)live_groups : 0.f;
        float dev = 0.f;
#pragma unroll 8
        for (int w = 0; w < 4 * NW; ++w) {
            const float d = red[0][w][lane] - mu;
            dev = __builtin_fmaf(w < live_groups ? d : 0.f, d, dev);
        }
        const long rec = (long)b * gridDim.y + tile;
        float *pr = partials + rec * 3 * Co;
        pr[c] = (float)live_groups * (float)k;
        pr[Co + c] = mu;
        pr[2 * Co + c] = sM + (float)k * dev;
    }
}

// BatchNorm statistics from R per-workgroup records (n, mean, M2) per channel, in fp64:
//   A = sum n, Bm = sum n*mean, Cm = sum (M2 + n*mean^2) over the records, then mean = Bm/A, M2 = Cm - Bm^2/A (fp64: the
//   subtraction is benign), invstd, running-statistics update.  (A 64-thread workgroup reading all records alone took
//   22 us for 1024 records: load latency, one record in flight per lane.)
constexpr int FIN_S = 16;   // (slices of the first version: still sizes the unused stage area behind the records)

// One launch, ONE memory round trip per thread (round 4): a 1024-thread workgroup owns EIGHT channels, 128 slices of threads
// stride over the records -- at the 512 records of a config-2 EdgeConv a thread reads four records (12 loads requested together),
// where the first version (64 channels per workgroup, 16 slices) walked 32 records in eight dependent batches: 7.5 us for a
// kernel that adds 393 KB, most of it L2 / Infinity-Cache round trips one behind the other, on ONE CU.  Eight workgroups (64
// channels) now run beside each other, each finalising its own channels: no second stage.  Sums in fp64, folded in LDS in a
// fixed order (reproducible).
constexpr int FIN_CH = 8, FIN_SL = 128;
__global__ __launch_bounds__(FIN_CH * FIN_SL) void bn_finalize_kernel(const float *__restrict__ partials, int R, int Co,
                                                                      float eps, float momentum,
                                                                      float *__restrict__ mean_out,
                                                                      float *__restrict__ invstd_out,
                                                                      float *__restrict__ running_mean,
                                                                      float *__restrict__ running_var) {
    __shared__ double red[3][FIN_SL][FIN_CH];
    const int ch = threadIdx.x & (FIN_CH - 1), sl = threadIdx.x / FIN_CH;
    const int c = blockIdx.x * FIN_CH + ch;
    double a = 0.0, bm = 0.0, cm = 0.0;
    for (int r0 = sl; r0 < R; r0 += 4 * FIN_SL) {       // four records per round: their loads go out together
        float nv[4], mv[4], qv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = min(r0 + u * FIN_SL, R - 1);
            const float *pr = partials + (long)r * 3 * Co + c;
            nv[u] = pr[0];
            mv[u] = pr[Co];
            qv[u] = pr[2 * Co];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (r0 + u * FIN_SL < R) {
                const double n = nv[u], mu = mv[u];
                a += n;
                bm += n * mu;
                cm += (double)qv[u] + n * mu * mu;
            }
    }
    red[0][sl][ch] = a;
    red[1][sl][ch] = bm;
    red[2][sl][ch] = cm;
    __syncthreads();
    // fold the 128 slices: 16 threads per channel take 8 slices each, then one thread per channel the 16 partial sums
    __shared__ double red2[3][16][FIN_CH];
    if (sl < 16) {
        double x0 = 0.0, x1 = 0.0, x2 = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x0 += red[0][8 * sl + u][ch];
            x1 += red[1][8 * sl + u][ch];
            x2 += red[2][8 * sl + u][ch];
        }
        red2[0][sl][ch] = x0;
        red2[1][sl][ch] = x1;
        red2[2][sl][ch] = x2;
    }
    __syncthreads();
    if (sl != 0) return;
    double n = 0.0;
    bm = 0.0;
    cm = 0.0;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
        n += red2[0][s2][ch];
        bm += red2[1][s2][ch];
        cm += red2[2][s2][ch];
    }
    const double mu = n > 0.0 ? bm / n : 0.0;
    double M2 = n > 0.0 ? cm - bm * mu : 0.0;
    M2 = M2 > 0.0 ? M2 : 0.0;
    const double var = n > 0.0 ? M2 / n : 0.0;
    mean_out[c] = (float)mu;
    invstd_out[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = n > 1.0 ? M2 / (n - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}

// out[b,c,i] = lrelu(gamma (ysel - mean) invstd + beta), point-major -> channel-major through LDS
__global__ __launch_bounds__(256) void ec1_apply_kernel(const float *__restrict__ ysel, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, const float *__restrict__ mean,
                                                         const float *__restrict__ invstd, int N, int Co, float slope,
                                                         float *__restrict__ out, float *__restrict__ out_pm) {
    __shared__ float tile[64][65];
    const int b = blockIdx.x, i0 = blockIdx.y * 64, cg = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = cg * 64 + lane;
    const float g = gamma[c] * invstd[c], sh = beta[c] - mean[c] * g;
    // the wave's sixteen rows are requested together (a load inside the bounds check of every iteration was one dependent
    // round trip per row: the whole 9 us of this kernel)
    float yv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) yv[u] = ysel[((long)b * N + min(i0 + wave + 4 * u, N - 1)) * Co + c];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int p = wave + 4 * u, i = i0 + p;
        float v = 0.f;
        if (i < N) {
            v = lrelu(__builtin_fmaf(yv[u], g, sh), slope);
            if (out_pm) out_pm[((long)b * N + i) * Co + c] = v;
        }
        tile[lane][p] = v;
    }
    __syncthreads();
    for (int cc = wave; cc < 64; cc += 4) {
        const int i = i0 + lane;
        if (i < N) out[((long)b * Co + cg * 64 + cc) * N + i] = tile[cc][lane];
    }
}

// The same pass with the next layer's graph build PREPARED on the way (csrc/knn_split.hip, fp16 form, 64 channels): the tile
// this workgroup has just produced is exactly what knn_split_prep_kernel would read back from HBM, so its products -- the
// oracle's squared norms, the centred scaled norms, the fp16 operand image; the point-major copy IS out_pm -- are emitted here
// and the graph build starts at its main kernel (one launch and ~11 us less per feature-space graph).  Centre and scale of
// the image come from the same fixed sample of 64 points (four runs of 16 at 0, N/4, N/2, 3N/4) that every workgroup of the
// cloud evaluates identically from ysel -- any centre / scale is correct, a representative one keeps the nominee lists short.
// PQ (round 4): the pass also emits the NEXT EdgeConv's per-point rows pq_next (B, N, 128) = x1 [W_rel | W_ctr - W_rel]^T for
// the 64 points whose activations it holds in LDS -- the product that used to be a vendor-library GEMM launch (7.3 us + a launch
// boundary per layer) rides here as 64 x 128 x 64 on v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fma chain per output,
// reproducible): wave w owns output columns 32 w .. 32 w + 31 for both 32-point halves, the weight is staged transposed in LDS.
typedef unsigned ec_u32x4 __attribute__((ext_vector_type(4)));
typedef float ec_f32x16 __attribute__((ext_vector_type(16)));
template <bool PQ>
__global__ __launch_bounds__(256) void ec1_apply_prep_kernel(const float *__restrict__ ysel, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, const float *__restrict__ mean,
                                                              const float *__restrict__ invstd, int N, float slope,
                                                              float *__restrict__ out, float *__restrict__ out_pm,
                                                              float *__restrict__ xx, float *__restrict__ xs,
                                                              ec_u32x4 *__restrict__ cand, float *__restrict__ cscale,
                                                              const float *__restrict__ w_next, float *__restrict__ pq_next) {
    constexpr int Co = 64, KS = 4, CN = 128;
    __shared__ float tile[64][65];
    __shared__ float red[4][64], mu[64], wmax[4];
    __shared__ float wt[PQ ? 64 : 1][PQ ? CN + 1 : 1];       // w_next transposed: wt[k][n] = w_next[n][k]
    const int b = blockIdx.x, i0 = blockIdx.y * 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane;
    if (PQ) {     // (requested first: the weight is not on this kernel's dependency chain)
        typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int e = threadIdx.x; e < CN * 16; e += 256) {
            const int n = e >> 4, k4 = 4 * (e & 15);
            const f4 v = *reinterpret_cast<const f4 *>(w_next + (long)n * Co + k4);
            wt[k4][n] = v[0]; wt[k4 + 1][n] = v[1]; wt[k4 + 2][n] = v[2]; wt[k4 + 3][n] = v[3];
        }
    }
    const float g = gamma[c] * invstd[c], sh = beta[c] - mean[c] * g;
    // the sample: wave w takes sample points 16 w .. 16 w + 15 (one of the four runs), lane = channel
    float sv[16], ssum = 0.f;
    {
        const int run0 = min((int)(((long)wave * N / 4) & ~15L), N - 16);
#pragma unroll
        for (int u = 0; u < 16; ++u) sv[u] = lrelu(__builtin_fmaf(ysel[((long)b * N + run0 + u) * Co + c], g, sh), slope);
#pragma unroll
        for (int u = 0; u < 16; ++u) ssum += sv[u];
        red[wave][lane] = ssum;
    }
    float yv[16];      // the wave's sixteen rows, requested together (see ec1_apply_kernel)
#pragma unroll
    for (int u = 0; u < 16; ++u) yv[u] = ysel[((long)b * N + min(i0 + wave + 4 * u, N - 1)) * Co + c];
#pragma unroll
    for (int u = 0; u < 16; ++u) {          // N % 64 == 0 (launch condition): every row of the tile exists
        const int p = wave + 4 * u, i = i0 + p;
        const float v = lrelu(__builtin_fmaf(yv[u], g, sh), slope);
        out_pm[((long)b * N + i) * Co + c] = v;
        tile[lane][p] = v;
    }
    __syncthreads();
    const float m = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) * (1.0f / 64.0f);
    float dv = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) dv = fmaxf(dv, fabsf(sv[u] - m));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dv = fmaxf(dv, __shfl_xor(dv, off));
    if (wave == 0) mu[lane] = m;
    if (lane == 0) wmax[wave] = dv;
#pragma unroll 4
    for (int cc = wave; cc < 64; cc += 4) out[((long)b * Co + cc) * N + i0 + lane] = tile[cc][lane];
    __syncthreads();
    if (PQ) {
        // pq_next tile: rows = the 64 points, columns 32 wave .. 32 wave + 31; A = tile (lane m = point, k half = lane / 32),
        // B = wt (lane n = column); D: lane (n, h) holds rows 8 (e / 4) + 4 h + e % 4
        const int mn = lane & 31, kk = lane >> 5;
        ec_f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll 8
        for (int k0 = 0; k0 < Co; k0 += 2) {
            const float a0 = tile[k0 + kk][mn], a1 = tile[k0 + kk][32 + mn], bb = wt[k0 + kk][32 * wave + mn];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bb, acc1, 0, 0, 0);
        }
        float *pr = pq_next + ((long)b * N + i0) * CN + 32 * wave + mn;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = 8 * (e >> 2) + 4 * kk + (e & 3);
            pr[(long)row * CN] = acc0[e];
            pr[(long)(32 + row) * CN] = acc1[e];
        }
    }
    const float maxdev = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    // power of two that maps the sample's largest deviation into [2^9, 2^10) (as knn_split_prep_kernel)
    int e2 = 9 - ((int)((__float_as_uint(maxdev) >> 23) & 255u) - 127);
    e2 = max(-100, min(100, e2));
    const float sigma = (maxdev > 0.f && maxdev < 3.0e38f) ? __uint_as_float((unsigned)(e2 + 127) << 23) : 1.f;
    if (blockIdx.y == 0 && threadIdx.x == 0) cscale[b] = sigma;
    if (threadIdx.x < 64) {
        // norms of point i0 + tid: the oracle's channel-ordered fma chain and the centred scaled one; a coordinate beyond 2^14
        // scaled units marks the point (NaN): its cloud takes the exact slow path
        const int pt = threadIdx.x;
        float a = 0.f, c2 = 0.f;
        bool bad = false;
#pragma unroll 8
        for (int ch = 0; ch < 64; ++ch) {
            const float v = tile[ch][pt];
            a = __builtin_fmaf(v, v, a);
            const float vs = (v - mu[ch]) * sigma;
            bad |= !(fabsf(vs) < 16384.0f);
            c2 = __builtin_fmaf(vs, vs, c2);
        }
        const bool in = i0 + pt < N;
        xx[(long)b * N + i0 + pt] = in ? a : INFINITY;
        xs[(long)b * N + i0 + pt] = in ? (bad ? __uint_as_float(0x7FC00000u) : c2) : INFINITY;
    }
    // fp16 operand image: two 32-point tiles x KS k-steps x 64 lanes of 16 bytes
    const long T = N / 32;
    for (int e = threadIdx.x; e < 2 * KS * 64; e += 256) {
        const int ln = e & 63, s2 = (e >> 6) % KS, half_tile = e / (KS * 64);
        const int mm = ln & 31, h = ln >> 5;
        unsigned cw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = 16 * s2 + 8 * h + i, pt = 32 * half_tile + mm;
            float v = 0.f;
            if (i0 + pt < N) v = (tile[ch][pt] - mu[ch]) * sigma;
            const _Float16 hv = (_Float16)v;
            cw[i] = (unsigned)__builtin_bit_cast(unsigned short, hv);
        }
        cand[(((long)b * T + (i0 / 32 + half_tile)) * KS + s2) * 64 + ln] =
            ec_u32x4{cw[0] | (cw[1] << 16), cw[2] | (cw[3] << 16), cw[4] | (cw[5] << 16), cw[6] | (cw[7] << 16)};
    }
}

// ------------------------------------------------------------------ backward
// h = grad_out * f'(u) on the selected edge (point-major), per-workgroup partial sums of h and h*yhat
// (the point-major gradient may arrive as up to two tensors with their own row strides -- e.g. one from the next layer and
// one slice of the gradient of the concatenated features: they are summed here instead of by an ATen add + a copy)
__global__ __launch_bounds__(256) void ec1_bwd_point_kernel(const float *__restrict__ gout, const float *__restrict__ gout_pm,
                                                             long ld_pm, const float *__restrict__ gout_pm2, long ld_pm2,
                                                             const float *__restrict__ ysel,
                                                             const float *__restrict__ gamma,
                                                             const float *__restrict__ beta,
                                                             const float *__restrict__ mean,
                                                             const float *__restrict__ invstd, int N, int Co, float slope,
                                                             float *__restrict__ h, float *__restrict__ partials) {
    __shared__ float tile[64][65];
    __shared__ float red[2][4][64];
    const int b = blockIdx.x, i0 = blockIdx.y * 64, cg = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int cc = wave; cc < 64; cc += 4) {
        const int i = i0 + lane;
        tile[cc][lane] = (gout && i < N) ? gout[((long)b * Co + cg * 64 + cc) * N + i] : 0.f;
    }
    __syncthreads();
    const int c = cg * 64 + lane;
    const float r = invstd[c], ga = gamma[c], be = beta[c], mu = mean[c];
    float sb = 0.f, sg = 0.f;
    // four rows per round with all their loads issued first (a loop that breaks at the cloud's end is one dependent round
    // trip per row: 16 in a row were the whole 12 us of this kernel); the sums keep the row order
    float ys[16], g1[16], g2[16];      // the wave's sixteen rows of every operand, requested together
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int i = i0 + wave + 16 * (q >> 2) + 4 * (q & 3);
        const bool ok = i < N;
        const long row = (long)b * N + (ok ? i : i0);
        ys[q] = ysel[row * Co + c];
        g1[q] = gout_pm ? gout_pm[row * ld_pm + c] : 0.f;
        g2[q] = gout_pm2 ? gout_pm2[row * ld_pm2 + c] : 0.f;
    }
    // two copies of the row loop: tiles inside the cloud (workgroup-uniform) run it without the per-row test and its branches
    auto rows = [&](auto checked) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {      // same row order as the loads: the sums keep their association
            const int p = wave + 16 * (q >> 2) + 4 * (q & 3), i = i0 + p;
            if (!decltype(checked)::value || i < N) {
                const long o = ((long)b * N + i) * Co + c;
                const float yhat = (ys[q] - mu) * r;
                const float u = __builtin_fmaf(ga, yhat, be);
                float gv = tile[lane][p];
                if (gout_pm) gv += g1[q];
                if (gout_pm2) gv += g2[q];
                const float hv = gv * (u > 0.f ? 1.f : slope);
                h[o] = hv;
                sb += hv;
                sg = __builtin_fmaf(hv, yhat, sg);
            }
        }
    };
    if (i0 + 64 <= N) rows(std::false_type{});
    else rows(std::true_type{});
    red[0][wave][lane] = sb;
    red[1][wave][lane] = sg;
    __syncthreads();
    if (wave == 0) {
        const long rec = (long)b * gridDim.y + blockIdx.y;
        float *pr = partials + rec * 2 * Co;
        pr[c] = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
        pr[Co + c] = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
    }
}

// sums R records of `nvec` vectors of Co floats in fp64: out[v][c].  Sixteen 64-lane slices per 64 channels stride over the
// records (with four slices a thread walked R/4 records one dependent L2 round trip after the other: 11 us at R = 1024),
// merged in slice order -- fixed order, reproducible
constexpr int SUMP_CH = 16, SUMP_SLICES = 64;   // round 4: 16 channels x 64 slices per workgroup (was 64 x 16): at 256 records a
// thread reads four of them in ONE batch, and four workgroups per vector run beside each other
__global__ __launch_bounds__(SUMP_CH * SUMP_SLICES) void sum_partials_kernel(const float *__restrict__ partials, int R, int Co,
                                                                             int nvec, float *__restrict__ out0,
                                                                             float *__restrict__ out1) {
    __shared__ double red[SUMP_SLICES][SUMP_CH];
    const int ch = threadIdx.x & (SUMP_CH - 1), slice = threadIdx.x / SUMP_CH;
    const int c = blockIdx.x * SUMP_CH + ch, v = blockIdx.y;
    double acc = 0.0;
    for (int r0 = slice; r0 < R; r0 += 4 * SUMP_SLICES) {
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = partials[((long)min(r0 + u * SUMP_SLICES, R - 1) * nvec + v) * Co + c];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (r0 + u * SUMP_SLICES < R) acc += t[u];
    }
    red[slice][ch] = acc;
    __syncthreads();
    if (slice < 8) {           // eight threads per channel fold eight slices each, the first folds the eight partial sums
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[8 * slice + w][ch];
        red[8 * slice][ch] = t;
    }
    __syncthreads();
    if (slice == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[8 * w][ch];
        (v == 0 ? out0 : out1)[c] = (float)t;
    }
}

// one wave per destination point j: dP_j (reverse-graph gather) and dQ_j.  Round 4 (the counters show the first version VALU-
// bound like the forward pass: 6.2 M vector instructions per launch, 56 % of the SIMD cycles): a lane owns FOUR channels, group
// g = lane / 16 takes the in-edges t = g (mod 4) of the destination -- one 16-byte load per lane and row piece, addresses once per
// four values, the Q sums on packed fp32 adds.  The four groups' partial sums meet through LDS in a fixed order
// ((g0 + g1) + (g2 + g3)) and the in-edges of a destination are sorted: reproducible bit for bit from run to run, as before.
__global__ __launch_bounds__(256) void ec1_bwd_gather_kernel(
    const float *__restrict__ pq, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ h, const uint8_t *__restrict__ arg, const float *__restrict__ ssum,
    const float *__restrict__ gamma, const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ dbeta, const float *__restrict__ dgamma, int N, int k, int Co, int training, float invM,
    float *__restrict__ grad_pq) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ float comb[4][2][4][64];        // [wave][ah | aq][group][channel]
    const int b = blockIdx.x, cg = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = blockIdx.y * 4 + wave;
    if (j >= N) return;                        // (wave-uniform; no workgroup barrier below)
    const int g = lane >> 4, c16 = lane & 15;
    const int c0 = cg * 64 + 4 * c16;
    const int ld = 2 * Co;
    const float *P = pq + (long)b * N * ld;
    const float *Q = P + Co;
    const float *hb = h + (long)b * N * Co;
    const uint8_t *ab = arg + (long)b * N * Co;
    const int beg = rowptr[(long)b * (N + 1) + j], end = rowptr[(long)b * (N + 1) + j + 1];
    const int32_t *cb = col + (long)b * N * k;
    float ah[4] = {0.f, 0.f, 0.f, 0.f};
    f2 aq[2] = {{0.f, 0.f}, {0.f, 0.f}};
    constexpr int RU = 5;                     // in-edges in flight per group (20 per wave and round trip)
    for (int t0 = beg; t0 < end; t0 += 64) {  // in-edges in chunks of 64: one coalesced load, then ds_bpermute broadcasts
        const int mye = (t0 + lane < end) ? cb[t0 + lane] : 0;
        const int cnt = min(64, end - t0);
        for (int t = 0; t < cnt; t += 4 * RU) {
            f4 hv[RU], qv[RU];
            unsigned av[RU];
            int sl[RU];
            bool ok[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int tt = t + 4 * u + g;
                ok[u] = tt < cnt;
                const int e = __builtin_amdgcn_ds_bpermute(4 * min(tt, cnt - 1), mye);
                const long o = (long)(e >> 6) * Co + c0;
                hv[u] = *reinterpret_cast<const f4 *>(hb + o);
                av[u] = *reinterpret_cast<const unsigned *>(ab + o);
                sl[u] = ok[u] ? (e & 63) : 255;                        // (no slot is 255: a masked edge never hits)
                qv[u] = training ? *reinterpret_cast<const f4 *>(Q + (long)(e >> 6) * ld + c0) : f4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) ah[e4] += ((int)((av[u] >> (8 * e4)) & 255u) == sl[u]) ? hv[u][e4] : 0.f;
                if (training) {
                    const float m = ok[u] ? 1.f : 0.f;
                    aq[0] += f2{qv[u][0], qv[u][1]} * f2{m, m};
                    aq[1] += f2{qv[u][2], qv[u][3]} * f2{m, m};
                }
            }
        }
    }
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
        comb[wave][0][g][4 * c16 + e4] = ah[e4];
        comb[wave][1][g][4 * c16 + e4] = aq[e4 >> 1][e4 & 1];
    }
    __builtin_amdgcn_wave_barrier();          // (a wave's own LDS operations complete in order)
    const int c = cg * 64 + lane;
    const float ahs = (comb[wave][0][0][lane] + comb[wave][0][1][lane]) + (comb[wave][0][2][lane] + comb[wave][0][3][lane]);
    const float aqs = (comb[wave][1][0][lane] + comb[wave][1][1][lane]) + (comb[wave][1][2][lane] + comb[wave][1][3][lane]);
    const float r = invstd[c], coef = r * gamma[c], mu = mean[c];
    const long oj = ((long)b * N + j) * Co + c;
    float dp = ahs, dq = h[oj];
    if (training) {
        const float db = dbeta[c] * invM, dg = dgamma[c] * invM * r;
        const float deg = (float)(end - beg);
        dp -= deg * db + dg * (deg * (P[(long)j * ld + c] - mu) + aqs);
        dq -= (float)k * db + dg * (ssum[oj] - (float)k * mu);
    }
    float *gp = grad_pq + ((long)b * N + j) * ld;
    gp[c] = coef * dp;
    gp[Co + c] = coef * dq;
}

}  // namespace

// launch helpers shared with edgeconv2.hip
int fsg_ec_stats1_records(int B, int N) { return B * fsg_cdiv(N, TP * TPW); }

int fsg_ec_stats1_launch(const float *pq, const int32_t *idx, const float *gamma, int B, int N, int k, int Co,
                         float *ysel, uint8_t *arg, float *ssum, float *partials, hipStream_t st) {
    FSG_REQUIRE((long)N * 2 * Co * 4 < (1L << 31), "edgeconv/stats: a cloud's [P | Q] rows must stay below 2 GiB (N=%d, Co=%d)", N, Co);
    // (statistics + per-point sums only: the selection of a two-layer EdgeConv belongs to its second layer; ysel / arg untouched)
    hipLaunchKernelGGL(ec1_stats_select_kernel<false>, dim3(B, fsg_cdiv(N, TP * TPW), Co / 64), dim3(256 * TPW), 0, st, pq, idx,
                       gamma, N, k, Co, 1, ysel, arg, ssum, partials);
    FSG_CHECK_LAUNCH("edgeconv/stats");
    return FSG_OK;
}

// `partials` holds R records of 3*Co floats FOLLOWED by the fp64 stage area (fsg_ec_finalize_stage_floats(Co) floats)
size_t fsg_ec_finalize_stage_floats(int Co) { return (size_t)FIN_S * 3 * Co * 2 + 2; }

int fsg_ec_finalize_launch(const float *partials, int R, int Co, float eps, float momentum, float *mean, float *invstd,
                           float *running_mean, float *running_var, hipStream_t st) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(Co / FIN_CH), dim3(FIN_CH * FIN_SL), 0, st, partials, R, Co, eps, momentum, mean, invstd,
                       running_mean, running_var);
    FSG_CHECK_LAUNCH("edgeconv/finalize");
    return FSG_OK;
}

int fsg_knn_split_ws_pointers(void *ws, size_t ws_bytes, int B, int N, int c_knn, float **xx, float **xs, void **cand,
                              float **cscale);       // knn_split.hip

// apply + the next graph build's prep (out_pm required, Co == 64, N % 64 == 0, N inside the coarse-sweep kernel's envelope)
int fsg_ec_apply_prep_launch(const float *ysel, const float *gamma, const float *beta, const float *mean, const float *invstd,
                             int B, int N, int Co, float slope, float *out, float *out_pm, void *knn_ws, size_t knn_ws_bytes,
                             const float *w_next, float *pq_next, hipStream_t st) {
    float *xx, *xs, *cscale;
    void *cand;
    if (Co != 64 || N % 64 != 0 || !out_pm || !out ||
        fsg_knn_split_ws_pointers(knn_ws, knn_ws_bytes, B, N, 64, &xx, &xs, &cand, &cscale) != FSG_OK) {
        fsg_set_error("fsg_edgeconv_apply_f32: the graph-build prep needs Co == 64, N %% 64 == 0, both output layouts and a "
                      "workspace of fsg_knn_dense_workspace_bytes(B, N, 64) bytes inside the coarse-sweep kernel's envelope");
        return FSG_ERR_ARG;
    }
    if (w_next)
        hipLaunchKernelGGL(ec1_apply_prep_kernel<true>, dim3(B, N / 64), dim3(256), 0, st, ysel, gamma, beta, mean, invstd, N,
                           slope, out, out_pm, xx, xs, reinterpret_cast<ec_u32x4 *>(cand), cscale, w_next, pq_next);
    else
        hipLaunchKernelGGL(ec1_apply_prep_kernel<false>, dim3(B, N / 64), dim3(256), 0, st, ysel, gamma, beta, mean, invstd, N,
                           slope, out, out_pm, xx, xs, reinterpret_cast<ec_u32x4 *>(cand), cscale, nullptr, nullptr);
    FSG_CHECK_LAUNCH("fsg_edgeconv_apply_f32/prep");
    return FSG_OK;
}

int fsg_ec_apply_launch(const float *ysel, const float *gamma, const float *beta, const float *mean, const float *invstd,
                        int B, int N, int Co, float slope, float *out, float *out_pm, hipStream_t st) {
    hipLaunchKernelGGL(ec1_apply_kernel, dim3(B, fsg_cdiv(N, 64), Co / 64), dim3(256), 0, st, ysel, gamma, beta, mean,
                       invstd, N, Co, slope, out, out_pm);
    FSG_CHECK_LAUNCH("edgeconv/apply");
    return FSG_OK;
}

int fsg_ec_bwd_point_launch(const float *gout, const float *gout_pm, long ld_pm, const float *gout_pm2, long ld_pm2,
                            const float *ysel, const float *gamma,
                            const float *beta, const float *mean, const float *invstd, int B, int N, int Co, float slope,
                            float *h, float *partials, float *dbeta, float *dgamma, hipStream_t st) {
    const int tiles64 = fsg_cdiv(N, 64);
    hipLaunchKernelGGL(ec1_bwd_point_kernel, dim3(B, tiles64, Co / 64), dim3(256), 0, st, gout, gout_pm, ld_pm, gout_pm2,
                       ld_pm2, ysel, gamma,
                       beta, mean, invstd, N, Co, slope, h, partials);
    FSG_CHECK_LAUNCH("edgeconv/bwd_point");
    hipLaunchKernelGGL(sum_partials_kernel, dim3(Co / SUMP_CH, 2), dim3(SUMP_CH * SUMP_SLICES), 0, st, partials, B * tiles64, Co, 2, dbeta,
                       dgamma);
    FSG_CHECK_LAUNCH("edgeconv/bwd_sum");
    return FSG_OK;
}

int fsg_ec_sum_launch(const float *partials, int R, int L, int nvec, float *out0, float *out1, hipStream_t st) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3(L / SUMP_CH, nvec), dim3(SUMP_CH * SUMP_SLICES), 0, st, partials, R, L, nvec, out0, out1);
    FSG_CHECK_LAUNCH("edgeconv/sum");
    return FSG_OK;
}

extern "C" size_t fsg_graph_reverse_csr_workspace_bytes(int B, int N, int k) {
    // per-slice counts of the multi-workgroup builder + one copy of the edge list for rows above the LDS sort capacity
    const size_t b = (size_t)(B > 0 ? B : 0), n = (size_t)(N > 0 ? N : 0);
    return sizeof(int32_t) * (b * FSG_CSR_SPLIT * n + b * n * (size_t)(k > 0 ? k : 0));
}

// shared with chamfer.hip: reverse of a bipartite graph (NS sources x k slots -> N destinations), multi-workgroup builder
int fsg_csr_bipartite_launch(const int32_t *idx, int B, int NS, int N, int k, int32_t *rowptr, int32_t *col, int32_t *cnt,
                             int32_t *tmp, hipStream_t st) {
    const int G = FSG_CSR_SPLIT;
    const size_t ldsN = sizeof(int) * (size_t)N;
    if (ldsN > 64 * 1024 || B > 65535) return FSG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(csr_count_kernel, dim3(B, G), dim3(1024), ldsN, st, idx, NS, N, k, G, cnt);
    FSG_CHECK_LAUNCH("fsg_graph_reverse_csr/count");
    if (2 * ldsN <= 64 * 1024) {
        hipLaunchKernelGGL(csr_scan_fill_kernel, dim3(B, G), dim3(1024), 2 * ldsN, st, idx, NS, N, k, G, cnt, rowptr, col);
        FSG_CHECK_LAUNCH("fsg_graph_reverse_csr/scan_fill");
    } else {
        hipLaunchKernelGGL(csr_scan_kernel, dim3(B), dim3(1024), 0, st, NS, N, k, G, cnt, rowptr);
        FSG_CHECK_LAUNCH("fsg_graph_reverse_csr/scan");
        hipLaunchKernelGGL(csr_fill_kernel, dim3(B, G), dim3(1024), ldsN, st, idx, NS, N, k, G, cnt, col);
        FSG_CHECK_LAUNCH("fsg_graph_reverse_csr/fill");
    }
    hipLaunchKernelGGL(csr_sort_rows_kernel, dim3(fsg_cdiv(N, 4), B), dim3(256), 0, st, rowptr, col, N, NS * k, tmp);
    FSG_CHECK_LAUNCH("fsg_graph_reverse_csr/sort");
    return FSG_OK;
}

extern "C" int fsg_graph_reverse_csr(const int32_t *idx, int B, int N, int k, int32_t *rowptr, int32_t *col,
                                     void *workspace, fsg_stream_t stream) {
    FSG_REQUIRE(idx && rowptr && col, "fsg_graph_reverse_csr: NULL pointer");
    FSG_REQUIRE(B >= 0 && N > 0 && k > 0 && k <= 64 && N <= 8192 * 4, "fsg_graph_reverse_csr: bad shape N=%d k=%d", N, k);
    if (B == 0) return FSG_OK;
    if (workspace) {
        int32_t *cnt = (int32_t *)workspace, *tmp = cnt + (size_t)B * FSG_CSR_SPLIT * N;
        const int rc = fsg_csr_bipartite_launch(idx, B, N, N, k, rowptr, col, cnt, tmp, (hipStream_t)stream);
        if (rc != FSG_ERR_UNSUPPORTED) return rc;
    }
    const size_t lds = sizeof(int) * ((size_t)N + 1024);
    static FsgLdsGrant grant;
    if (!grant.raise((const void *)csr_build_kernel, lds)) {
        fsg_set_error("fsg_graph_reverse_csr: cannot raise dynamic LDS to %zu", lds);
        return FSG_ERR_HIP;
    }
    hipLaunchKernelGGL(csr_build_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, idx, N, k, rowptr, col);
    FSG_CHECK_LAUNCH("fsg_graph_reverse_csr");
    hipLaunchKernelGGL(csr_sort_rows_kernel, dim3(fsg_cdiv(N, 4), B), dim3(256), 0, (hipStream_t)stream, rowptr, col, N, N * k,
                       (int32_t *)nullptr);
    FSG_CHECK_LAUNCH("fsg_graph_reverse_csr/sort");
    return FSG_OK;
}

// W = [W_rel | W_ctr] (Co, 2C)  ->  [W_rel ; W_ctr - W_rel] (2Co, C): the weight of the per-point GEMM that produces the
// [P | Q] rows, and its transpose rule for the gradient -- one tiny launch each (ATen: slice, subtract, cat = 2-4 launches)
__global__ __launch_bounds__(256) void edge_weights_fwd_kernel(const float *__restrict__ W, int Co, int C,
                                                               float *__restrict__ Wt) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 2 * Co * C) return;
    const int r = t / C, c = t - r * C;
    Wt[t] = r < Co ? W[(long)r * 2 * C + c] : W[(long)(r - Co) * 2 * C + C + c] - W[(long)(r - Co) * 2 * C + c];
}
__global__ __launch_bounds__(256) void edge_weights_bwd_kernel(const float *__restrict__ g, int Co, int C,
                                                               float *__restrict__ gW) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 2 * Co * C) return;
    const int r = t / (2 * C), c2 = t - r * 2 * C;
    gW[t] = c2 < C ? g[(long)r * C + c2] - g[(long)(Co + r) * C + c2] : g[(long)(Co + r) * C + (c2 - C)];
}

// all EdgeConv layers of a model at once (blockIdx.y = layer): the transforms depend on the weights only, so one launch at
// the head of the forward (and one at the tail of the backward) replaces one per layer
__global__ __launch_bounds__(256) void edge_weights_many_kernel(fsg_edge_weight_jobs jobs, int backward) {
    const int j = blockIdx.y;
    const int Co = jobs.Co[j], C = jobs.C[j];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 2 * Co * C) return;
    const float *src = jobs.src[j];
    float *dst = jobs.dst[j];
    if (!backward) {
        const int r = t / C, c = t - r * C;
        dst[t] = r < Co ? src[(long)r * 2 * C + c] : src[(long)(r - Co) * 2 * C + C + c] - src[(long)(r - Co) * 2 * C + c];
    } else {
        const int r = t / (2 * C), c2 = t - r * 2 * C;
        dst[t] = c2 < C ? src[(long)r * C + c2] - src[(long)(Co + r) * C + c2] : src[(long)(Co + r) * C + (c2 - C)];
    }
}

extern "C" int fsg_edge_weights_many_f32(const fsg_edge_weight_jobs *jobs, int backward, fsg_stream_t stream) {
    FSG_REQUIRE(jobs && jobs->n >= 1 && jobs->n <= FSG_EDGE_WEIGHT_MAX_JOBS, "fsg_edge_weights_many_f32: 1..%d jobs",
                FSG_EDGE_WEIGHT_MAX_JOBS);
    long most = 0;
    for (int j = 0; j < jobs->n; ++j) {
        FSG_REQUIRE(jobs->src[j] && jobs->dst[j] && jobs->Co[j] > 0 && jobs->C[j] > 0, "fsg_edge_weights_many_f32: bad job %d", j);
        const long e = 2L * jobs->Co[j] * jobs->C[j];
        most = e > most ? e : most;
    }
    hipLaunchKernelGGL(edge_weights_many_kernel, dim3(fsg_cdiv(most, 256), jobs->n), dim3(256), 0, (hipStream_t)stream, *jobs,
                       backward);
    FSG_CHECK_LAUNCH("fsg_edge_weights_many_f32");
    return FSG_OK;
}

extern "C" int fsg_edge_weights_fwd_f32(const float *W, int Co, int C, float *Wt, fsg_stream_t stream) {
    FSG_REQUIRE(W && Wt && Co > 0 && C > 0, "fsg_edge_weights_fwd_f32: bad arguments");
    hipLaunchKernelGGL(edge_weights_fwd_kernel, dim3(fsg_cdiv(2L * Co * C, 256)), dim3(256), 0, (hipStream_t)stream, W, Co, C, Wt);
    FSG_CHECK_LAUNCH("fsg_edge_weights_fwd_f32");
    return FSG_OK;
}
extern "C" int fsg_edge_weights_bwd_f32(const float *grad_Wt, int Co, int C, float *grad_W, fsg_stream_t stream) {
    FSG_REQUIRE(grad_Wt && grad_W && Co > 0 && C > 0, "fsg_edge_weights_bwd_f32: bad arguments");
    hipLaunchKernelGGL(edge_weights_bwd_kernel, dim3(fsg_cdiv(2L * Co * C, 256)), dim3(256), 0, (hipStream_t)stream, grad_Wt, Co,
                       C, grad_W);
    FSG_CHECK_LAUNCH("fsg_edge_weights_bwd_f32");
    return FSG_OK;
}

extern "C" size_t fsg_edgeconv1_workspace_bytes(int B, int N, int Co) {
    const size_t rec = (size_t)B * (size_t)fsg_cdiv(N, TP * TPW);
    return sizeof(float) * (rec * 3 * (size_t)Co + fsg_ec_finalize_stage_floats(Co));
}

extern "C" int fsg_edgeconv1_fwd_f32(const float *pq, const int32_t *idx, const float *gamma, const float *beta,
                                     float *running_mean, float *running_var, int B, int N, int k, int Co, int training,
                                     float momentum, float eps, float slope, float *out, float *out_pm, float *ysel,
                                     uint8_t *arg, float *ssum, float *mean, float *invstd, float *workspace,
                                     fsg_stream_t stream) {
    FSG_REQUIRE(pq && idx && gamma && beta && ysel && arg && mean && invstd, "fsg_edgeconv1_fwd_f32: NULL pointer");
    FSG_REQUIRE(B > 0 && N > 0 && k > 0 && k <= 64 && Co > 0 && Co % 64 == 0 && B <= 65535,
                "fsg_edgeconv1_fwd_f32: bad shape B=%d N=%d k=%d Co=%d (Co must be a multiple of 64)", B, N, k, Co);
    FSG_REQUIRE(!training || workspace, "fsg_edgeconv1_fwd_f32: training needs the workspace");
    FSG_REQUIRE((long)N * 2 * Co * 4 < (1L << 31), "fsg_edgeconv1_fwd_f32: a cloud's [P | Q] rows must stay below 2 GiB (N=%d, Co=%d)", N, Co);
    hipStream_t st = (hipStream_t)stream;
    const int tiles = fsg_cdiv(N, TP * TPW);
    hipLaunchKernelGGL(ec1_stats_select_kernel<true>, dim3(B, tiles, Co / 64), dim3(256 * TPW), 0, st, pq, idx, gamma, N, k, Co,
                       training, ysel, arg, ssum, workspace);
    FSG_CHECK_LAUNCH("fsg_edgeconv1_fwd_f32/stats");
    if (training) {
        const int rc = fsg_ec_finalize_launch(workspace, B * tiles, Co, eps, momentum, mean, invstd, running_mean,
                                              running_var, st);
        if (rc != FSG_OK) return rc;
    }
    if (!out) return FSG_OK;      // the caller applies BatchNorm + LeakyReLU itself (fsg_edgeconv_apply_f32)
    hipLaunchKernelGGL(ec1_apply_kernel, dim3(B, fsg_cdiv(N, 64), Co / 64), dim3(256), 0, st, ysel, gamma, beta, mean,
                       invstd, N, Co, slope, out, out_pm);
    FSG_CHECK_LAUNCH("fsg_edgeconv1_fwd_f32/apply");
    return FSG_OK;
}

// The last pass of the fused EdgeConv forward on its own: out (B,Co,N) [+ out_pm (B,N,Co)] = lrelu(BN(ysel)) from the selected
// pre-norm values -- for callers that ran fsg_edgeconv{1,2}_fwd_* with out == NULL.  knn_workspace != NULL (Co == 64,
// N % 64 == 0): the pass also PREPARES the feature-space graph build of the next layer (see ec1_apply_prep_kernel);
// fsg_knn_dense_prepared_f32 then starts at the main kernel.
extern "C" int fsg_edgeconv_apply_f32(const float *ysel, const float *gamma, const float *beta, const float *mean,
                                      const float *invstd, int B, int N, int Co, float slope, float *out, float *out_pm,
                                      void *knn_workspace, size_t knn_workspace_bytes, fsg_stream_t stream) {
    FSG_REQUIRE(ysel && gamma && beta && mean && invstd && out, "fsg_edgeconv_apply_f32: NULL pointer");
    FSG_REQUIRE(B > 0 && N > 0 && Co > 0 && Co % 64 == 0 && B <= 65535, "fsg_edgeconv_apply_f32: bad shape B=%d N=%d Co=%d", B, N, Co);
    if (knn_workspace)
        return fsg_ec_apply_prep_launch(ysel, gamma, beta, mean, invstd, B, N, Co, slope, out, out_pm, knn_workspace,
                                        knn_workspace_bytes, nullptr, nullptr, (hipStream_t)stream);
    return fsg_ec_apply_launch(ysel, gamma, beta, mean, invstd, B, N, Co, slope, out, out_pm, (hipStream_t)stream);
}

// fsg_edgeconv_apply_f32 with a knn_workspace, PLUS the per-point rows of the NEXT fused EdgeConv over this block's output:
// pq_next (B, N, 128) = out_pm w_next^T, w_next (128, 64) row-major = [W_rel ; W_ctr - W_rel] of the next block's first conv
// (fsg_edge_weights_many_f32) -- the plain GEMM of fsg_edgeconv1_fwd_f32's contract, computed here on the tile the pass holds in
// LDS (exact fp32 matrix instruction) instead of by a library launch.  Co == 64, N % 64 == 0, both layouts, next width 64.
extern "C" int fsg_edgeconv_apply_pq_f32(const float *ysel, const float *gamma, const float *beta, const float *mean,
                                         const float *invstd, int B, int N, int Co, float slope, float *out, float *out_pm,
                                         void *knn_workspace, size_t knn_workspace_bytes, const float *w_next, int rows_next,
                                         float *pq_next, fsg_stream_t stream) {
    FSG_REQUIRE(ysel && gamma && beta && mean && invstd && out && out_pm && knn_workspace && w_next && pq_next,
                "fsg_edgeconv_apply_pq_f32: NULL pointer");
    FSG_REQUIRE(B > 0 && N > 0 && Co == 64 && rows_next == 128 && B <= 65535,
                "fsg_edgeconv_apply_pq_f32: bad shape B=%d N=%d Co=%d rows_next=%d (Co must be 64, the next block 64 wide)", B, N, Co,
                rows_next);
    return fsg_ec_apply_prep_launch(ysel, gamma, beta, mean, invstd, B, N, Co, slope, out, out_pm, knn_workspace,
                                    knn_workspace_bytes, w_next, pq_next, (hipStream_t)stream);
}

extern "C" int fsg_edgeconv1_bwd_f32(const float *grad_out, const float *grad_out_pm, int64_t ld_pm, const float *grad_out_pm2,
                                     int64_t ld_pm2, const float *pq, const int32_t *rowptr, const int32_t *col,
                                     const float *gamma, const float *beta, const float *mean, const float *invstd,
                                     const float *ysel, const uint8_t *arg, const float *ssum, int B, int N, int k,
                                     int Co, int training, float slope, float *grad_pq, float *grad_gamma,
                                     float *grad_beta, float *h_scratch, float *workspace, fsg_stream_t stream) {
    FSG_REQUIRE((grad_out || grad_out_pm || grad_out_pm2) && pq && rowptr && col && gamma && beta && mean && invstd && ysel &&
                    arg && grad_pq && grad_gamma && grad_beta && h_scratch && workspace,
                "fsg_edgeconv1_bwd_f32: NULL pointer");
    FSG_REQUIRE((!grad_out_pm || ld_pm >= Co) && (!grad_out_pm2 || ld_pm2 >= Co),
                "fsg_edgeconv1_bwd_f32: row stride of a point-major gradient below Co=%d", Co);
    FSG_REQUIRE(!training || ssum, "fsg_edgeconv1_bwd_f32: training needs ssum");
    FSG_REQUIRE(B > 0 && N > 0 && k > 0 && k <= 64 && Co > 0 && Co % 64 == 0 && B <= 65535,
                "fsg_edgeconv1_bwd_f32: bad shape B=%d N=%d k=%d Co=%d", B, N, k, Co);
    hipStream_t st = (hipStream_t)stream;
    const int tiles64 = fsg_cdiv(N, 64);
    hipLaunchKernelGGL(ec1_bwd_point_kernel, dim3(B, tiles64, Co / 64), dim3(256), 0, st, grad_out, grad_out_pm, (long)ld_pm,
                       grad_out_pm2, (long)ld_pm2, ysel, gamma, beta, mean, invstd, N, Co, slope, h_scratch, workspace);
    FSG_CHECK_LAUNCH("fsg_edgeconv1_bwd_f32/point");
    hipLaunchKernelGGL(sum_partials_kernel, dim3(Co / SUMP_CH, 2), dim3(SUMP_CH * SUMP_SLICES), 0, st, workspace, B * tiles64, Co, 2, grad_beta,
                       grad_gamma);
    FSG_CHECK_LAUNCH("fsg_edgeconv1_bwd_f32/sum");
    const float invM = 1.0f / ((float)B * (float)N * (float)k);
    hipLaunchKernelGGL(ec1_bwd_gather_kernel, dim3(B, fsg_cdiv(N, 4), Co / 64), dim3(256), 0, st, pq, rowptr, col,
                       h_scratch, arg, ssum, gamma, mean, invstd, grad_beta, grad_gamma, N, k, Co, training, invM,
                       grad_pq);
    FSG_CHECK_LAUNCH("fsg_edgeconv1_bwd_f32/gather");
    return FSG_OK;
}

// Chamfer nearest neighbour + its gradient -- include/fsg_hip.h: fsg_chamfer_nn_f32 / _bwd_f32.
// Replaces the pytorch3d.loss.chamfer_distance call of losses/chamfer_loss.py:19.
//
// VALU-bound (B*N*M pair evaluations, ~8 flop each; inputs are a few hundred KB): two query points per lane, the other
// cloud streamed through LDS in 1024-point tiles (float4 per point: one broadcast ds_read_b128 per candidate) that the
// eight waves of a workgroup split between them.  d = fma(dz,dz, fma(dy,dy, dx*dx)) -- bit-exact with
// oracle/fsg_oracle.c.  (First version: one lane per query, 128-thread workgroups = 2 waves per CU at 8 x 4096 points:
// 122 us; the backward used atomicAdd = CAS loops: 123 us.)
#include "fsg_common.h"

namespace {

__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

constexpr int WAVES = 8;             // 512 threads
constexpr int QW = 128;              // queries per workgroup: two per lane
constexpr int TILE = 1024;           // candidates staged in LDS per sweep step; each wave scans TILE / WAVES of them

// One workgroup = 128 query points (two per lane, so a candidate read from LDS serves two distance updates) x eight
// waves that split every 1024-candidate tile between them; the eight partial (distance, index) minima of a query are
// merged at the end (lower index on equal distance, like the sequential scan of the oracle).
__global__ __launch_bounds__(WAVES * 64) void chamfer_nn_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                 int N, int M, float *__restrict__ dist,
                                                                 int32_t *__restrict__ arg) {
    __shared__ float4 ty[TILE];
    __shared__ float pd[WAVES][QW];
    __shared__ int pj[WAVES][QW];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i0 = blockIdx.x * QW + lane, i1 = i0 + 64;
    const float *yb = y + (long)b * M * 3;
    float ax = 0.f, ay = 0.f, az = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
    if (i0 < N) { const float *p = x + ((long)b * N + i0) * 3; ax = p[0]; ay = p[1]; az = p[2]; }
    if (i1 < N) { const float *p = x + ((long)b * N + i1) * 3; bx = p[0]; by = p[1]; bz = p[2]; }
    float best0 = INFINITY, best1 = INFINITY;
    int j0b = 0, j1b = 0;
    constexpr int PER = TILE / WAVES;
    for (int t0 = 0; t0 < M; t0 += TILE) {
        const int cnt = min(TILE, M - t0);
        __syncthreads();
        for (int t = tid; t < cnt; t += WAVES * 64) {
            const float *q = yb + (long)(t0 + t) * 3;
            ty[t] = make_float4(q[0], q[1], q[2], 0.f);
        }
        __syncthreads();
        const int lo = wave * PER, hi = min(cnt, lo + PER);
#pragma unroll 4
        for (int j = lo; j < hi; ++j) {
            const float4 c = ty[j];   // same address in every lane: LDS broadcast
            const float d0 = sqdist3(ax, ay, az, c.x, c.y, c.z), d1 = sqdist3(bx, by, bz, c.x, c.y, c.z);
            if (d0 < best0) { best0 = d0; j0b = t0 + j; }
            if (d1 < best1) { best1 = d1; j1b = t0 + j; }
        }
    }
    pd[wave][lane] = best0; pj[wave][lane] = j0b;
    pd[wave][lane + 64] = best1; pj[wave][lane + 64] = j1b;
    __syncthreads();
    if (tid < QW) {
        float bd = pd[0][tid];
        int bj = pj[0][tid];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            const float d = pd[w][tid];
            const int j = pj[w][tid];
            if (d < bd || (d == bd && j < bj)) { bd = d; bj = j; }
        }
        const int i = blockIdx.x * QW + tid;
        if (i < N) {
            dist[(long)b * N + i] = bd;
            arg[(long)b * N + i] = bj;
        }
    }
}

__global__ __launch_bounds__(256) void chamfer_bwd_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                           const int32_t *__restrict__ arg,
                                                           const float *__restrict__ g, int N, int M,
                                                           float *__restrict__ gx, float *__restrict__ gy) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool active = i < N;
    int a = 0;
    float v[3] = {0.f, 0.f, 0.f};
    if (active) {
        const long xi = ((long)b * N + i) * 3;
        a = arg[(long)b * N + i];
        const long ya = ((long)b * M + a) * 3;
        const float s = 2.0f * g[(long)b * N + i];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            v[d] = s * (x[xi + d] - y[ya + d]);
            gx[xi + d] += v[d];                // this thread is the only writer of its own point in this launch
        }
    }
    // Early in training the reconstruction is collapsed: thousands of points share a handful of nearest targets, and their
    // atomics serialise on one address (120 us for 8 x 4096 points).  Lanes of a wave that hit the same target are summed
    // first -- up to four leader rounds, stopping as soon as a round finds a small group -- and send ONE atomic per group.
    for (int round = 0; round < 4; ++round) {
        const unsigned long long act = __ballot(active);
        if (!act) break;
        const int leader = __ffsll((long long)act) - 1;
        const int ta = __builtin_amdgcn_readlane(a, leader);
        const bool same = active && a == ta;
        const int members = __popcll(__ballot(same));
        if (members < 4) break;                // spread-out targets: the plain atomics below are cheaper
        float sum[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float t = same ? v[d] : 0.f;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
            sum[d] = t;
        }
        if ((int)(threadIdx.x & 63) == leader) {
            const long ya = ((long)b * M + ta) * 3;
#pragma unroll
            for (int d = 0; d < 3; ++d) unsafeAtomicAdd(gy + ya + d, -sum[d]);
        }
        active = active && !same;
    }
    if (active) {
        const long ya = ((long)b * M + a) * 3;
#pragma unroll
        for (int d = 0; d < 3; ++d) unsafeAtomicAdd(gy + ya + d, -v[d]);   // hardware fp32 atomic (atomicAdd is a CAS loop)
    }
}

// deterministic variant: every target point sums its in-edges (the query points whose nearest neighbour it is) in
// ascending query order through the reverse graph of `arg` -- no atomics
__global__ __launch_bounds__(256) void chamfer_bwd_x_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                             const int32_t *__restrict__ arg, const float *__restrict__ g, int N,
                                                             int M, float *__restrict__ gx) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const long xi = ((long)b * N + i) * 3, ya = ((long)b * M + arg[(long)b * N + i]) * 3;
    const float s = 2.0f * g[(long)b * N + i];
#pragma unroll
    for (int d = 0; d < 3; ++d) gx[xi + d] += s * (x[xi + d] - y[ya + d]);
}

__global__ __launch_bounds__(256) void chamfer_bwd_y_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                             const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                             const float *__restrict__ g, int N, int M, float *__restrict__ gy) {
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    const int32_t *rp = rowptr + (long)b * (M + 1);
    const int32_t *cl = col + (long)b * N;
    const long yj = ((long)b * M + j) * 3;
    const float y0 = y[yj], y1 = y[yj + 1], y2 = y[yj + 2];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int t = rp[j]; t < rp[j + 1]; ++t) {
        const int i = cl[t] >> 6;
        const long xi = ((long)b * N + i) * 3;
        const float s = 2.0f * g[(long)b * N + i];
        a0 -= s * (x[xi] - y0);
        a1 -= s * (x[xi + 1] - y1);
        a2 -= s * (x[xi + 2] - y2);
    }
    gy[yj] += a0;
    gy[yj + 1] += a1;
    gy[yj + 2] += a2;
}

}  // namespace

int fsg_csr_bipartite_launch(const int32_t *idx, int B, int NS, int N, int k, int32_t *rowptr, int32_t *col, int32_t *cnt,
                             int32_t *tmp, hipStream_t st);   // edgeconv.hip

extern "C" size_t fsg_chamfer_nn_bwd_workspace_bytes(int B, int N, int M) {
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    return sizeof(int32_t) * ((size_t)B * (M + 1) + (size_t)B * N + (size_t)B * 16 * M + (size_t)B * N);   // + sort copy
}

extern "C" int fsg_chamfer_nn_f32(const float *x, const float *y, int B, int N, int M, float *dist, int32_t *arg,
                                  fsg_stream_t stream) {
    FSG_REQUIRE(x && y && dist && arg, "fsg_chamfer_nn_f32: NULL pointer");
    FSG_REQUIRE(B >= 0 && N > 0 && M > 0 && B <= 65535, "fsg_chamfer_nn_f32: bad shape B=%d N=%d M=%d", B, N, M);
    if (B == 0) return FSG_OK;
    hipLaunchKernelGGL(chamfer_nn_kernel, dim3(fsg_cdiv(N, QW), B), dim3(WAVES * 64), 0, (hipStream_t)stream, x, y, N, M,
                       dist, arg);
    FSG_CHECK_LAUNCH("fsg_chamfer_nn_f32");
    return FSG_OK;
}

extern "C" int fsg_chamfer_nn_bwd_f32(const float *x, const float *y, const int32_t *arg, const float *g_dist, int B,
                                      int N, int M, float *grad_x, float *grad_y, void *workspace, fsg_stream_t stream) {
    FSG_REQUIRE(x && y && arg && g_dist && grad_x && grad_y, "fsg_chamfer_nn_bwd_f32: NULL pointer");
    FSG_REQUIRE(B >= 0 && N > 0 && M > 0 && B <= 65535, "fsg_chamfer_nn_bwd_f32: bad shape");
    if (B == 0) return FSG_OK;
    if (workspace) {   // reproducible path: reverse graph of arg (N sources, one slot each -> M targets), ordered sums
        int32_t *rowptr = (int32_t *)workspace, *col = rowptr + (size_t)B * (M + 1), *cnt = col + (size_t)B * N;
        int32_t *tmp = cnt + (size_t)B * 16 * M;   // a collapsed reconstruction gives single targets thousands of in-edges
        const int rc = fsg_csr_bipartite_launch(arg, B, N, M, 1, rowptr, col, cnt, tmp, (hipStream_t)stream);
        if (rc == FSG_OK) {
            hipLaunchKernelGGL(chamfer_bwd_x_kernel, dim3(fsg_cdiv(N, 256), B), dim3(256), 0, (hipStream_t)stream, x, y, arg,
                               g_dist, N, M, grad_x);
            hipLaunchKernelGGL(chamfer_bwd_y_kernel, dim3(fsg_cdiv(M, 256), B), dim3(256), 0, (hipStream_t)stream, x, y, rowptr,
                               col, g_dist, N, M, grad_y);
            FSG_CHECK_LAUNCH("fsg_chamfer_nn_bwd_f32/ordered");
            return FSG_OK;
        }
        if (rc != FSG_ERR_UNSUPPORTED) return rc;
    }
    hipLaunchKernelGGL(chamfer_bwd_kernel, dim3(fsg_cdiv(N, 256), B), dim3(256), 0, (hipStream_t)stream, x, y, arg,
                       g_dist, N, M, grad_x, grad_y);
    FSG_CHECK_LAUNCH("fsg_chamfer_nn_bwd_f32");
    return FSG_OK;
}

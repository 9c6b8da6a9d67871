// Dense kNN graph build on the matrix cores -- the production path behind fsg_knn_dense_f32 for c_knn <= 128 and
// k + drop <= 48 (everything the reference's configurations use); knn_dense.hip keeps the general fallback.
//
// Structure (one 256-thread workgroup = 32 query points x 4 candidate quarters, one quarter per wave):
//   * distances: v_mfma_f32_32x32x2_f32 tiles, rows = 32 candidates, columns = 32 queries, K = channels two at a
//     time.  The instruction is an exact k-ordered fp32 fma chain, so dot(i,j) has the same bits as the oracle's
//     fmaf loop; d = (xx_i - 2 dot) + xx_j is finished on the VALU with xx from a pre-pass (same chain).
//     A lane owns ONE query (column l&31) and 16 of the tile's 32 candidates (rows (r&3)+8(r>>2)+4(l>>5)); its
//     partner lane l^32 owns the other 16.
//   * selection: every lane keeps a threshold tau (an upper bound of its query's k-th smallest key) and appends
//     candidates with key <= tau to its own column of an LDS buffer (bank = lane, conflict-free).  When a column
//     could overflow, all lanes sort their column with a 64-input bitonic network held in registers, keep the
//     k smallest (key, index) pairs and tighten tau.  tau is shared (min) with the partner lane and, through LDS,
//     with the three other waves that scan the same queries -- any lane's k-th smallest is a valid bound for the
//     query, so after the first compress almost nothing passes the filter any more.
//   * merge: the 8 sorted partial lists of a query (4 waves x 2 lane halves) are merged by one lane.
// Keys: distance bits mapped to an order-preserving uint32 (hi) and the candidate index (lo) in one uint64, so
// ties go to the lower index exactly like the oracle.
#include "fsg_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

constexpr int CAP = 64;  // LDS column depth per lane; k + drop + 16 <= CAP

__device__ __forceinline__ unsigned f2o(float d) {
    const unsigned u = __float_as_uint(d);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float o2f(unsigned k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

__global__ __launch_bounds__(256) void knn_sqnorm_kernel(const float *__restrict__ x, int N, long sb, long sc, int c_knn,
                                                          float *__restrict__ xx) {
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const float *xb = x + (long)b * sb;
    float a = 0.f;
    for (int c = 0; c < c_knn; ++c) a = __builtin_fmaf(xb[c * sc + j], xb[c * sc + j], a);
    xx[(long)b * N + j] = a;
}

// ascending bitonic sort of 64 uint64 held in registers (fully unrolled, static indices only)
__device__ __forceinline__ void sort64(u64 (&v)[CAP]) {
#pragma unroll
    for (int k = 2; k <= CAP; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int i = 0; i < CAP; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const u64 a = v[i], c = v[l];
                    const bool sw = up ? (a > c) : (a < c);
                    v[i] = sw ? c : a;
                    v[l] = sw ? a : c;
                }
            }
        }
    }
}

template <int KSTEPS>
__global__ __launch_bounds__(256) void knn_mfma_kernel(const float *__restrict__ x, const float *__restrict__ xx, int N,
                                                        long sb, long sc, int c_knn, int k, int flags,
                                                        int32_t *__restrict__ idx_out, float *__restrict__ dist_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *bkey_all = reinterpret_cast<unsigned *>(smem);                                   // [4][CAP][64]
    unsigned short *bidx_all = reinterpret_cast<unsigned short *>(smem + 4 * CAP * 64 * 4);    // [4][CAP][64]
    unsigned *tau_sh = reinterpret_cast<unsigned *>(smem + 4 * CAP * 64 * 6);                  // [4][32]
    int *cnt_sh = reinterpret_cast<int *>(tau_sh + 4 * 32);                                    // [4][64]

    const int b = blockIdx.y, q0 = blockIdx.x * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & 31, half = lane >> 5;
    const int q = q0 + ql;
    const float *xb = x + (long)b * sb;
    const float *xxb = xx + (long)b * N;
    unsigned *bkey = bkey_all + wave * CAP * 64;
    unsigned short *bidx = bidx_all + wave * CAP * 64;
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    const int KK = k + drop;

    if (lane < 32) tau_sh[wave * 32 + lane] = 0xFFFFFFFFu;
    __syncthreads();

    // B operand (queries), resident for the whole sweep: channel 2s + half of query ql
    float qreg[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        const int ch = 2 * s + half;
        qreg[s] = (ch < c_knn && q < N) ? xb[ch * sc + q] : 0.f;
    }
    const float xxq = q < N ? xxb[q] : 0.f;

    const int T = (N + 31) >> 5;  // candidate tiles of 32, split into 4 contiguous quarters
    const int t_beg = (int)((long)wave * T / 4), t_end = (int)((long)(wave + 1) * T / 4);
    unsigned tau = 0xFFFFFFFFu;
    int cnt = 0;

    auto compress = [&]() {
        u64 v[CAP];
#pragma unroll
        for (int s = 0; s < CAP; ++s)
            v[s] = (s < cnt) ? (((u64)bkey[s * 64 + lane] << 32) | bidx[s * 64 + lane]) : ~0ull;
        if (!(flags & 256)) sort64(v);
        const int had = cnt;
        cnt = min(cnt, KK);
#pragma unroll
        for (int s = 0; s < CAP - 16; ++s)
            if (s < cnt) {
                bkey[s * 64 + lane] = (unsigned)(v[s] >> 32);
                bidx[s * 64 + lane] = (unsigned short)(v[s] & 0xFFFFu);
            }
        if (had >= KK) {
            unsigned t = 0xFFFFFFFFu;
#pragma unroll
            for (int s = 0; s < CAP - 16; ++s)
                if (s == KK - 1) t = (unsigned)(v[s] >> 32);
            tau = min(tau, t);
        }
        tau = min(tau, (unsigned)__shfl_xor((int)tau, 32));   // partner lane scans the other 16 rows of the same query
        if (lane < 32) tau_sh[wave * 32 + lane] = tau;
    };

    for (int t = t_beg; t < t_end; ++t) {
        const int j0 = t << 5;
        if (__any(cnt > CAP - 16)) compress();
        // thresholds published by the waves scanning the other candidate quarters (stale values are still valid)
        tau = min(min(tau, tau_sh[ql]), min(tau_sh[32 + ql], min(tau_sh[64 + ql], tau_sh[96 + ql])));

        // ---- 32 candidates x 32 queries distance tile on the matrix core
        const int ja = j0 + ql;  // candidate whose channels this lane feeds as the A operand
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int ch = 2 * s + half;
            const float a = (ch < c_knn && ja < N) ? xb[ch * sc + ja] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qreg[s], acc, 0, 0, 0);
        }
        // ---- finish d = (xx_q - 2 dot) + xx_j and filter; rows of this lane: (r&3) + 8*(r>>2) + 4*half
        const bool diag = (flags & FSG_KNN_FIX_DIAG) && (j0 < q0 + 32) && (j0 + 32 > q0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int jr = j0 + 8 * g + 4 * half;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = jr + e;
                const float xj = j < N ? xxb[j] : 0.f;
                const float tt = xxq - 2.0f * acc[4 * g + e];
                float d = tt + xj;
                if (diag && j == q) d = 0.f;
                const unsigned key = f2o(d);
                if (j < N && key <= tau && !(flags & 1024)) {
                    bkey[cnt * 64 + lane] = key;
                    bidx[cnt * 64 + lane] = (unsigned short)j;
                    ++cnt;
                }
            }
        }
    }
    compress();  // leaves min(cnt, KK) entries sorted ascending in slots [0, cnt)
    cnt_sh[wave * 64 + lane] = cnt;
    __syncthreads();

    // ---- merge the 8 partial lists of each query (4 waves x 2 halves); one lane per query
    if (wave == 0 && lane < 32 && q < N && !(flags & 512)) {
        int head[8], len[8];
        u64 cur[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int w = p >> 1, col = ql + 32 * (p & 1);
            head[p] = 0;
            len[p] = cnt_sh[w * 64 + col];
            cur[p] = len[p] > 0 ? (((u64)bkey_all[w * CAP * 64 + col] << 32) | bidx_all[w * CAP * 64 + col]) : ~0ull;
        }
        for (int r = 0; r < KK; ++r) {
            u64 best = cur[0];
            int bp = 0;
#pragma unroll
            for (int p = 1; p < 8; ++p)
                if (cur[p] < best) { best = cur[p]; bp = p; }
            if (r >= drop) {
                const long o = ((long)b * N + q) * k + (r - drop);
                idx_out[o] = (int)(best & 0xFFFFu);
                if (dist_out) dist_out[o] = o2f((unsigned)(best >> 32));
            }
#pragma unroll
            for (int p = 0; p < 8; ++p)
                if (p == bp) {
                    const int w = p >> 1, col = ql + 32 * (p & 1);
                    const int h = ++head[p];
                    cur[p] = h < len[p] ? (((u64)bkey_all[(w * CAP + h) * 64 + col] << 32) | bidx_all[(w * CAP + h) * 64 + col])
                                        : ~0ull;
                }
        }
    }
}

}  // namespace

// returns FSG_ERR_UNSUPPORTED when the shape is outside this kernel's envelope (caller falls back)
int fsg_knn_mfma_launch(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn, int k, int flags,
                        int32_t *idx_out, float *dist_out, float *xx_scratch, hipStream_t st) {
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    if (c_knn > 128 || k + drop > CAP - 16 || N > 65535 || xx_scratch == nullptr) return FSG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(knn_sqnorm_kernel, dim3(fsg_cdiv(N, 256), B), dim3(256), 0, st, x, N, (long)stride_b,
                       (long)stride_c, c_knn, xx_scratch);
    FSG_CHECK_LAUNCH("fsg_knn_dense_f32/sqnorm");
    const size_t lds = 4 * CAP * 64 * 6 + 4 * 32 * 4 + 4 * 64 * 4;
    dim3 grid(fsg_cdiv(N, 32), B);
#define FSG_KNN_MFMA(KS)                                                                                              \
    do {                                                                                                              \
        static FsgLdsGrant grant;                                                                                     \
        if (!grant.raise((const void *)knn_mfma_kernel<KS>, (int)lds)) {                                              \
            fsg_set_error("fsg_knn_dense_f32: cannot raise dynamic LDS to %zu", lds);                                 \
            return FSG_ERR_HIP;                                                                                       \
        }                                                                                                             \
        hipLaunchKernelGGL(knn_mfma_kernel<KS>, grid, dim3(256), lds, st, x, xx_scratch, N, (long)stride_b,           \
                           (long)stride_c, c_knn, k, flags, idx_out, dist_out);                                       \
    } while (0)
    if (c_knn <= 4) FSG_KNN_MFMA(2);
    else if (c_knn <= 16) FSG_KNN_MFMA(8);
    else if (c_knn <= 64) FSG_KNN_MFMA(32);
    else FSG_KNN_MFMA(64);
#undef FSG_KNN_MFMA
    FSG_CHECK_LAUNCH("fsg_knn_dense_f32/mfma");
    return FSG_OK;
}

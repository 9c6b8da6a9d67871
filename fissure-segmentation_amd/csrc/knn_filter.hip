// Dense kNN graph build, "filter" design behind fsg_knn_dense_f32 (c_knn <= 128, k + drop <= 64).
//
// The two-phase kernel (knn_rows_mfma.hip) parks every distance of a 32 x 1024 block in LDS (131 KB: one workgroup per
// CU) and then runs an exact selection over all of them.  But once a query row holds K candidates, a new candidate can
// only matter if it beats the row's K-th best distance tau -- after n candidates that happens with probability K/n.
// Here the MFMA accumulators are compared against tau straight out of the matrix cores and only the survivors are
// appended to a small per-row list in LDS; nothing else is stored:
//   * a workgroup owns 32 queries (A operand resident in registers, two 16-row blocks) and sweeps the candidates in
//     rounds of WAVES x 16 (one 16-candidate MFMA tile per wave and round; the next tile's operand loads are in flight
//     while the current one is on the matrix cores);
//   * epilogue per tile: d = (xx_q - 2 dot) + xx_c (bit-identical to the oracle), `d <= tau[row]` -> LDS atomic on
//     the row's counter + one 8-byte store of (order-preserving distance key << 32 | index);
//   * when any row's list could overflow in the next round (and after the last round) the rows are merged: one wave per
//     row ranks the carried best list + survivors by counting ((distance, index) keys are distinct), keeps the K
//     smallest and publishes the new tau.  Round 1 runs with tau = +inf, so the first merge sees WAVES x 16 entries;
//     afterwards the expected number of survivors per row is K ln(N / (16 WAVES)) for the whole sweep.
// LDS: 16 KB best lists + 32 KB survivor lists -> three workgroups per CU, so one workgroup's merge overlaps the
// others' MFMA rounds.
// STATUS: exact (passes the kNN parity suite) but MEASURED SLOWER than the two-phase kernel on MI355X and therefore
// opt-in (flag 16384 / FSG_KNN_FILTER=1): B=8 N=2048 k=20: C=3 206 vs 59 us, C=64 228 vs 130 us; B=4 N=8192 k=40 C=64:
// 1083 vs 869 us.  In-kernel cycle counters (wave 0, C=64): tiles 140k cycles, barriers 5k, merges 165k -- the sweep
// starts with tau = +inf, so every row ranks ~270 entries over ~4 merges, and rank-by-counting costs ~70 cycles per
// entry when twelve waves per CU contend for LDS broadcasts: the same ~2000 vector operations per row as the two-phase
// selection, minus its coalesced ds_read_b128 row scans.  Kept as the record of that experiment.  Exactness: everything with d <= tau is kept (ties on the boundary included) and the final order
// is ascending (distance, index) like oracle/fsg_oracle.c; a stale (larger) tau only admits extra survivors.
#include "fsg_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int QB = 32;

__device__ __forceinline__ unsigned f2o(float d) {
    const unsigned u = __float_as_uint(d);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float o2f(unsigned k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

template <int KS, int WAVES, int CAP, int CK>
__global__ __launch_bounds__(WAVES * 64) void knn_filter_kernel(const float *__restrict__ x, const float *__restrict__ xx,
                                                                int N, long sb, long sc, int c_knn, int k, int flags,
                                                                int32_t *__restrict__ idx_out,
                                                                float *__restrict__ dist_out) {
    constexpr int R = WAVES * 16;                 // candidates per round
    constexpr int EPL = (CAP + CK + 63) / 64;     // merge: entries per lane
    static_assert(CAP >= 2 * R, "a round must fit behind the merge trigger");
    __shared__ u64 carry[QB * CK];                // best list per row, ascending
    __shared__ u64 slist[QB * CAP];               // survivors since the last merge
    __shared__ int scount[QB], ccount[QB];
    __shared__ float tauf[QB];
    __shared__ int need[3];   // merge-request flag of round r lives in need[r % 3] (reset two rounds ahead: no read/reset race)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int b = blockIdx.y, q0 = blockIdx.x * QB;
    const float *xb = x + (long)b * sb;
    const float *xxb = xx + (long)b * N;
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    const int KK = k + drop;
    const bool fix_diag = (flags & FSG_KNN_FIX_DIAG) != 0;

    if (tid < QB) { scount[tid] = 0; ccount[tid] = 0; tauf[tid] = INFINITY; }
    if (tid < 3) need[tid] = 0;

    float qa[2][KS];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int ch = 4 * s + l4, q = q0 + blk * 16 + l15;
            qa[blk][s] = (ch < c_knn && q < N) ? xb[ch * sc + q] : 0.f;
        }
    float xxq[2][4], tq[2][4];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = q0 + blk * 16 + l4 * 4 + e;
            xxq[blk][e] = q < N ? xxb[q] : 0.f;
            tq[blk][e] = q < N ? INFINITY : -INFINITY;   // rows beyond the cloud never accept
        }
    __syncthreads();

    const int ntile = (N + 15) >> 4;
    const int rounds = (ntile + WAVES - 1) / WAVES;
    float bn[KS], xn = 0.f;
    auto load_tile = [&](int t) {
        const int jc = t * 16 + l15;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int ch = 4 * s + l4;
            bn[s] = (ch < c_knn && jc < N) ? xb[ch * sc + jc] : 0.f;
        }
        xn = jc < N ? xxb[jc] : 0.f;
    };
    if (wave < ntile) load_tile(wave);

    for (int r = 0; r < rounds; ++r) {
        const int t = r * WAVES + wave;
        int *flag = &need[r % 3];
        if (tid == 0) need[(r + 1) % 3] = 0;   // last read in round r-2, separated from here by the barrier of round r-1
        if (t < ntile) {
            const int jc = t * 16 + l15;
            float bv[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) bv[s] = bn[s];
            const float xc = xn;
            if (t + WAVES < ntile) load_tile(t + WAVES);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[0][s], bv[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[1][s], bv[s], acc1, 0, 0, 0);
            }
            if (jc < N) {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const f32x4 acc = blk ? acc1 : acc0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int qr = blk * 16 + l4 * 4 + e;
                        const float tt = xxq[blk][e] - 2.0f * acc[e];
                        float d = tt + xc;
                        if (fix_diag && jc == q0 + qr) d = 0.f;
                        if (d <= tq[blk][e]) {
                            const int pos = atomicAdd(&scount[qr], 1);
                            slist[qr * CAP + pos] = ((u64)f2o(d) << 32) | (unsigned)jc;
                            if (pos >= CAP - R) *flag = 1;
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (*flag || r == rounds - 1) {   // uniform: the flag is final after the barrier
            for (int qi = wave; qi < QB; qi += WAVES) {
                const int sn = scount[qi];
                if (sn == 0) continue;
                const int cc = ccount[qi], total = cc + sn;
                const u64 *cl = carry + qi * CK, *sl = slist + qi * CAP;
                u64 ent[EPL];
                int rank[EPL];
#pragma unroll
                for (int u = 0; u < EPL; ++u) {
                    const int i = lane + 64 * u;
                    ent[u] = i < total ? (i < cc ? cl[i] : sl[i - cc]) : ~0ull;
                    rank[u] = 0;
                }
                const int nu = (total + 63) >> 6;   // lanes' slots actually in use (uniform)
                // rank by counting: every entry is read back as an LDS broadcast (same address in all lanes), eight reads
                // in flight per wait.  (Broadcasting out of registers with v_readlane measured slower: 130 vs 70 cycles per
                // entry, the 64-bit compares against an SGPR pair serialise.)
                if (nu <= 1) {
#pragma unroll 8
                    for (int t2 = 0; t2 < cc; ++t2) rank[0] += cl[t2] < ent[0] ? 1 : 0;
#pragma unroll 8
                    for (int t2 = 0; t2 < sn; ++t2) rank[0] += sl[t2] < ent[0] ? 1 : 0;
                } else {
#pragma unroll 8
                    for (int t2 = 0; t2 < cc; ++t2) {
                        const u64 v = cl[t2];
#pragma unroll
                        for (int u = 0; u < EPL; ++u) rank[u] += v < ent[u] ? 1 : 0;
                    }
#pragma unroll 8
                    for (int t2 = 0; t2 < sn; ++t2) {
                        const u64 v = sl[t2];
#pragma unroll
                        for (int u = 0; u < EPL; ++u) rank[u] += v < ent[u] ? 1 : 0;
                    }
                }
                __builtin_amdgcn_wave_barrier();   // every lane has read the old lists
                const int keep = total < KK ? total : KK;
#pragma unroll
                for (int u = 0; u < EPL; ++u) {
                    const int i = lane + 64 * u;
                    if (i < total && rank[u] < keep) {
                        carry[qi * CK + rank[u]] = ent[u];
                        if (rank[u] == KK - 1) tauf[qi] = o2f((unsigned)(ent[u] >> 32));   // K-th best: the new bound
                    }
                }
                if (lane == 0) { ccount[qi] = keep; scount[qi] = 0; }
            }
            __syncthreads();
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qr = blk * 16 + l4 * 4 + e;
                    if (q0 + qr < N) tq[blk][e] = tauf[qr];
                }
        }
    }
    // ---------------------------------------------------------------- output: rank r of row qi -> lane r
    for (int qi = wave; qi < QB; qi += WAVES) {
        const int q = q0 + qi;
        if (q >= N) break;
        if (lane >= drop && lane < KK) {
            const u64 v = carry[qi * CK + lane];
            const long o = ((long)b * N + q) * k + (lane - drop);
            idx_out[o] = (int)(unsigned)(v & 0xFFFFFFFFull);
            if (dist_out) dist_out[o] = o2f((unsigned)(v >> 32));
        }
    }
}

__global__ __launch_bounds__(256) void knn_sqnorm3_kernel(const float *__restrict__ x, int N, long sb, long sc, int c_knn,
                                                          float *__restrict__ xx) {
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const float *xb = x + (long)b * sb;
    float a = 0.f;
    int c = 0;
    for (; c + 8 <= c_knn; c += 8) {  // eight loads in flight, then the channel-ordered fma chain
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xb[(c + u) * sc + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) a = __builtin_fmaf(v[u], v[u], a);
    }
    for (; c < c_knn; ++c) a = __builtin_fmaf(xb[c * sc + j], xb[c * sc + j], a);
    xx[(long)b * N + j] = a;
}

}  // namespace

// returns FSG_ERR_UNSUPPORTED when the shape is outside this kernel's envelope (caller falls back)
int fsg_knn_filter_launch(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn, int k, int flags,
                          int32_t *idx_out, float *dist_out, float *xx_scratch, hipStream_t st) {
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    if (c_knn > 128 || k + drop > 64 || N > 65535 * 16 || xx_scratch == nullptr) return FSG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(knn_sqnorm3_kernel, dim3(fsg_cdiv(N, 256), B), dim3(256), 0, st, x, N, (long)stride_b, (long)stride_c,
                       c_knn, xx_scratch);
    FSG_CHECK_LAUNCH("fsg_knn_dense_f32/sqnorm");
    dim3 grid(fsg_cdiv(N, QB), B);
#define FSG_KNN_FL(KS, WV, CP, CKK)                                                                                      \
    hipLaunchKernelGGL((knn_filter_kernel<KS, WV, CP, CKK>), grid, dim3((WV) * 64), 0, st, x, xx_scratch, N, (long)stride_b, \
                       (long)stride_c, c_knn, k, flags, idx_out, dist_out)
    if (c_knn <= 4) FSG_KNN_FL(1, 4, 128, 64);
    else if (c_knn <= 16) FSG_KNN_FL(4, 4, 128, 64);
    else if (c_knn <= 64) FSG_KNN_FL(16, 4, 128, 64);
    else FSG_KNN_FL(32, 4, 128, 64);
#undef FSG_KNN_FL
    FSG_CHECK_LAUNCH("fsg_knn_dense_f32/filter");
    return FSG_OK;
}

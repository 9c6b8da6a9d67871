// Dense kNN graph build, wave-specialised pipeline (production path for c_knn <= 64 and k + drop <= 32, i.e. every DGCNN
// layer of the reference's configurations; knn_rows_mfma.hip covers the rest).
//
// Same arithmetic and selection rules as knn_rows_mfma.hip, different schedule.  A 1024-thread workgroup owns 32 query
// points and sweeps the candidates in chunks of 512 through a DOUBLE-BUFFERED LDS distance block (2 x 66 KB):
//   waves 0-3  (one per SIMD) are PRODUCERS: they compute chunk i+1 on the matrix cores (v_mfma_f32_16x16x4_f32, operands
//              prefetched one tile ahead) into buffer (i+1)&1;
//   waves 4-15 (three per SIMD) are CONSUMERS: exact top-k selection of chunk i from buffer i&1 (threshold from the K-th of the
//              128 lane minima by ballot bit-search, mbcnt compaction, rank-by-counting, carried best list);
//   one workgroup barrier per chunk hands the buffers over.  MFMA and VALU are separate pipes, so the producers' matrix work and
//   the consumers' vector work run concurrently on every SIMD instead of alternating as in the two-phase kernel.
#include "fsg_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int QB = 32;
constexpr int CH = 512;
constexpr int STRIDE = CH + 4;   // 4*STRIDE = 16 (mod 32): conflict-free accumulator stores
constexpr int VPL = CH / 64;     // 8 values per lane in the selection
constexpr int NPROD = 4, NCONS = 12;
constexpr int SURV = 128;        // survivor slots per consumer wave
constexpr int CK = 32;           // carried best list capacity (k + drop <= 32)

__device__ __forceinline__ unsigned f2o(float d) {
    const unsigned u = __float_as_uint(d);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float o2f(unsigned k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

template <int KS>
__global__ __launch_bounds__(1024) void knn_pipe_kernel(const float *__restrict__ x, const float *__restrict__ xx, int N,
                                                         long sb, long sc, int c_knn, int k, int flags,
                                                         int32_t *__restrict__ idx_out, float *__restrict__ dist_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *rows_all = reinterpret_cast<float *>(smem);                                        // [2][QB][STRIDE]
    u64 *carry = reinterpret_cast<u64 *>(smem + sizeof(float) * 2 * QB * STRIDE);             // [QB][CK]
    u64 *surv = carry + QB * CK;                                                              // [NCONS][SURV]
    int *ccount = reinterpret_cast<int *>(surv + NCONS * SURV);                               // [QB]
    // query operand of the MFMAs: in registers for few channels, in LDS ([2][KS][64] floats, lane-linear: conflict-free
    // ds_read_b32, one per MFMA) for 64 channels -- 1024 threads leave 128 VGPRs per lane
    constexpr bool QA_LDS = KS > 4;
    float *qa_lds = reinterpret_cast<float *>(ccount + QB);                                   // [2][KS][64] when QA_LDS

    const int b = blockIdx.y, q0 = blockIdx.x * QB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const float *xb = x + (long)b * sb;
    const float *xxb = xx + (long)b * N;
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    const int KK = k + drop;
    const bool fix_diag = (flags & FSG_KNN_FIX_DIAG) != 0;
    const bool producer = wave < NPROD;
    const int nchunk = (N + CH - 1) / CH;

    if (tid < QB) ccount[tid] = 0;

    // ---- producer state: query operand (A) of the two 16-row blocks, their squared norms, prefetched candidate operand
    float qa[2][QA_LDS ? 1 : KS], xxq[2][4], bn[KS], xn = 0.f;
    if (producer) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int ch = 4 * s + l4, q = q0 + blk * 16 + l15;
                const float qv = (ch < c_knn && q < N) ? xb[ch * sc + q] : 0.f;
                if (QA_LDS) {
                    if (wave == 0) qa_lds[(blk * KS + s) * 64 + lane] = qv;   // identical for every producer wave
                } else {
                    qa[blk][s] = qv;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int q = q0 + blk * 16 + l4 * 4 + e;
                xxq[blk][e] = q < N ? xxb[q] : 0.f;
            }
        }
    }
    auto load_tile = [&](int gt) {  // gt = global tile index (16 candidates each)
        const int jc = gt * 16 + l15;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int ch = 4 * s + l4;
            bn[s] = (ch < c_knn && jc < N) ? xb[ch * sc + jc] : 0.f;
        }
        xn = jc < N ? xxb[jc] : 0.f;
    };
    auto produce = [&](int chunk) {  // distance block of `chunk` into buffer chunk&1; tiles wave, wave+4, ... of the chunk
        float *rows = rows_all + (chunk & 1) * QB * STRIDE;
        const int c0 = chunk * CH;
        for (int t = wave; t < CH / 16; t += NPROD) {
            const int jc = c0 + t * 16 + l15;
            float bv[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) bv[s] = bn[s];
            const float xc = xn;
            // prefetch the next tile of this wave (possibly the first tile of the next chunk)
            int nt = t + NPROD, nc = chunk;
            if (nt >= CH / 16) { nt -= CH / 16; nc = chunk + 1; }
            if (nc < nchunk) load_tile(nc * (CH / 16) + nt);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const float a = QA_LDS ? qa_lds[(blk * KS + s) * 64 + lane] : qa[blk][QA_LDS ? 0 : s];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[s], acc, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qr = blk * 16 + l4 * 4 + e;
                    const float tt = xxq[blk][e] - 2.0f * acc[e];
                    float d = tt + xc;
                    if (fix_diag && jc == q0 + qr) d = 0.f;
                    if (jc >= N) d = INFINITY;
                    rows[qr * STRIDE + t * 16 + l15] = d;
                }
            }
        }
    };
    auto consume = [&](int chunk) {  // exact selection of `chunk` for the rows of this consumer wave
        const float *rows = rows_all + (chunk & 1) * QB * STRIDE;
        const int c0 = chunk * CH;
        const int cw = wave - NPROD;
        u64 *sv = surv + cw * SURV;
        for (int qi = cw; qi < QB; qi += NCONS) {
            if (q0 + qi >= N) break;
            const float *row = rows + qi * STRIDE;
            float v[VPL];
#pragma unroll
            for (int s = 0; s < VPL / 4; ++s) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(row + s * 256 + 4 * lane);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * s + e] = w[e];
            }
            const int cc = ccount[qi];
            unsigned tau;
            if (cc >= KK) {
                tau = (unsigned)(carry[qi * CK + KK - 1] >> 32);
            } else {
                float f1 = INFINITY, f2 = INFINITY;
#pragma unroll
                for (int e = 0; e < VPL; ++e) {
                    const float hi = fmaxf(v[e], f1);
                    f1 = fminf(v[e], f1);
                    f2 = fminf(f2, hi);
                }
                if (lane < cc) {
                    const float cv = o2f((unsigned)(carry[qi * CK + lane] >> 32));
                    const float hi = fmaxf(cv, f1);
                    f1 = fminf(cv, f1);
                    f2 = fminf(f2, hi);
                }
                const unsigned m1 = f2o(f1), m2 = f2o(f2);
                unsigned prefix = 0u;
#pragma unroll 4
                for (int bit = 31; bit >= 12; --bit) {
                    const unsigned t = prefix | ((1u << bit) - 1u);
                    const int c = __popcll(__ballot(m1 <= t)) + __popcll(__ballot(m2 <= t));
                    if (c < KK) prefix |= 1u << bit;
                }
                tau = prefix | 0xFFFu;
            }
            const float tau_f = tau >= 0xFF800000u ? INFINITY : o2f(tau);
            if (lane < cc) sv[lane] = carry[qi * CK + lane];
            int total = cc;
#pragma unroll
            for (int e = 0; e < VPL; ++e) {
                const int j = c0 + (e >> 2) * 256 + 4 * lane + (e & 3);
                const bool pass = v[e] <= tau_f && j < N;
                const u64 mask = __ballot(pass);
                if (mask) {
                    const int pos = total + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                           __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    if (pass && pos < SURV) sv[pos] = ((u64)f2o(v[e]) << 32) | (unsigned)j;
                    total += __popcll(mask);
                }
            }
            if (total <= SURV) {
                __builtin_amdgcn_wave_barrier();
                const u64 e0 = lane < total ? sv[lane] : ~0ull;
                const u64 e1 = (64 + lane) < total ? sv[64 + lane] : ~0ull;
                int r0 = 0, r1 = 0;
#pragma unroll 4
                for (int t = 0; t < total; ++t) {
                    const u64 xk = sv[t];
                    r0 += xk < e0 ? 1 : 0;
                    r1 += xk < e1 ? 1 : 0;
                }
                if (lane < total && r0 < KK) carry[qi * CK + r0] = e0;
                if ((64 + lane) < total && r1 < KK) carry[qi * CK + r1] = e1;
                __builtin_amdgcn_wave_barrier();
            } else {  // massive ties: K rounds of wave arg-min over the 8 row values + the carried entry of each lane
                unsigned taken = 0u;                                   // bit e: row value e consumed, bit 31: carried entry
                const u64 ck = lane < cc ? carry[qi * CK + lane] : ~0ull;
                __builtin_amdgcn_wave_barrier();
                for (int r = 0; r < KK; ++r) {
                    u64 best = (taken >> 31) ? ~0ull : ck;
#pragma unroll
                    for (int e = 0; e < VPL; ++e) {
                        const int j = c0 + (e >> 2) * 256 + 4 * lane + (e & 3);
                        const bool ok = v[e] <= tau_f && j < N && !((taken >> e) & 1u);
                        const u64 key = ok ? (((u64)f2o(v[e]) << 32) | (unsigned)j) : ~0ull;
                        best = key < best ? key : best;
                    }
                    const u64 mine = best;
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) {
                        const u64 o = __shfl_xor(best, off);
                        best = o < best ? o : best;
                    }
                    if (mine == best && best != ~0ull) {              // this lane owns the winner: mark it consumed
                        if (!(taken >> 31) && ck == best) taken |= 1u << 31;
#pragma unroll
                        for (int e = 0; e < VPL; ++e) {
                            const int j = c0 + (e >> 2) * 256 + 4 * lane + (e & 3);
                            if ((((u64)f2o(v[e]) << 32) | (unsigned)j) == best && v[e] <= tau_f && j < N) taken |= 1u << e;
                        }
                    }
                    if (lane == 0) carry[qi * CK + r] = best;
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (lane == 0) ccount[qi] = min(total, KK);
        }
    };

    // ---- pipeline: producers one chunk ahead of the consumers
    if (QA_LDS) __syncthreads();   // query operand written by wave 0
    if (producer) {
        load_tile(wave);
        produce(0);
    }
    __syncthreads();
    for (int i = 0; i < nchunk; ++i) {
        if (producer) {
            if (i + 1 < nchunk) produce(i + 1);
        } else {
            consume(i);
        }
        __syncthreads();
    }
    if (!producer) {
        const int cw = wave - NPROD;
        for (int qi = cw; qi < QB; qi += NCONS) {
            const int q = q0 + qi;
            if (q >= N) break;
            if (lane >= drop && lane < KK) {
                const u64 vv = carry[qi * CK + lane];
                const long o = ((long)b * N + q) * k + (lane - drop);
                idx_out[o] = (int)(unsigned)(vv & 0xFFFFFFFFull);
                if (dist_out) dist_out[o] = o2f((unsigned)(vv >> 32));
            }
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
    for (; c + 8 <= c_knn; c += 8) {
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
int fsg_knn_pipe_launch(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn, int k, int flags,
                        int32_t *idx_out, float *dist_out, float *xx_scratch, hipStream_t st) {
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    if (c_knn > 64 || k + drop > CK || xx_scratch == nullptr) return FSG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(knn_sqnorm3_kernel, dim3(fsg_cdiv(N, 256), B), dim3(256), 0, st, x, N, (long)stride_b,
                       (long)stride_c, c_knn, xx_scratch);
    FSG_CHECK_LAUNCH("fsg_knn_dense_f32/sqnorm");
    dim3 grid(fsg_cdiv(N, QB), B);
#define FSG_KNN_PIPE(KS)                                                                                              \
    do {                                                                                                              \
        const size_t lds = sizeof(float) * 2 * QB * STRIDE + sizeof(u64) * (QB * CK + NCONS * SURV) + sizeof(int) * QB + \
                           ((KS) > 4 ? sizeof(float) * 2 * (KS) * 64 : 0);                                            \
        static bool granted = false;                                                                                  \
        if (!granted) {                                                                                               \
            if (hipFuncSetAttribute((const void *)knn_pipe_kernel<KS>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                    (int)lds) != hipSuccess) {                                                        \
                fsg_set_error("fsg_knn_dense_f32: cannot raise dynamic LDS to %zu", lds);                             \
                return FSG_ERR_HIP;                                                                                   \
            }                                                                                                         \
            granted = true;                                                                                           \
        }                                                                                                             \
        hipLaunchKernelGGL(knn_pipe_kernel<KS>, grid, dim3(1024), lds, st, x, xx_scratch, N, (long)stride_b,          \
                           (long)stride_c, c_knn, k, flags, idx_out, dist_out);                                       \
    } while (0)
    if (c_knn <= 4) FSG_KNN_PIPE(1);
    else if (c_knn <= 16) FSG_KNN_PIPE(4);
    else FSG_KNN_PIPE(16);
#undef FSG_KNN_PIPE
    FSG_CHECK_LAUNCH("fsg_knn_dense_f32/pipe");
    return FSG_OK;
}

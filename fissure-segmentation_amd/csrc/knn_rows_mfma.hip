// Dense kNN graph build, production kernel behind fsg_knn_dense_f32 (c_knn <= 128, N <= 65535, k+drop <= 64).
//
// A 512-thread workgroup owns 32 query points of one cloud and sweeps the candidates in chunks of 1024:
//   phase A  the 32 x 1024 distance block of the chunk is produced on the matrix cores
//            (v_mfma_f32_16x16x4_f32: rows = queries, columns = 16 candidates, K = channels four at a time -- an
//            exact channel-ordered fp32 fma chain, bit-identical to the oracle's fmaf loop), finished as
//            d = (xx_q - 2 dot) + xx_c on the VALU and parked in LDS (131 KB, row stride 1028 floats so that the
//            accumulator stores are bank-conflict-free).  The (B,N,N) matrix never reaches HBM.
//   phase B  one wave per query row, wave-cooperative exact selection:
//            (1) every lane scans its 16 values (4 x ds_read_b128) and keeps its two smallest;
//            (2) the K-th smallest of these 128 lane minima (K = k + drop; bitonic sort across lanes) is an upper
//                bound tau of the row's K-th smallest distance, because at least K row values are <= it -- and it
//                is tight: about K .. 1.3 K row values survive `key <= tau`; later chunks reuse the K-th best
//                distance found so far, which is tighter still;
//            (3) survivors are compacted (wave prefix sum) behind the carried best list of the previous chunks and
//                sorted as (distance key, index) uint64 across the lanes; the first K are the new best list.
//            Rows with more than 128 survivors (massive ties) take a slow exact path (K rounds of wave arg-min).
// Keys: order-preserving uint32 image of the fp32 distance in the high word, candidate index in the low word, so
// ties go to the lower index exactly like oracle/fsg_oracle.c.
#include "fsg_common.h"

#ifdef FSG_KNN_STATS
__device__ unsigned long long fsg_knn_stats[8];
extern "C" int fsg_debug_knn_stats(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fsg_knn_stats), sizeof(fsg_knn_stats)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(fsg_knn_stats), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#define KSTAT(i, v) atomicAdd(&fsg_knn_stats[i], (unsigned long long)(v))
#define KSTATMAX(i, v) atomicMax(&fsg_knn_stats[i], (unsigned long long)(v))
#else
#define KSTAT(i, v) ((void)0)
#define KSTATMAX(i, v) ((void)0)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int QB = 32;            // queries per workgroup (two 16-row MFMA blocks)
// CH = candidates per chunk (template): 1024 -> 131 KB of distance rows, one workgroup per CU; 512 -> 66 KB, TWO workgroups
// per CU, so one workgroup's selection phase overlaps the other's MFMA phase.  LDS row stride CH + 4 floats:
// 4*STRIDE = 16 (mod 32) -> conflict-free accumulator stores.  CK = capacity of the carried best list (>= k + drop).

__device__ __forceinline__ unsigned f2o(float d) {
    const unsigned u = __float_as_uint(d);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float o2f(unsigned k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// inclusive prefix sum over the 64 lanes on the DPP network (Kogge-Stone inside each row of 16, then the row totals are
// broadcast down): 6 VALU steps; the shuffle version (ds_bpermute) paid ~100 cycles of LDS-crossbar latency per step
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
    (void)lane;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2 and 3
    return v;
}

// bitonic sort, ascending, of 128 values spread as element (slot*64 + lane); T = unsigned or u64
template <typename T>
__device__ __forceinline__ void sort128(T &v0, T &v1, int lane) {
#pragma unroll
    for (int kk = 2; kk <= 128; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1) {
            if (j == 64) {
                const T lo = v0 < v1 ? v0 : v1, hi = v0 < v1 ? v1 : v0;
                v0 = lo;
                v1 = hi;
            } else {
                const bool lower = (lane & j) == 0;
                const bool up0 = (lane & kk) == 0 || kk == 128;            // slot 0: index = lane
                const bool up1 = ((64 + lane) & kk) == 0 || kk == 128;     // slot 1: index = 64 + lane
                const T o0 = __shfl_xor(v0, j), o1 = __shfl_xor(v1, j);
                const T mn0 = v0 < o0 ? v0 : o0, mx0 = v0 < o0 ? o0 : v0;
                const T mn1 = v1 < o1 ? v1 : o1, mx1 = v1 < o1 ? o1 : v1;
                v0 = (lower == up0) ? mn0 : mx0;
                v1 = (lower == up1) ? mn1 : mx1;
            }
        }
    }
}

// bitonic sort, ascending, of 64 values, one per lane
template <typename T>
__device__ __forceinline__ void sort64(T &v, int lane) {
#pragma unroll
    for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1) {
            const bool lower = (lane & j) == 0;
            const bool up = (lane & kk) == 0 || kk == 64;
            const T o = __shfl_xor(v, j);
            const T mn = v < o ? v : o, mx = v < o ? o : v;
            v = (lower == up) ? mn : mx;
        }
    }
}

// WAVES waves per workgroup (16 when the register budget allows: everything here is latency-bound, so thread-level
// parallelism is the lever), SURV = survivor slots per row (they live in the row's own LDS storage once it is read)
// SEG = true: the packed-segment query of the PointTransformer path (fsg_knn_segment_f32): x = candidate coordinates
// (n,3), xq = query coordinates (m,3), cumulative segment ends in seg_c / seg_q; phase A evaluates the direct form
// fma(dz,dz, fma(dy,dy, dx*dx)) on the VALU (bit-identical to the oracle's orc_knn_segment_f32), phase B is shared.
struct SegArgs {
    const float *xq;
    const int32_t *seg_c, *seg_q;
    int nseg;
};

// QAL = true: the query (A) operand lives in LDS instead of 2*KS registers per lane and the two 16-query blocks run one
// after the other -- the register budget then allows 16 waves per workgroup at 64 channels (128 VGPRs), and both phases
// are latency-bound, i.e. scale with the number of resident waves.
// STREAM = true: only the FIRST chunk a workgroup visits -- the one that holds its own queries, so that spatially ordered
// clouds get their tightest bound at once -- goes through the LDS distance block and the full selection.  For every later
// chunk each row's K-th best distance tau is known and only ~K * CH / (candidates seen) of the CH new distances can beat it:
// they are compared against tau straight out of the accumulators (3 VALU operations per value, nothing parked in LDS),
// the few survivors are appended to the row's list (LDS atomic counter + one 8-byte store; the list lives in the row's
// idle distance storage) and one small rank-by-counting merge per row and chunk refreshes the best list and tau.  A
// chunk whose lists overflow (adversarial candidate order, massive ties) is simply redone through the distance block.
// Exactness is untouched: everything with d <= tau is kept and tau never drops below the final K-th distance.
template <int KS, int WAVES, int SURV, int CH, int CK, bool SEG = false, bool QAL = false, bool STREAM = false>
__global__ __launch_bounds__(WAVES * 64, (STREAM && WAVES == 8) ? 4 : 1) void knn_rows_mfma_kernel(const float *__restrict__ x, const float *__restrict__ xx,
                                                             int N, long sb, long sc, int c_knn, int k, int flags,
                                                             int32_t *__restrict__ idx_out,
                                                             float *__restrict__ dist_out, SegArgs sa) {
    constexpr int STRIDE = CH + 4;
    constexpr int VPL = CH / 64;      // values per lane in the selection phase
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *rows = reinterpret_cast<float *>(smem);                                        // [QB][STRIDE]
    u64 *carry = reinterpret_cast<u64 *>(smem + sizeof(float) * QB * STRIDE);             // [QB][CK] best list so far
    int *ccount = reinterpret_cast<int *>(carry + QB * CK);                               // [QB]
    // A operand copy (QAL only): [2 query halves][4*KS channels][16 queries].  Lane (l4, l15) reads channel 4s + l4,
    // query l15 of a half -> word 16 (4s + l4) + l15: the 64 lanes of a wave hit the 64 LDS banks exactly once.
    float *qal = reinterpret_cast<float *>(ccount + QB);
    int *lcount = reinterpret_cast<int *>(qal + (QAL ? 4 * KS * QB : 0));                 // [QB] survivors of a streamed chunk
    float *tauf = reinterpret_cast<float *>(lcount + QB);                                 // [QB] K-th best distance so far

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: scalar
    const int l15 = lane & 15, l4 = lane >> 4;
    int b = blockIdx.y, q0 = blockIdx.x * QB;
    if (!SEG && !(flags & 65536)) {   // flag 65536: plain placement (A/B timing)
        // XCD-aware placement: workgroups go to the 8 XCDs round-robin by linear id, so consecutive ids of one cloud
        // would spread every cloud over all eight L2s.  Renumber so that XCD x owns a contiguous 1/8 of the (cloud, tile)
        // space: with 8 clouds each L2 holds exactly one cloud's features.
        const unsigned L = blockIdx.x + gridDim.x * blockIdx.y, total = gridDim.x * gridDim.y;
        if ((total & 7u) == 0) {
            const unsigned V = (L & 7u) * (total >> 3) + (L >> 3);
            b = (int)(V / gridDim.x);
            q0 = (int)(V % gridDim.x) * QB;
        }
    }
    int NQ = N;             // queries of this cloud / segment
    long cbase = 0, qbase = 0;  // first candidate / query row of the segment (SEG)
    if (SEG) {
        // workgroup -> (segment, query tile): tiles are numbered segment by segment
        int t = blockIdx.x, sgm = 0, found = 0;
        for (; sgm < sa.nseg; ++sgm) {
            const int qs = sgm ? sa.seg_q[sgm - 1] : 0, qe = sa.seg_q[sgm];
            const int nt = (qe - qs + QB - 1) / QB;
            if (t < nt) { found = 1; break; }
            t -= nt;
        }
        if (!found) return;  // grid is an upper bound (m/QB + nseg tiles); uniform per workgroup, before any barrier
        b = sgm;
        q0 = t * QB;
        qbase = sgm ? sa.seg_q[sgm - 1] : 0;
        NQ = sa.seg_q[sgm] - (int)qbase;
        cbase = sgm ? sa.seg_c[sgm - 1] : 0;
        N = sa.seg_c[sgm] - (int)cbase;
    }
    const float *xb = SEG ? x + 3 * cbase : x + (long)b * sb;
    const float *xxb = SEG ? nullptr : xx + (long)b * N;
    const int drop = (!SEG && (flags & FSG_KNN_DROP_FIRST)) ? 1 : 0;
    const int KK = k + drop;
    const int LCAP = SURV - KK;   // STREAM: list slots per row behind the carried copy (the merge handles SURV entries)

    if (tid < QB) {
        ccount[tid] = 0;
        if (STREAM) { lcount[tid] = 0; tauf[tid] = INFINITY; }
    }

    // A operand: queries, rows l15 of the two 16-row blocks, channel 4s + l4; resident for the whole sweep
    float qa[QAL ? 1 : 2][QAL ? 1 : KS];
    if (QAL) {
        for (int t = tid; t < 4 * KS * QB; t += WAVES * 64) {
            const int ch = t / QB, qq = t % QB;
            qal[(qq >> 4) * (64 * KS) + ch * 16 + (qq & 15)] = (ch < c_knn && q0 + qq < N) ? xb[ch * sc + q0 + qq] : 0.f;
        }
        qa[0][0] = 0.f;
    } else {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int ch = 4 * s + l4, q = q0 + blk * 16 + l15;
                qa[QAL ? 0 : blk][QAL ? 0 : s] = (!SEG && ch < c_knn && q < N) ? xb[ch * sc + q] : 0.f;
            }
    }
    __shared__ float qsh[SEG ? QB * 3 : 1];   // SEG: coordinates of the 32 queries
    if (SEG) {
        if (tid < QB * 3) {
            const int q = q0 + tid / 3;
            qsh[tid] = q < NQ ? sa.xq[3 * (qbase + q) + tid % 3] : 0.f;
        }
    }
    // squared norms of the 4 accumulator rows of this lane: row = l4*4 + e
    float xxq[2][4];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = q0 + blk * 16 + l4 * 4 + e;
            xxq[blk][e] = (!SEG && q < N) ? xxb[q] : 0.f;
        }
    const bool fix_diag = (flags & FSG_KNN_FIX_DIAG) != 0;
    if (QAL) __syncthreads();   // A operand copy complete

    // rank-by-counting merge of one row (one wave): sv[0..cc) = the carried best list (ascending), sv[cc..total) = new
    // survivors in any order, total <= SURV; the KK smallest (distance key, index) entries land in carry[qi][rank]
    auto rank_merge = [&](const u64 *sv, int qi, int cc, int total) {
        __builtin_amdgcn_wave_barrier();
        // rank of every entry among the `total` entries (distinct keys: the index is part of the key); rank < K goes
        // to slot `rank`.  sv[0..cc) is the carried best list, ascending, so only the NEW survivors sv[cc..total) are
        // counted by looping (LDS broadcast reads); against the sorted prefix a carried entry knows its rank (its
        // position) and a survivor finds it with a 7-step binary search.
        auto prefix_rank = [&](int i, u64 e) {
            if (i < cc) return i;
            int lo = 0;
#pragma unroll
            for (int step = 64; step >= 1; step >>= 1) {
                const int m = lo + step;
                if (m <= cc && sv[m - 1] < e) lo = m;
            }
            return lo;
        };
        const u64 e0 = lane < total ? sv[lane] : ~0ull;
        int r0 = 0;
        if (SURV <= 64 || total <= 64) {   // the usual case: one entry per lane
#pragma unroll 8
            for (int t = cc; t < total; ++t) r0 += sv[t] < e0 ? 1 : 0;   // same address in every lane
            r0 += prefix_rank(lane, e0);
            if (lane < total && r0 < KK) carry[qi * CK + r0] = e0;
        } else {
            const u64 e1 = (64 + lane) < total ? sv[(64 + lane) % SURV] : ~0ull;
            int r1 = 0;
#pragma unroll 8
            for (int t = cc; t < total; ++t) {
                const u64 xk = sv[t];
                r0 += xk < e0 ? 1 : 0;
                r1 += xk < e1 ? 1 : 0;
            }
            r0 += prefix_rank(lane, e0);
            r1 += prefix_rank(64 + lane, e1);
            if (lane < total && r0 < KK) carry[qi * CK + r0] = e0;
            if ((64 + lane) < total && r1 < KK) carry[qi * CK + r1] = e1;
        }
        __builtin_amdgcn_wave_barrier();
    };

    // operand loads through a buffer resource over the c_knn channel rows of this cloud (dense mode)
    // operand loads through a buffer resource over the c_knn channel rows of this cloud: ONE per-lane byte offset
    // (channel l4, candidate col) + a scalar offset of four channel rows per MFMA step, instead of 16 per-lane
    // 64-bit pointers (32 VGPRs -- the compiler's choice for plain pointers); channels >= c_knn fall outside the
    // resource and read as 0, which is exactly the zero padding of the last K group.
    const uintptr_t xba = reinterpret_cast<uintptr_t>(xb);   // uniform by construction: tell the compiler so
    const unsigned xlo = __builtin_amdgcn_readfirstlane((unsigned)xba);
    const unsigned xhi = __builtin_amdgcn_readfirstlane((unsigned)(xba >> 32));
    // the resource ends with column N-1 of the LAST channel row: a read past it returns 0, every read before it
    // stays inside this cloud's own extent (with a channel stride > N, or the last cloud of a tensor, "the whole
    // row of every channel" would reach beyond the allocation); columns >= N of the earlier rows read whatever
    // lies between the rows -- those columns are overwritten with +inf below
    const int nrec = __builtin_amdgcn_readfirstlane((int)((((long)c_knn - 1) * sc + N) * 4));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void *>(((uintptr_t)xhi << 32) | xlo), 0, nrec, 0x00020000);

    const unsigned sc32f = (unsigned)sc;
    constexpr int TPC = CH / 16;          // tiles per chunk
    constexpr int TPW = TPC / WAVES;      // tiles per wave and chunk
    static_assert(SEG || TPW % 2 == 0, "tiles per wave must be even (two operand register sets)");
    const int nch = (N + CH - 1) / CH;
    // STREAM: the chunk that holds the workgroup's own queries goes first (ring order from there on)
    const int cd = (STREAM && nch > 0) ? min(q0 / CH, nch - 1) : 0;
    auto chunk_base = [&](int ring) { return ((cd + ring) % nch) * CH; };   // wave-uniform

    auto ld = [&](float (&bt)[KS], float &xt, unsigned col) {
        const unsigned vo = ((unsigned)l4 * sc32f + col) * 4u;
#pragma unroll
        for (int s = 0; s < KS; ++s)
            bt[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, (unsigned)(16 * s) * sc32f, 0));
        xt = xxb[min(col, (unsigned)(N - 1))];   // columns >= N: any finite value, the epilogue writes +inf
    };
    // products of one 16-candidate tile against the 32 queries: two independent accumulator chains, alternating
    auto mma = [&](const float (&bt)[KS], f32x4 &acc0, f32x4 &acc1) {
        if (QAL) {
            // the A operand is re-read from LDS for every tile ON PURPOSE (it would cost 2 KS registers): the offset
            // goes through an empty asm so that the reads are not hoisted out of the tile loop
            int qoff = l4 * 16 + l15;
            asm volatile("" : "+v"(qoff));
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qal[64 * s + qoff], bt[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qal[64 * KS + 64 * s + qoff], bt[s], acc1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[0][QAL ? 0 : s], bt[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[QAL ? 0 : 1][QAL ? 0 : s], bt[s], acc1, 0, 0, 0);
            }
        }
    };
    // tile tl of the chunk at c0: distances into the LDS block
    auto tile = [&](const float (&bt)[KS], float xc, int c0, int tl) {
        float *dst = rows + (l4 * 4) * STRIDE + tl * 16 + l15;
        if (c0 + tl * 16 >= N || (flags & 512)) {   // tile beyond the cloud (wave-uniform): nothing to compute
                                                    // (flag 512: timing ablation of phase A)
#pragma unroll
            for (int r = 0; r < 8; ++r) dst[((r >> 2) * 16 + (r & 3)) * STRIDE] = INFINITY;
            return;
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        mma(bt, acc0, acc1);
        const int drel = c0 + tl * 16 - q0;   // wave-uniform: columns drel .. drel+15 against the rows 0 .. 31
        if ((fix_diag && drel > -16 && drel < QB) || c0 + tl * 16 + 16 > N) {
            const int jc = c0 + tl * 16 + l15;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const f32x4 acc = blk ? acc1 : acc0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qr = blk * 16 + l4 * 4 + e;
                    const float tt = xxq[blk][e] - 2.0f * acc[e];
                    float d = tt + xc;
                    if (fix_diag && jc == q0 + qr) d = 0.f;
                    if (jc >= N) d = INFINITY;
                    dst[(blk * 16 + e) * STRIDE] = d;
                }
            }
        } else {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const f32x4 acc = blk ? acc1 : acc0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float tt = xxq[blk][e] - 2.0f * acc[e];
                    dst[(blk * 16 + e) * STRIDE] = tt + xc;
                }
            }
        }
    };
    // WIDE operand loads (two-phase chunks, N % 4 == 0): the 1024 loads of four bytes per lane that a workgroup issued per
    // chunk kept the texture path busy for as long as the matrix cores (in-kernel cycle stamps, tools/knn_phase_cycles.py:
    // phase A took 2.8x its MFMA time, and with the products switched off the loads alone cost the same again).  A wave
    // now owns 64 consecutive candidates per group: lane (l4, l15) loads channel 4s + l4 of the FOUR candidates
    // 4 l15 .. 4 l15 + 3 with one 16-byte load (full 256-byte row segments), and the four "virtual tiles" u = 0..3 (candidate
    // 4 l15 + u in lane l15) reuse the loaded registers -- a quarter of the load instructions.  No software pipelining: the
    // other three waves of the SIMD keep the matrix pipe busy while one waits for its loads.
    const bool wide = !SEG && (N & 3) == 0 && !(flags & 1048576);   // flag 1048576: the four-byte loads (A/B timing)
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(xxb), 0, SEG ? 0 : N * 4, 0x00020000);
    auto ld4 = [&](f32x4 (&bt)[KS], f32x4 &xt, unsigned col) {
        const unsigned vo = ((unsigned)l4 * sc32f + col) * 4u;
#pragma unroll
        for (int s = 0; s < KS; ++s)
            bt[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (unsigned)(16 * s) * sc32f, 0));
        xt = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, col * 4u, 0, 0));
    };
    // virtual tile u of the group at chunk column cb: this lane's candidate is column cb + 4 l15 + u of the chunk at c0
    auto gtile = [&](const float (&bt)[KS], float xc, int c0, int cb, int u) {
        const int colrel = cb + 4 * l15 + u;
        float *dst = rows + (l4 * 4) * STRIDE + colrel;
        if (c0 + cb >= N || (flags & 512)) {   // group beyond the cloud (wave-uniform) / timing ablation of phase A
#pragma unroll
            for (int r = 0; r < 8; ++r) dst[((r >> 2) * 16 + (r & 3)) * STRIDE] = INFINITY;
            return;
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        mma(bt, acc0, acc1);
        const int drel = c0 + cb - q0;   // wave-uniform: the group's columns drel .. drel+63 against the rows 0 .. 31
        if ((fix_diag && drel > -64 && drel < QB) || c0 + cb + 64 > N) {
            const int jc = c0 + colrel;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const f32x4 acc = blk ? acc1 : acc0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qr = blk * 16 + l4 * 4 + e;
                    const float tt = xxq[blk][e] - 2.0f * acc[e];
                    float d = tt + xc;
                    if (fix_diag && jc == q0 + qr) d = 0.f;
                    if (jc >= N) d = INFINITY;
                    dst[(blk * 16 + e) * STRIDE] = d;
                }
            }
        } else {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const f32x4 acc = blk ? acc1 : acc0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float tt = xxq[blk][e] - 2.0f * acc[e];
                    dst[(blk * 16 + e) * STRIDE] = tt + xc;
                }
            }
        }
    };
    // STREAM, epochs after the first chunk: the same products, but each distance is only compared with its row's tau;
    // survivors go to the row's list (slot = LDS atomic on the row's counter; entries beyond LCAP are dropped and the
    // epoch is redone through the distance block)
    float tq[2][4];
    auto ftile = [&](const float (&bt)[KS], float xc, int c0, int tl) {
        if (c0 + tl * 16 >= N) return;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        mma(bt, acc0, acc1);
        const int jc = c0 + tl * 16 + l15;
        const int drel = c0 + tl * 16 - q0;
        const bool special = (fix_diag && drel > -16 && drel < QB) || c0 + tl * 16 + 16 > N;   // wave-uniform
        float d[2][4];
        bool any = false;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const f32x4 acc = blk ? acc1 : acc0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float tt = xxq[blk][e] - 2.0f * acc[e];
                float dd = tt + xc;
                if (special) {
                    if (fix_diag && jc == q0 + blk * 16 + l4 * 4 + e) dd = 0.f;
                    if (jc >= N) dd = INFINITY;
                }
                d[blk][e] = dd;
                any |= dd <= tq[blk][e] && (!special || jc < N);
            }
        }
        if (__builtin_amdgcn_ballot_w64(any) != 0) {   // late in the sweep most tiles have no survivor at all
            int slot[2][4];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)     // all slot requests first (independent LDS atomics in flight together) ...
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    slot[blk][e] = LCAP;
                    if (d[blk][e] <= tq[blk][e] && (!special || jc < N)) slot[blk][e] = atomicAdd(&lcount[blk * 16 + l4 * 4 + e], 1);
                }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)     // ... then the stores
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (slot[blk][e] < LCAP)
                        reinterpret_cast<u64 *>(rows + (blk * 16 + l4 * 4 + e) * STRIDE)[KK + slot[blk][e]] =
                            ((u64)f2o(d[blk][e]) << 32) | (unsigned)jc;
        }
    };

    // Epochs.  Two-phase mode: one chunk per epoch, every chunk through the distance block (phase A) and the full
    // selection (phase B).  STREAM: after the first chunk the epochs double (as many chunks as were seen before), so
    // the expected number of survivors per row and epoch stays about K -- one small merge per row and epoch.
    int done = 0;
    while (done < nch) {
        const bool stream_ok = STREAM && !SEG && done > 0 && !(flags & 1024);   // flag 1024: every chunk through LDS
        // epoch length: as many chunks as were seen before (survivors per row ~ K), or more while the expected number of
        // survivors K * span / done stays well inside the list (factor 1.5 of headroom)
        const int span = stream_ok ? min(nch - done, done) : 1;
        bool redo = false;
        for (int attempt = 0; attempt < 2; ++attempt) {
            const bool fast = stream_ok && attempt == 0;
            const int nsub = fast ? 1 : span;      // the redo walks the epoch chunk by chunk
            for (int sub = 0; sub < nsub; ++sub) {
                const int c0 = chunk_base(done + sub);
                // ------------------------------------------------------------ phase A
                if (SEG) {
                    __syncthreads();  // qsh written (first chunk) / previous chunk's rows consumed
                    for (int cidx = tid; cidx < CH; cidx += WAVES * 64) {
                        const int jc = c0 + cidx;
                        float cx = 0.f, cy = 0.f, cz = 0.f;
                        if (jc < N) { cx = xb[3L * jc]; cy = xb[3L * jc + 1]; cz = xb[3L * jc + 2]; }
        #pragma unroll 8
                        for (int qr = 0; qr < QB; ++qr) {
                            const float dx = qsh[3 * qr] - cx, dy = qsh[3 * qr + 1] - cy, dz = qsh[3 * qr + 2] - cz;
                            const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                            rows[qr * STRIDE + cidx] = jc < N ? d : INFINITY;
                        }
                    }
                }
                // Phase A proper (every variant up to 64 channels): the wave's CH/16/WAVES tiles over two operand register sets
                // (no copies), unconditional loads off one per-lane offset, and the diagonal / out-of-range fix-ups only in the
                // tiles that can need them (wave-uniform tests).  The first version of this loop spent ~230 VALU issues per
                // tile on per-load predicates, 64-bit address pairs and operand copies against 32 MFMAs (rocprofv3:
                // SQ_INSTS_VALU 17.1 M vs SQ_INSTS_MFMA 2.1 M per launch at C=64).
                // (Requesting the next chunk's first tile before the end-of-chunk barrier was tried: the 17 registers held
                // across the barrier cost more in spills at 64 channels than the hidden latency returns.)

                if (!SEG) {
                    if (fast) {
#pragma unroll
                        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int qr = blk * 16 + l4 * 4 + e;
                                tq[blk][e] = q0 + qr < NQ ? tauf[qr] : -INFINITY;   // rows beyond the cloud never accept
                            }
                    }
                    // the wave's i-th tile of the epoch is tile T = wave + i * WAVES of the epoch's chunk sequence
                    const int ntile = (fast ? span : 1) * TPW;
                    auto where = [&](int i, int &cb, int &tl) {      // wave-uniform
                        const int T = wave + i * WAVES;
                        cb = fast ? chunk_base(done + T / TPC) : c0;
                        tl = T % TPC;
                    };
                    constexpr bool WIDE_OK = TPC % (4 * WAVES) == 0 && KS <= 16;   // 4 KS operand registers per lane
                    if (WIDE_OK && wide && !fast) {
                        // (Requesting the next chunk's operands before the selection phase -- 4 KS + 4 registers held across
                        // it -- was measured: the allocator spills, 226 vs 79 us.)
                        constexpr int GPW = WIDE_OK ? TPC / 4 / WAVES : 1;   // groups of 64 candidates per wave and chunk
#pragma unroll 1
                        for (int g = 0; g < GPW; ++g) {
                            const int cb = 64 * (wave + g * WAVES);
                            f32x4 b4[KS], x4;
                            ld4(b4, x4, (unsigned)(c0 + cb + 4 * l15));
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                float bu[KS];
#pragma unroll
                                for (int s2 = 0; s2 < KS; ++s2) bu[s2] = b4[s2][u];
                                gtile(bu, x4[u], c0, cb, u);
                            }
                        }
                    } else {
                    float b0[KS], b1[KS], x0, x1;
                    int cb0, tl0, cb1, tl1;
                    where(0, cb0, tl0);
                    ld(b0, x0, (unsigned)(cb0 + tl0 * 16 + l15));
#pragma unroll 1
                    for (int i = 0; i < ntile; i += 2) {
                        where(i + 1, cb1, tl1);
                        ld(b1, x1, (unsigned)(cb1 + tl1 * 16 + l15));
                        if (fast) ftile(b0, x0, cb0, tl0); else tile(b0, x0, cb0, tl0);
                        if (i + 2 < ntile) {
                            where(i + 2, cb0, tl0);
                            ld(b0, x0, (unsigned)(cb0 + tl0 * 16 + l15));
                        }
                        if (fast) ftile(b1, x1, cb1, tl1); else tile(b1, x1, cb1, tl1);
                    }
                    }
                }
                __syncthreads();

                if (fast) {
                    // -------------------------------------------------------- streamed epoch: small merge per row, or redo
                    const int mine_l = tid < QB ? lcount[tid] : 0;
                    redo = __syncthreads_or(mine_l > LCAP) != 0;
                    if (tid == 0) { KSTAT(0, 1); KSTAT(1, redo ? 1 : 0); KSTAT(5, span); }
                    if (tid < QB) { KSTAT(2, mine_l); KSTAT(3, 1); KSTATMAX(4, mine_l); }
                    if (redo) {
                        if (tid < QB) lcount[tid] = 0;
                        break;   // attempt 1 (the barrier above orders the list reads before the block is rewritten)
                    }
                    for (int qi = wave; qi < QB; qi += WAVES) {
                        if (q0 + qi >= NQ || (flags & 256)) break;
                        const int lc = lcount[qi];
                        if (lc > 0) {
                            u64 *sv = reinterpret_cast<u64 *>(rows + qi * STRIDE);
                            const int cc = ccount[qi];
                            // the survivors sit at sv[KK .. KK+lc), the carried copy goes to sv[0..cc); cc < KK (a first
                            // chunk with fewer than KK candidates followed by a short one) leaves a gap to close
                            u64 mv0 = 0, mv1 = 0;
                            if (cc < KK) {
                                if (lane < lc) mv0 = sv[KK + lane];
                                if (64 + lane < lc) mv1 = sv[KK + 64 + lane];
                                __builtin_amdgcn_wave_barrier();
                                if (lane < lc) sv[cc + lane] = mv0;
                                if (64 + lane < lc) sv[cc + 64 + lane] = mv1;
                            }
                            if (lane < cc) sv[lane] = carry[qi * CK + lane];
                            __builtin_amdgcn_wave_barrier();
                            const int total = cc + lc;
                            rank_merge(sv, qi, cc, total);
                            if (lane == 0) {
                                const int ncc = min(total, KK);
                                ccount[qi] = ncc;
                                lcount[qi] = 0;
                                if (ncc >= KK) tauf[qi] = o2f((unsigned)(carry[qi * CK + KK - 1] >> 32));
                            }
                        }
                    }
                    __syncthreads();
                    continue;
                }

                // (Two rows per wave at once -- two interleaved dependency chains -- was measured: 88.9 vs 90.4 us at N=2048,
                // 560 vs 543 us at N=8192 k=40: the selection is not bound by the latency of one chain.)
                // ---------------------------------------------------------------- phase B: exact selection, one wave per row
                for (int qi = wave; qi < QB; qi += WAVES) {
                    if (q0 + qi >= NQ || (flags & 256)) break;  // flag 256: timing ablation of phase B
                    const float *row = rows + qi * STRIDE;
                    float v[VPL];
        #pragma unroll
                    for (int s = 0; s < VPL / 4; ++s) {
                        const f32x4 w = *reinterpret_cast<const f32x4 *>(row + s * 256 + 4 * lane);
        #pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * s + e] = w[e];
                    }
                    // the row now lives in registers: its LDS storage (4 KB, 8-byte aligned) becomes the survivor buffer of this
                    // wave (up to SURV entries).  LDS serves a wave's instructions in order, so the reads above are ahead of every
                    // write below; the empty asm keeps the compiler from moving a store (different type: no alias assumed) up.
                    asm volatile("" ::: "memory");
                    u64 *sv = reinterpret_cast<u64 *>(rows + qi * STRIDE);
                    const int cc = ccount[qi];
                    unsigned tau;
                    if (cc >= KK) {
                        tau = (unsigned)(carry[qi * CK + KK - 1] >> 32);   // K-th best so far: tighter than any chunk estimate
                    } else {
                        // Upper bound of the row's K-th smallest value: the K-th smallest of a set of DISTINCT row elements.
                        // Carried lists of up to 32 entries (CK == 32): one minimum per lane (64 elements); longer lists: the two
                        // smallest of every lane (128 elements).  Binary search on the top 16 key bits, low bits rounded up.
                        constexpr bool TWO = CK > 32;
                        float f1 = INFINITY, f2 = INFINITY;
        #pragma unroll
                        for (int e = 0; e < VPL; ++e) {
                            if (TWO) {
                                const float hi = fmaxf(v[e], f1);
                                f2 = fminf(f2, hi);
                            }
                            f1 = fminf(v[e], f1);
                        }
                        if (lane < cc) {  // carried entries (fewer than K) also count as candidates of the bound
                            const float cv = o2f((unsigned)(carry[qi * CK + lane] >> 32));
                            const float hi = fmaxf(cv, f1);
                            f1 = fminf(cv, f1);
                            f2 = fminf(f2, hi);
                        }
                        const unsigned m1 = f2o(f1), m2 = f2o(f2);
                        unsigned prefix = 0u;
        #pragma unroll 4
                        for (int bit = 31; bit >= 16; --bit) {   // sign, exponent and 7 mantissa bits: the bound is within 0.8 %
                            const unsigned t = prefix | ((1u << bit) - 1u);
                            int c = __popcll(__ballot(m1 <= t));
                            if (TWO) c += __popcll(__ballot(m2 <= t));
                            if (c < KK) prefix |= 1u << bit;
                        }
                        tau = prefix | 0xFFFFu;
                    }
                    // columns >= N hold +inf in the block, so `v <= tau` rejects them by itself unless tau is +inf (fewer than K
                    // candidates so far): then tau is replaced by the largest finite value and +inf entries never pass
                    const float tau_f = tau >= 0x7F800000u + 0x80000000u ? 3.4028234e38f : o2f(tau);
                    // compact the survivors behind the carried list: per-lane count, wave prefix sum, per-lane stores
                    if (lane < cc) sv[lane] = carry[qi * CK + lane];
                    int mine = 0;
        #pragma unroll
                    for (int e = 0; e < VPL; ++e) mine += v[e] <= tau_f ? 1 : 0;
                    const int incl = wave_incl_scan(mine, lane);
                    const int total = cc + __builtin_amdgcn_readlane(incl, 63);
                    if (total <= SURV) {
                        int pos = cc + incl - mine;
        #pragma unroll
                        for (int e = 0; e < VPL; ++e) {
                            const int j = c0 + (e >> 2) * 256 + 4 * lane + (e & 3);
                            if (v[e] <= tau_f) sv[pos++] = ((u64)f2o(v[e]) << 32) | (unsigned)j;
                        }
                    }
                    if (total <= SURV) {
                        rank_merge(sv, qi, cc, total);
                    } else {
                        // slow exact path (massive ties): KK rounds of wave arg-min over the 16 row values + carried entry
                        u64 mykeys[VPL + 1];
        #pragma unroll
                        for (int e = 0; e < VPL; ++e) {
                            const int j = c0 + (e >> 2) * 256 + 4 * lane + (e & 3);
                            mykeys[e] = v[e] <= tau_f ? (((u64)f2o(v[e]) << 32) | (unsigned)j) : ~0ull;
                        }
                        mykeys[VPL] = lane < cc ? carry[qi * CK + lane] : ~0ull;
                        for (int r = 0; r < KK; ++r) {
                            u64 best = mykeys[0];
        #pragma unroll
                            for (int e = 1; e <= VPL; ++e) best = mykeys[e] < best ? mykeys[e] : best;
        #pragma unroll
                            for (int off = 32; off >= 1; off >>= 1) {
                                const u64 o = __shfl_xor(best, off);
                                best = o < best ? o : best;
                            }
        #pragma unroll
                            for (int e = 0; e <= VPL; ++e) mykeys[e] = mykeys[e] == best ? ~0ull : mykeys[e];
                            if (lane == 0) carry[qi * CK + r] = best;
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) {
                        const int ncc = min(total, KK);
                        ccount[qi] = ncc;
                        if (STREAM && ncc >= KK) tauf[qi] = o2f((unsigned)(carry[qi * CK + KK - 1] >> 32));
                    }
                }
                __syncthreads();  // rows are rewritten by the next chunk; carry/ccount visible
            }
            if (!redo || attempt == 1) break;
        }
        done += span;
    }
    // ---------------------------------------------------------------- output: rank r of row qi -> lane r
    for (int qi = wave; qi < QB; qi += WAVES) {
        const int q = q0 + qi;
        if (q >= NQ) break;
        if (SEG) {
            // global row numbers; segments shorter than nsample are padded with (first row of the segment, 1e10)
            if (lane < KK) {
                const bool have = lane < ccount[qi];
                const u64 v = carry[qi * CK + lane];
                const long o = (qbase + q) * k + lane;
                idx_out[o] = have ? (int)(cbase + (long)(unsigned)(v & 0xFFFFFFFFull)) : (int)cbase;
                if (dist_out) dist_out[o] = have ? o2f((unsigned)(v >> 32)) : 1e10f;
            }
        } else if (lane >= drop && lane < KK) {
            const u64 v = carry[qi * CK + lane];
            const long o = ((long)b * N + q) * k + (lane - drop);
            idx_out[o] = (int)(unsigned)(v & 0xFFFFFFFFull);
            if (dist_out) dist_out[o] = o2f((unsigned)(v >> 32));
        }
    }
}

__global__ __launch_bounds__(256) void knn_sqnorm2_kernel(const float *__restrict__ x, int N, long sb, long sc, int c_knn,
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
int fsg_knn_rows_mfma_launch(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn, int k, int flags,
                             int32_t *idx_out, float *dist_out, float *xx_scratch, hipStream_t st) {
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    if (c_knn > 128 || k + drop > 64 || N > 65535 * 16 || xx_scratch == nullptr) return FSG_ERR_UNSUPPORTED;
    if ((long)c_knn * stride_c >= (1L << 29)) return FSG_ERR_UNSUPPORTED;   // buffer resource: 2 GB
    hipLaunchKernelGGL(knn_sqnorm2_kernel, dim3(fsg_cdiv(N, 256), B), dim3(256), 0, st, x, N, (long)stride_b,
                       (long)stride_c, c_knn, xx_scratch);
    FSG_CHECK_LAUNCH("fsg_knn_dense_f32/sqnorm");
    dim3 grid(fsg_cdiv(N, QB), B);
#define FSG_KNN_RM(KS, WV, SV, CHK, CKK) FSG_KNN_RMQS(KS, WV, SV, CHK, CKK, false, false)
#define FSG_KNN_RMQ(KS, WV, SV, CHK, CKK, QL) FSG_KNN_RMQS(KS, WV, SV, CHK, CKK, QL, false)
#define FSG_KNN_RMQS(KS, WV, SV, CHK, CKK, QL, STRM)                                                                   \
    do {                                                                                                               \
        const size_t lds = sizeof(float) * QB * ((CHK) + 4) + sizeof(u64) * (QB * (CKK)) + sizeof(int) * QB +           \
                           ((QL) ? sizeof(float) * 4 * (KS) * QB : 0) + ((STRM) ? 2 * sizeof(int) * QB : 0);           \
        static FsgLdsGrant grant;                                                                                     \
        if (!grant.raise((const void *)knn_rows_mfma_kernel<KS, WV, SV, CHK, CKK, false, QL, STRM>, (int)lds)) {      \
            fsg_set_error("fsg_knn_dense_f32: cannot raise dynamic LDS to %zu", lds);                                 \
            return FSG_ERR_HIP;                                                                                       \
        }                                                                                                             \
        hipLaunchKernelGGL((knn_rows_mfma_kernel<KS, WV, SV, CHK, CKK, false, QL, STRM>), grid, dim3((WV) * 64), lds, st, x, \
                           xx_scratch, N,                                                                              \
                           (long)stride_b, (long)stride_c, c_knn, k, flags, idx_out, dist_out, SegArgs{});             \
    } while (0)
    const bool small_k = k + drop <= 32;                      // carried list fits 32 slots
    // 512-candidate chunks fit two workgroups per CU in LDS but not in registers (241 VGPRs -> 2 waves/SIMD): measured
    // 148 us against 136 us for the 1024-candidate chunks at C=64, so they are opt-in (flag 2048, tests)
    const bool half = small_k && (flags & 2048);
    // 16 waves per workgroup wherever the registers allow (both phases are latency-bound); flag 8192: the 8-wave variants
    const bool w8 = (flags & 8192) != 0;
    // streamed selection (STREAM = true, 512-candidate chunks, 16 waves), formerly flag 131072.  Exact (same parity suite) but
    // MEASURED SLOWER than the two-phase kernel and therefore opt-in: B=8 N=2048 k=20 C=64 101 vs 88 us, C=3 70 vs 52 us;
    // B=4 N=8192 k=40 C=64 567 vs 542 us (tools/knn_ablate_phases.py, eager timing incl. the squared-norm launch).  What the
    // instrumented build (-DFSG_KNN_STATS; the statistics script went with the streamed variant in round 3) showed: the filter works as designed -- 57 survivors
    // per row for a 1536-candidate epoch (expected 60), 3.7 % of the workgroups redo an epoch -- but a streamed epoch costs
    // as much as the distance block + full selection it replaces: with 2-6 tiles per wave and epoch the operand-load
    // latency, the two barriers and the rank-by-counting merge (which dominates the selection either way) are all exposed,
    // and 512-candidate chunks are 14 us slower than 1024-candidate ones for the same reason.  Two co-resident 8-wave
    // workgroups per CU (flag 262144: 256-candidate first chunk, 50 KB of LDS, 101 VGPRs) change nothing: 101 us.
    // (round 3: the streamed instantiations are no longer built into the library -- the STREAM template paths above are kept as
    // the record of the design; re-enable by dispatching FSG_KNN_RMQS(..., true) on a flag here)
    if (c_knn <= 4) { if (!w8) FSG_KNN_RM(1, 16, 128, 1024, 64); else FSG_KNN_RM(1, 8, 128, 1024, 64); }
    else if (c_knn <= 16) { if (!w8) FSG_KNN_RM(4, 16, 128, 1024, 64); else FSG_KNN_RM(4, 8, 128, 1024, 64); }
    else if (c_knn <= 64) {
        if (half) FSG_KNN_RM(16, 8, 64, 512, 32);
        else if (w8) FSG_KNN_RM(16, 8, 128, 1024, 64);
        else if (small_k) FSG_KNN_RMQ(16, 16, 128, 1024, 32, true);
        else FSG_KNN_RMQ(16, 16, 128, 1024, 64, true);   // 131.6 KB rows + 16 KB best lists + 8 KB query operand
    }
    else {
        if (half) FSG_KNN_RM(32, 8, 64, 512, 32);
        else if (!w8 && small_k) FSG_KNN_RMQ(32, 16, 128, 1024, 32, true);   // 131.6 KB rows + 8 KB lists + 16 KB query operand
        else FSG_KNN_RM(32, 8, 128, 1024, 64);
    }
#undef FSG_KNN_RM
#undef FSG_KNN_RMQ
#undef FSG_KNN_RMQS
    FSG_CHECK_LAUNCH("fsg_knn_dense_f32/rows_mfma");
    return FSG_OK;
}

// packed-segment query on the same selection machinery; FSG_ERR_UNSUPPORTED -> caller keeps its scalar kernel
int fsg_knn_segment_rows_launch(const float *xyz, const float *new_xyz, const int32_t *offset, const int32_t *new_offset,
                                int b, int n, int m, int nsample, int32_t *idx, float *dist2, hipStream_t st) {
    if (nsample > 32 || b > 4096) return FSG_ERR_UNSUPPORTED;
    constexpr int WV = 16, SV = 96, CHK = 1024, CKK = 64;
    const size_t lds = sizeof(float) * QB * (CHK + 4) + sizeof(u64) * (QB * CKK) + sizeof(int) * QB;
    static FsgLdsGrant grant;
    if (!grant.raise((const void *)knn_rows_mfma_kernel<1, WV, SV, CHK, CKK, true>, (int)lds)) {
        fsg_set_error("fsg_knn_segment_f32: cannot raise dynamic LDS to %zu", lds);
        return FSG_ERR_HIP;
    }
    const SegArgs sa{new_xyz, offset, new_offset, b};
    hipLaunchKernelGGL((knn_rows_mfma_kernel<1, WV, SV, CHK, CKK, true>), dim3(fsg_cdiv(m, QB) + b), dim3(WV * 64), lds, st,
                       xyz, (const float *)nullptr, n, 0L, 0L, 3, nsample, 0, idx, dist2, sa);
    FSG_CHECK_LAUNCH("fsg_knn_segment_f32/rows");
    return FSG_OK;
}

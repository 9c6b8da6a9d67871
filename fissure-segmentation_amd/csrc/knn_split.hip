// Dense kNN graph build, "coarse sweep + exact refine" behind fsg_knn_dense_ws_f32 / fsg_knn_dense_prepared_f32 /
// fsg_knn_dense_ws_pq_f32 (1024 <= N <= 8192, c_knn <= 128 -- above 64 channels N <= 4096, enforced by plan() --, k + drop <= 64).
// Replaces utils/general_utils.py:43-53,315-327 like the two-phase kernel of knn_rows_mfma.hip and returns the SAME bits (indices
// and distances of oracle/fsg_oracle.c): the matrix cores only NOMINATE candidates, every distance that is ranked or returned
// is the oracle's fp32 fma chain.
//
// Since round 4 the build is TWO launches (knn_nominate_kernel + knn_refine_kernel, second half of this file: read its header
// first); the one-launch kernel of rounds 2-3 (knn_split_kernel, first half) is kept for the one shape class where it is still
// ahead (k + drop > 32 at N > 4096 on 64+ channels: BASELINE config 4) and as an independent cross-check (flag 536870912).
// The algorithm, common to both:
//
//   prep    one pass over the cloud: squared norms (the oracle's chain), a point-major fp32 copy (rows for the refine) and
//           the coarse image of the points in the REGISTER LAYOUT of a 32x32x16 MFMA operand (one 1-KiB block per 32 points
//           and k-step: a wave loads an operand with one fully coalesced 16-byte load per lane).  Two forms:
//           * above 4 channels: ONE fp16 image of the points centred on a sampled mean and scaled by a power of two (both the
//             same for every workgroup of a cloud); points outside the fp16 range are marked and send their cloud down the
//             exact slow path;
//           * up to 4 channels (and flag 1073741824): two bf16 pieces x = hi + lo + r, |r| <= 2^-16 |x| (round-to-nearest-
//             even in integer arithmetic) of the points as they are, three products -- which share ONE k-step at <= 4 channels.
//           A query operand is the same image times -2 (one exponent step, done in the main kernel).
//           (Feature-space builds of DGCNN-seg: emitted by the EdgeConv pass that produces the points, edgeconv.hip; the
//           coordinate build at N = 2048: built by the nominate kernel itself from the (B, C, N) points -- no prep launch.)
//   sweep 1 a workgroup owns 64 queries (two 32-column blocks, resident as B operands); its 8 waves take the candidate
//           tiles (32 rows, A operand) round-robin.  s~(i,j) = |x_j|^2 - 2 x_i.x_j in coarse arithmetic comes out of one fp16
//           (three bf16) MFMAs per k-step with the squared norm as the accumulator's initial value.  Lane (n, h) holds 16
//           candidates of ONE query per tile, so the running minimum of a tile group is a per-lane register: NMIN = 64 group
//           minima per query.
//   tau     K-th smallest of the group minima (disjoint candidate groups, so at least K candidates have s~ <= tau); with
//           |s~ - F| <= eps_i for every candidate (F = the oracle's distance, in the image's units, minus a per-query
//           constant) the K-th smallest oracle distance is <= tau + eps_i and every true neighbour has s~ <= tau + 2 eps_i.
//   sweep 2 the same MFMAs again (bit-identical values); s~ <= tau + 2 eps becomes one bit per candidate in a per-query
//           bitmap in LDS -- branch-free; ~27 bits set per query at k = 20.
//   refine  candidate rows are loaded WHOLE from the point-major copy into an LDS stage, one lane per candidate runs the
//           oracle's distance d = (xx_i - 2 dot) + xx_j, dot = channel-ordered fmaf chain from +0; the (d, j) keys are ranked
//           by counting inside the query's segment and ranks < K are written.
//   slow    a query whose candidate list overflows (massive ties) -- and every query of a cloud with a marked outlier -- is
//           redone by a whole workgroup from the oracle's distances of ALL candidates: exact, slow, rare.
//
// Error bound of the bf16 form (n_i = |x_i|, R = max_j |x_j|, both rounded up; the fp16 form's terms are listed where eps is
// computed, in the centred and scaled units):
//   dropped product terms lo.lo + r.(..)   <= 3.1 * 2^-16 n_i R, times the factor 2          -> 1.0e-4 n_i R
//   fp32 accumulation of 193 terms in the matrix core, any order, truncation allowed          -> 2.6e-5 (R^2 + 2.1 n_i R)
//   the oracle's own fp32 chains against real arithmetic (dot, both norms, two roundings)     -> 9.2e-6 (n_i + R)^2
//
// Debug / measurement flags (bits of `flags`; tools/knn_split_check.py, tools/knn_nominate_stamps.py, tools/knn_split_stamps.py):
// 65536 plain workgroup placement; 4194304 every query through the slow path; 1073741824 the bf16 form above 4 channels;
// 33554432 statistics (fsg_debug_knn_split_stats); 268435456 cycle stamps (fsg_debug_knn_split_stamps / _refine_stamps; forces
// the two-launch form unless 536870912 is set too); 536870912 the one-launch kernel; 134217728 two-launch form without resident
// operand tiles; 67108864 two-launch form with the prep launch where the no-prep path would run (A/B) -- and, one-launch kernel
// only: 67108864 / 8388608 / 16777216 return after the setup / sweep 1 / sweep 2 (timing ablations: the outputs are NOT written).
#include <type_traits>

#include "fsg_common.h"

// debug statistics (flag 33554432): [0] queries refined, [1] their listed candidates, [2] queries on the slow path,
// [3] largest list total
__device__ unsigned long long fsg_knn_split_stats[4];
extern "C" int fsg_debug_knn_split_stats(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fsg_knn_split_stats), sizeof(fsg_knn_split_stats)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[4] = {0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(fsg_knn_split_stats), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}

// cycle stamps (flag 268435456): [workgroup (first 256)][wave][16] shader-clock ticks at the phase boundaries
__device__ unsigned long long fsg_knn_split_stamps[256 * 8 * 16];
extern "C" int fsg_debug_knn_split_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fsg_knn_split_stamps), sizeof(fsg_knn_split_stamps)) != hipSuccess;
}

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int QB = 64;        // queries per workgroup
constexpr int WAVES = 8;
constexpr int MSL = 4;         // group-minimum slots per lane and query block in sweep 1
constexpr int NMIN = 2 * WAVES * MSL;   // group minima per query (64): tau = the K-th smallest of them

__device__ __forceinline__ unsigned f2o(float d) {
    const unsigned u = __float_as_uint(d);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float o2f(unsigned k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
// round-to-nearest-even bf16 of a finite float, in integer arithmetic
__device__ __forceinline__ unsigned bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf16_f(unsigned h) { return __uint_as_float(h << 16); }

struct Split { unsigned hi, lo; };
__device__ __forceinline__ Split split2(float v) {
    Split s;
    s.hi = bf16_rne(v);
    s.lo = bf16_rne(v - bf16_f(s.hi));   // v - hi is exact in fp32
    return s;
}

// wave-wide sum / maximum of a float on the DPP network (row_shr 1,2,4,8, then the row totals broadcast down): lane 63 holds
// the result, returned to every lane through a scalar read
__device__ __forceinline__ float wave_sum_f(float v) {
#define FSG_DPP_F(x, ctrl, rm, bc) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, rm, 0xf, bc))
    v += FSG_DPP_F(v, 0x111, 0xf, true);
    v += FSG_DPP_F(v, 0x112, 0xf, true);
    v += FSG_DPP_F(v, 0x114, 0xf, true);
    v += FSG_DPP_F(v, 0x118, 0xf, true);
    v += FSG_DPP_F(v, 0x142, 0xa, false);
    v += FSG_DPP_F(v, 0x143, 0xc, false);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// maximum of NON-NEGATIVE floats (a masked-out DPP source reads as +0, the identity here)
__device__ __forceinline__ float wave_max_nonneg_f(float v) {
    v = fmaxf(v, FSG_DPP_F(v, 0x111, 0xf, true));
    v = fmaxf(v, FSG_DPP_F(v, 0x112, 0xf, true));
    v = fmaxf(v, FSG_DPP_F(v, 0x114, 0xf, true));
    v = fmaxf(v, FSG_DPP_F(v, 0x118, 0xf, true));
    v = fmaxf(v, FSG_DPP_F(v, 0x142, 0xa, false));
    v = fmaxf(v, FSG_DPP_F(v, 0x143, 0xc, false));
#undef FSG_DPP_F
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---------------------------------------------------------------------------------------------------------------- prep
// grid (Np / 32, B), 256 threads: 32 points = one operand tile.
// KS = k-steps of 16 channels.  HALF: ONE fp16 image of the centred, scaled points (one product per k-step).  Otherwise two
// bf16 pieces of the points as they are (three products); PACK (c_knn <= 4) then packs all three into ONE k-step:
//   k-slots 0-3 hi.qhi, 4-7 hi.qlo, 8-11 lo.qhi, 12-15 zero.
template <int KS, bool PACK, bool HALF>
__global__ __launch_bounds__(256) void knn_split_prep_kernel(const float *__restrict__ x, int N, int Np, long sb, long sc,
                                                             int c_knn, float *__restrict__ xx, float *__restrict__ xt,
                                                             u32x4 *__restrict__ cand, float *__restrict__ xs,
                                                             float *__restrict__ cscale) {
    constexpr int CP = PACK ? 4 : 16 * KS;
    __shared__ float slab[CP][33];
    __shared__ float mu[CP < 64 ? 64 : CP], wred[4];
    const int tid = threadIdx.x, b = blockIdx.y, tile = blockIdx.x, j0 = tile * 32;
    const float *xb = x + (long)b * sb;
    {
        const int pt = tid & 31;
#pragma unroll
        for (int c = tid >> 5; c < CP; c += 8)
            slab[c][pt] = (c < c_knn && j0 + pt < N) ? xb[c * sc + j0 + pt] : 0.f;
    }
    float sigma = 1.f;
    if (HALF) {
        // Centre and scale of the cloud for the fp16 image, from a fixed SAMPLE (four runs of 16 consecutive points at 0, N/4,
        // N/2, 3N/4) that every workgroup of the cloud evaluates identically: any centre / scale is correct (the error bound
        // is written in the centred, scaled norms; points far outside the sample's range are marked, see below), a
        // representative one makes the bound tight.
        // lane = sample point (coalesced: a wave-load reads four 64-byte runs of one channel), wave w takes channels w + 4 r
        const int sp = tid & 63, w = tid >> 6;
        const int pos = min((int)(((long)(sp >> 4) * N / 4) & ~15L), N - 16) + (sp & 15);
        constexpr int NR = CP <= 64 ? 16 : CP / 4;   // channels per wave
        float sv[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) sv[r] = (w + 4 * r) < c_knn ? xb[(long)(w + 4 * r) * sc + pos] : 0.f;
        float dv = 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float m = wave_sum_f(sv[r]) * (1.0f / 64.0f);
            if (sp == 0) mu[w + 4 * r] = m;
            dv = fmaxf(dv, fabsf(sv[r] - m));
        }
        dv = wave_max_nonneg_f(dv);
        if (sp == 0) wred[w] = dv;
        __syncthreads();
        const float maxdev = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
        // power of two that maps the sample's largest deviation into [2^9, 2^10): fp16 keeps 11 bits down to 2^-14
        int ex = (int)((__float_as_uint(maxdev) >> 23) & 255u) - 127;
        int e2 = 9 - ex;
        e2 = max(-100, min(100, e2));
        sigma = (maxdev > 0.f && maxdev < 3.0e38f) ? __uint_as_float((unsigned)(e2 + 127) << 23) : 1.f;
        if (tile == 0 && tid == 0) cscale[b] = sigma;
    } else {
        __syncthreads();
    }
    if (tid < 32) {
        // the oracle's squared norm (channel-ordered fma chain from +0) and, HALF, the centred scaled one (the accumulator's
        // initial value in the sweeps); eight LDS reads in flight per step.  A point with a centred coordinate beyond 2^14
        // (twice that is still a finite fp16) or a non-finite one is an OUTLIER: NaN norm -- the main kernel sends every
        // query of a cloud with an outlier through its exact slow path and never looks at a coarse score of that cloud
        float a = 0.f, c2 = 0.f;
        bool bad = false;
        constexpr int CB = CP < 8 ? CP : 8;
#pragma unroll
        for (int c0 = 0; c0 < CP; c0 += CB) {
            float v[CB], m[CB];
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                v[u] = slab[c0 + u][tid];           // channels >= c_knn hold 0: fma(0, 0, a) = a exactly
                m[u] = HALF ? mu[c0 + u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                a = __builtin_fmaf(v[u], v[u], a);
                if (HALF) {
                    const float vs = (c0 + u < c_knn) ? (v[u] - m[u]) * sigma : 0.f;
                    bad |= !(fabsf(vs) < 16384.0f);
                    c2 = __builtin_fmaf(vs, vs, c2);
                }
            }
        }
        xx[(long)b * Np + j0 + tid] = (j0 + tid < N) ? a : INFINITY;
        if (HALF) {
            bad &= j0 + tid < N;
            xs[(long)b * Np + j0 + tid] = (j0 + tid < N) ? (bad ? __uint_as_float(0x7FC00000u) : c2) : INFINITY;
        }
    }
    for (int e = tid; e < 32 * (CP / 4); e += 256) {
        const int pt = e / (CP / 4), c4 = e % (CP / 4);
        f32x4 v = {slab[4 * c4][pt], slab[4 * c4 + 1][pt], slab[4 * c4 + 2][pt], slab[4 * c4 + 3][pt]};
        *reinterpret_cast<f32x4 *>(xt + ((long)b * Np + j0 + pt) * CP + 4 * c4) = v;
    }
    if (HALF) {
        // (an outlier's row is NOT zeroed: its cloud never uses a coarse score -- every query of it takes the slow path)
        const long T = Np / 32;
        for (int e = tid; e < KS * 64; e += 256) {
            const int lane = e & 63, s = e >> 6;
            const int m = lane & 31, h = lane >> 5;
            unsigned cw[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = 16 * s + 8 * h + i;
                float v = 0.f;
                if (c < c_knn && j0 + m < N) v = (slab[c < CP ? c : 0][m] - mu[c]) * sigma;
                const _Float16 hv = (_Float16)v;    // round to nearest even
                cw[i] = (unsigned)__builtin_bit_cast(unsigned short, hv);
            }
            cand[(((long)b * T + tile) * KS + s) * 64 + lane] =
                u32x4{cw[0] | (cw[1] << 16), cw[2] | (cw[3] << 16), cw[4] | (cw[5] << 16), cw[6] | (cw[7] << 16)};
        }
        return;
    }
    const long T = Np / 32;
    if (PACK) {
        if (tid >= 64 && tid < 128) {     // (the first wave carries the norm chain)
            const int lane = tid & 63, m = lane & 31, h = lane >> 5;
            unsigned hi[4], lo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const Split s = split2(slab[i][m]);
                hi[i] = s.hi; lo[i] = s.lo;
            }
            // candidate image [hi | hi] / [lo | 0]; the query image of the same point is [hi | lo] / [hi | 0] (times -2):
            // the main kernel rebuilds it from this block (lane m holds hi, lane 32 + m holds lo)
            u32x4 c;
            if (h == 0) c = u32x4{hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16)};
            else c = u32x4{lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), 0u, 0u};
            cand[((long)b * T + tile) * 64 + lane] = c;
        }
    } else {
        for (int e = tid; e < KS * 2 * 64; e += 256) {
            const int lane = e & 63, part = (e >> 6) & 1, s = e >> 7;
            const int m = lane & 31, h = lane >> 5;
            unsigned cw[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const Split sp = split2(slab[16 * s + 8 * h + i][m]);
                cw[i] = part ? sp.lo : sp.hi;
            }
            const long o = ((((long)b * T + tile) * KS + s) * 2 + part) * 64 + lane;
            cand[o] = u32x4{cw[0] | (cw[1] << 16), cw[2] | (cw[3] << 16), cw[4] | (cw[5] << 16), cw[6] | (cw[7] << 16)};
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- main
template <int NOPS, bool TWO>
struct Ops {
    u32x4 hi[NOPS], lo[TWO ? NOPS : 1];
};

// m = 2 m + (a < thr): sixteen of these leave bit e = (a[e] < thr) when fed e = 15 .. 0.  Two instructions and no scalar
// register in between: the SIGN of a - thr (exact for finite a != thr, +0 for a == thr, NaN with a clear sign bit for inf - inf)
// is shifted in with v_alignbit.  (A compare writes a scalar mask; the select that follows costs a wait state and the
// compiler's shift / or glue: 3.5 instructions per score.  An inline-asm version of that form also LOST neighbours: an asm
// consumer of MFMA results is invisible to the hazard recognizer -- no wait states after the matrix instruction.)
__device__ __forceinline__ unsigned push_lt(unsigned m, float a, float thr) {
    return __builtin_amdgcn_alignbit(m, __float_as_uint(a - thr), 31);
}
// the smallest float above thr: a <= thr  <=>  a < above(thr) for every finite thr (FLT_MAX -> +inf; -inf stays)
__device__ __forceinline__ float above(float thr) {
    const unsigned u = __float_as_uint(thr);
    if (thr == -INFINITY || thr != thr) return thr;
    if (thr == 0.f) return __uint_as_float(1u);
    return __uint_as_float(thr > 0.f ? u + 1u : u - 1u);
}

// inclusive prefix sum over the 64 lanes on the DPP network (as knn_rows_mfma.hip)
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

template <int KS, bool PACK, bool HALF>
__global__ __launch_bounds__(WAVES * 64, 2) void knn_split_kernel(const float *__restrict__ xx, const float *__restrict__ xt,
                                                                  const u32x4 *__restrict__ cand,
                                                                  const float *__restrict__ xsg,
                                                                  const float *__restrict__ cscale, int N, int Np, int k,
                                                                  int flags, int PC, int32_t *__restrict__ idx_out,
                                                                  float *__restrict__ dist_out) {
    constexpr int CP = PACK ? 4 : 16 * KS;
    constexpr int CPQ = CP + 4;
    constexpr bool PK3 = PACK && !HALF;          // the packed three-product k-step (bf16 pieces, <= 4 channels)
    constexpr int NOP = PK3 ? 1 : KS;            // operand blocks per piece and tile
    constexpr int OPT = HALF ? KS : (PACK ? 1 : 2 * KS);   // 1-KiB operand blocks per tile
    typedef Ops<NOP, !HALF && !PACK> OpsT;
    constexpr int QW = QB / WAVES;         // queries a wave refines
    constexpr int PR = CP > 64 ? 16 : 48;  // candidate rows per refine pass (the LDS stage holds PR rows per wave)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int T = Np / 32;                       // candidate tiles
    const int RSW = T + 1;                       // bitmap row stride in 32-bit words (one word per tile + 1: no bank conflicts)
    // U region: the squared norms and the bitmaps during the sweeps, the row stage of the refine afterwards
    float *xs = reinterpret_cast<float *>(smem);                              // [Np] squared norms (sweeps only)
    unsigned *bm = reinterpret_cast<unsigned *>(xs + Np);                     // [QB][RSW] survivor bitmaps (sweep 2)
    float *mins = reinterpret_cast<float *>(bm);                              // [QB][NMIN] group minima (sweep 1), same storage
    float *stage = reinterpret_cast<float *>(smem);                           // [WAVES][PR][CPQ] candidate rows (refine)
    float *dl = reinterpret_cast<float *>(smem);                              // slow path: [N] distances
    const size_t usz = max((size_t)4 * Np + max((size_t)4 * (((size_t)QB * RSW + 1) & ~(size_t)1), (size_t)4 * QB * NMIN),
                           (size_t)4 * WAVES * PR * CPQ);
    u64 *plist = reinterpret_cast<u64 *>(smem + ((usz + 15) & ~(size_t)15)); // [WAVES][PC] candidates, then keys, of a batch
    float *qrow = reinterpret_cast<float *>(plist + WAVES * PC);              // [QB][CPQ]: query row, then its squared norm
    float *thrL = qrow + QB * CPQ;                                            // [QB]
    int *slowq = reinterpret_cast<int *>(thrL + QB);                          // [QB]
    float *red = reinterpret_cast<float *>(slowq + QB);                       // [24]
    int *segL = reinterpret_cast<int *>(red + 24);                            // [WAVES][QW][2] segment bounds of a batch

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    const bool stamps = (flags & 268435456) != 0;
    auto stamp = [&](int i) {
        if (stamps) {
            const unsigned wg = blockIdx.x + gridDim.x * blockIdx.y;
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (wg < 256 && lane == 0) fsg_knn_split_stamps[(wg * 8 + wave) * 16 + i] = t;
        }
    };
    stamp(0);
    int b = blockIdx.y, q0 = blockIdx.x * QB;
    {   // XCD-aware placement (as knn_rows_mfma.hip): XCD x owns a contiguous eighth of the (cloud, tile) space
        const unsigned L = blockIdx.x + gridDim.x * blockIdx.y, total = gridDim.x * gridDim.y;
        if ((total & 7u) == 0 && !(flags & 65536)) {
            const unsigned V = (L & 7u) * (total >> 3) + (L >> 3);
            b = (int)(V / gridDim.x);
            q0 = (int)(V % gridDim.x) * QB;
        }
    }
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    const int KK = k + drop;
    const bool fix_diag = (flags & FSG_KNN_FIX_DIAG) != 0;
    const int TW = (T - wave + WAVES - 1) / WAVES;   // tiles of this wave (round-robin: tile = wave + 8 i)
    const float *xxb = xx + (long)b * Np;
    const float *xtb = xt + (long)b * Np * CP;
    const u32x4 *candb = cand + (long)b * T * OPT * 64 + lane;

    // ---- setup: the norms the sweeps start their accumulators from into LDS (HALF: of the centred, scaled points; otherwise
    // the oracle's), their maximum, the maximum of the oracle's norms; the workgroup's query rows, the query operands
    const float *xsb = HALF ? xsg + (long)b * Np : xxb;
    float mx = 0.f, mo = 0.f;
    bool outlier = false;
    for (int j = tid * 4; j < Np; j += WAVES * 64 * 4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(xsb + j);
        *reinterpret_cast<f32x4 *>(xs + j) = v;
        f32x4 vo = v;
        if (HALF) vo = *reinterpret_cast<const f32x4 *>(xxb + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (j + e < N) {
                outlier |= v[e] != v[e];            // NaN: the prep kernel's outlier mark
                mx = fmaxf(mx, v[e]);
                mo = fmaxf(mo, vo[e]);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off));
        mo = fmaxf(mo, __shfl_xor(mo, off));
    }
    const bool wave_outlier = __ballot(outlier) != 0;
    if (lane == 0) {
        red[wave] = mx;
        red[8 + wave] = mo;
        red[16 + wave] = wave_outlier ? 1.f : 0.f;
    }
    for (int e = tid; e < QB * (CP / 4); e += WAVES * 64) {
        const int pt = e / (CP / 4), c4 = e % (CP / 4);
        *reinterpret_cast<f32x4 *>(qrow + pt * CPQ + 4 * c4) =
            *reinterpret_cast<const f32x4 *>(xtb + (long)(q0 + pt) * CP + 4 * c4);
    }
    if (tid < QB) {
        slowq[tid] = 0;
        qrow[tid * CPQ + CP] = xxb[q0 + tid];
    }
#pragma unroll
    for (int bk = 0; bk < 2; ++bk)      // this thread's eight minimum slots: groups without a tile stay +inf
#pragma unroll
        for (int g = 0; g < MSL; ++g) mins[(32 * bk + n) * NMIN + wave * 2 * MSL + h * MSL + g] = INFINITY;
    // query operands: the candidate image of the workgroup's own two tiles times -2.  A bf16 / fp16 doubles by one exponent
    // step: (w ^ sign) + (1 << 7 | 1 << 10) per 16-bit half; a zero becomes the smallest normal number (bf16: ~1e-38, nothing;
    // fp16: 2^-14, inside the absolute error term of eps; products with the exact zeros of padding channels stay zero)
    auto neg2 = [](u32x4 w) {
        u32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (w[e] ^ 0x80008000u) + (HALF ? 0x04000400u : 0x00800080u);
        return r;
    };
    OpsT qo[2];
#pragma unroll
    for (int bk = 0; bk < 2; ++bk) {
        const u32x4 *p = candb + (long)(q0 / 32 + bk) * OPT * 64;
        if (HALF) {
#pragma unroll
            for (int s = 0; s < NOP; ++s) qo[bk].hi[s] = neg2(p[s * 64]);
        } else if (PACK) {
            // candidate block: lane (m, 0) = [hi | hi], lane (m, 1) = [lo | 0]; query block: (m, 0) = [hi | lo], (m, 1) = [hi | 0]
            const u32x4 mine = p[0], other = p[(lane ^ 32) - lane];
            u32x4 q;
            if (h == 0) q = u32x4{mine[0], mine[1], other[0], other[1]};
            else q = u32x4{other[0], other[1], 0u, 0u};
            q = neg2(q);
            if (h == 1) { q[2] = 0u; q[3] = 0u; }
            qo[bk].hi[0] = q;
        } else {
#pragma unroll
            for (int s = 0; s < NOP; ++s) {
                qo[bk].hi[s] = neg2(p[(2 * s) * 64]);
                qo[bk].lo[s] = neg2(p[(2 * s + 1) * 64]);
            }
        }
    }

    auto load_tile = [&](OpsT &c, int t) {
        const u32x4 *p = candb + (long)t * OPT * 64;
#pragma unroll
        for (int s = 0; s < NOP; ++s) {
            c.hi[s] = p[(HALF ? s : (PACK ? 0 : 2 * s)) * 64];
            if (!HALF && !PACK) c.lo[s] = p[(2 * s + 1) * 64];
        }
    };
    // s~ of the tile's 32 candidates (rows 8 (e/4) + 4 h + e%4) against the 32 queries of block bk (column n)
    auto scores = [&](const OpsT &c, const f32x16 &init, int bk) {
        f32x16 acc = init;
#pragma unroll
        for (int s = 0; s < NOP; ++s) {
            if (HALF) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, c.hi[s]),
                                                             __builtin_bit_cast(f16x8, qo[bk].hi[s]), acc, 0, 0, 0);
            } else {
                const bf16x8 ch = __builtin_bit_cast(bf16x8, c.hi[s]), qh = __builtin_bit_cast(bf16x8, qo[bk].hi[s]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch, qh, acc, 0, 0, 0);
                if (!PACK) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch, __builtin_bit_cast(bf16x8, qo[bk].lo[s]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c.lo[s]), qh, acc, 0, 0, 0);
                }
            }
        }
        return acc;
    };
    auto load_init = [&](int t) {
        f32x16 r;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xs + 32 * t + 8 * g4 + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[4 * g4 + e] = v[e];
        }
        return r;
    };
    // one sweep over the wave's tiles: operands through a ring of three register sets, requested two tiles ahead (a
    // wave's tile is ~0.3 us of matrix work, an L2 round trip under load several times that); f(scores0, scores1, t, i)
    // Every workgroup of a cloud starts its sweep at a different tile (rotation by the workgroup's position): 32 workgroups
    // walking the same tiles in lockstep would all pull the same cache lines out of the same L2 channels at once.
    const int rot = (int)((blockIdx.x * 5u) % (unsigned)(T / WAVES > 0 ? T / WAVES : 1));
    auto tile_of = [&](int i) {   // i-th tile of this wave: wave + 8 ((i + rot) mod TW)
        int ii = i + rot;
        if (ii >= TW) ii -= TW;
        return wave + WAVES * ii;
    };
    // One sweep over the wave's tiles.  Operands come through a ring of three register sets, requested two tiles ahead.
    // The consumer f of a chain's 16 x 64 scores runs one chain LATE: the vector work on chain c-1 is independent of the
    // twelve MFMAs of chain c, so the scheduler can place it in the matrix instructions' issue gaps (a wave is in-order:
    // consumed right after its own chain, the vector work and the matrix work of the two waves of a SIMD line up in phase
    // and the matrix pipe idles during every consume -- measured: 1500 of 3100 cycles per tile).
    auto sweep = [&](auto &&f) {
        OpsT ring[3];
        if (TW > 0) load_tile(ring[0], tile_of(0));
        if (TW > 1) load_tile(ring[1], tile_of(1));
        // the wave's TW tiles are spread over its MSL minimum groups as evenly as possible (a group ends whenever the running
        // sum of MSL per tile passes TW): every slot is used from TW = MSL on, so every query has its NMIN finite minima even
        // when the tile count is not a multiple of the waves
        int gacc = 0, g = 0;
        f32x16 prev;                // scores of the previous chain (block 1 of the previous tile), not yet consumed
#pragma unroll
        for (int e = 0; e < 16; ++e) prev[e] = INFINITY;
        int pt = TW > 0 ? tile_of(0) : 0, pg = 0;
        bool pgend = false;
        for (int i0 = 0; i0 < TW; i0 += 3) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = i0 + u;
                if (i < TW) {
                    const int t = tile_of(i);
                    if (i + 2 < TW) load_tile(ring[(u + 2) % 3], tile_of(i + 2));
                    gacc += MSL;
                    const bool gend = gacc >= TW;               // wave-uniform: the minimum group is complete
                    const f32x16 init = load_init(t);
                    const f32x16 a0 = scores(ring[u], init, 0);
                    f(prev, 1, pt, pg, pgend);                  // block 1 of the previous tile (first tile: +inf scores)
                    const f32x16 a1 = scores(ring[u], init, 1);
                    f(a0, 0, t, g, gend);
                    prev = a1;
                    pt = t;
                    pg = g;
                    pgend = gend;
                    if (gend) { ++g; gacc -= TW; }
                }
            }
        }
        if (TW > 0) f(prev, 1, pt, pg, pgend);
    };
    stamp(1);
    __syncthreads();   // xs, red, qrow
    stamp(2);
    if (flags & 67108864) return;   // timing ablation: setup only

    // ---------------------------------------------------------------- sweep 1: group minima
    {
        float run[2] = {INFINITY, INFINITY};
        float *mp[2] = {mins + n * NMIN + wave * 2 * MSL + h * MSL, mins + (32 + n) * NMIN + wave * 2 * MSL + h * MSL};
        sweep([&](const f32x16 &a, int bk, int t, int g, bool gend) {
            (void)t;
            float r = run[bk];
#pragma unroll
            for (int e = 0; e < 16; ++e) r = fminf(r, a[e]);
            if (gend) {
                mp[bk][g] = r;
                r = INFINITY;
            }
            run[bk] = r;
        });
    }
    stamp(3);
    __syncthreads();
    stamp(4);
    if (flags & 8388608) return;    // timing ablation: setup + sweep 1

    // ---------------------------------------------------------------- tau and the acceptance bound per query
    {
        float R2 = red[0], Ro2 = red[8], ao = red[16];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            R2 = fmaxf(R2, red[w]);
            Ro2 = fmaxf(Ro2, red[8 + w]);
            ao = fmaxf(ao, red[16 + w]);
        }
        const bool any_outlier = ao != 0.f;
        const float R = sqrtf(R2) * 1.0001f, Ro = sqrtf(Ro2) * 1.0001f;
        const float sg = HALF ? cscale[b] : 1.f;
        // eight lanes per query, eight group minima per lane: the K-th smallest of the 64 by bisection on the key bits with
        // the counts summed over the eight lanes on the DPP network (no scalar round trips: the ballot version of this
        // search spent 2000 cycles per query on VALU -> SALU dependencies)
        const int q = wave * QW + (lane >> 3);
        const float xq = (q0 + q < N) ? xs[q0 + q] : 0.f;
        constexpr int KPL = NMIN / 8;    // minima per lane of the eight that share a query
        // The search runs on distances (minimum + the query's norm): rounding tau up by 2^-11 of a DISTANCE is harmless, 2^-11
        // of s~ = d - |x_i|^2 would not be for clouds far from the origin.  Keys = order-preserving integer images of the
        // floats, HALVED so that the difference of two keys fits 32 bits with its sign: (key <= t) is then the sign bit of
        // key - (t + 1), shifted into a mask with v_alignbit -- two instructions per key and step and no scalar register in
        // between (compare + select + glue: 3.5 and a wait state; the search is VALU-issue-bound).
        unsigned key[KPL];
#pragma unroll
        for (int e4 = 0; e4 < KPL / 4; ++e4) {
            const f32x4 va = *reinterpret_cast<const f32x4 *>(mins + q * NMIN + KPL * (lane & 7) + 4 * e4);
#pragma unroll
            for (int e = 0; e < 4; ++e) key[4 * e4 + e] = f2o(va[e] + xq) >> 1;
        }
        unsigned prefix = 0u;
#pragma unroll 4
        for (int bit = 30; bit >= 11; --bit) {
            const unsigned t1 = (prefix | ((1u << bit) - 1u)) + 1u;
            unsigned m = 0u;
#pragma unroll
            for (int e = 0; e < KPL; ++e) m = __builtin_amdgcn_alignbit(m, key[e] - t1, 31);
            int c = __popc(m);
            c += __builtin_amdgcn_update_dpp(0, c, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
            c += __builtin_amdgcn_update_dpp(0, c, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
            c += __builtin_amdgcn_update_dpp(0, c, 0x141, 0xf, 0xf, true);   // row_half_mirror: the other quad of the eight
            if (c < KK) prefix |= 1u << bit;
        }
        prefix = (prefix << 1) | 1u;     // back to 32-bit keys, rounded up
        stamp(15);
        const unsigned tau = prefix | 0xFFFu;
        float thr;
        if (q0 + q >= N) {
            thr = -INFINITY;                                   // no such query
        } else if ((flags & 4194304) || any_outlier) {   // flag / a marked outlier in the cloud: straight to the slow path
            thr = -INFINITY;
            if ((lane & 7) == 0) slowq[q] = 1;
        } else if (tau >= 0xFF800000u) {                 // fewer than K finite minima: every finite candidate is a nominee
            thr = 3.4028234e38f;
        } else {
            const float td = o2f(tau);
            const float ni = sqrtf(xq) * 1.0001f;
            float eps;
            if (HALF) {
                // centred, scaled units (ni, R from xs; the oracle's norms no, Ro in original units, times the scale); CH = the
                // channel count the constants are derived for (64 up to four k-steps, 128 for the eight-step instantiation):
                //   fp16 rounding of both operands, factor 2           2 (2^-10 + 2^-22) ni R            -> 1.96e-3 ni R
                //   flushed / spurious values below 2^-14 (a = 2^-14)  2 (sqrt(CH) a (ni + R) + CH a^2)  -> 9.8e-4 (ni + R) + 5e-7 at CH = 64
                //   fp32 accumulation of CH + 1 terms in the matrix core   (CH + 2) 2^-23 (R^2 + 2.01 ni R) -> 7.9e-6 (..) at CH = 64
                //   centred norm chain, centring in fp32               CH 2^-24 R^2 + 1.2e-7 (ni + R)^2  -> 3.8e-6 R^2 at CH = 64
                //   the oracle's own fp32 chains against real arithmetic                                  -> 9.2e-6 (sg (no + Ro))^2
                constexpr float CH = KS == 8 ? 128.f : 64.f;
                constexpr float A14 = 6.103515625e-05f;                                   // 2^-14
                constexpr float kFlush1 = (KS == 8 ? 11.3138f : 8.f) * 2.f * A14 * 1.004f;  // 2 sqrt(CH) a, rounded up
                constexpr float kFlush0 = 2.f * CH * A14 * A14 * 1.05f;
                constexpr float kAcc = (CH + 2.f) * 1.1920929e-07f * 1.004f;              // (CH + 2) 2^-23
                constexpr float kNorm = CH * 5.9604645e-08f * 1.004f;                     // CH 2^-24
                const float no = sqrtf(qrow[q * CPQ + CP]) * 1.0001f;
                const float so = sg * (no + Ro);
                eps = (1.96e-3f * ni * R + kFlush1 * (ni + R) + kFlush0 + kAcc * (R * R + 2.01f * ni * R) + kNorm * R * R +
                       1.2e-7f * (ni + R) * (ni + R) + 9.2e-6f * so * so) * 1.001f;
            } else {
                eps = (1.0e-4f * ni * R + 2.6e-5f * (R * R + 2.1f * ni * R) + 9.2e-6f * (ni + R) * (ni + R)) * 1.001f;
            }
            thr = (td - xq) + 2.01f * eps + (fabsf(td) + xq) * 2.4e-7f;
            if (!(thr < 3.4028234e38f)) thr = 3.4028234e38f;
        }
        if ((lane & 7) == 0) thrL[q] = thr;
    }
    stamp(5);
    __syncthreads();   // thrL visible; the minima are consumed: the bitmaps may be written
    stamp(6);

    // ---------------------------------------------------------------- sweep 2: survivor bitmaps, branch-free
    // bit 16 h + e of word t of row q <-> candidate 32 t + 8 (e / 4) + 4 h + e % 4
    {
        const float thr[2] = {above(thrL[n]), above(thrL[32 + n])};   // a <= thr as a strict comparison
        unsigned short *bp[2] = {reinterpret_cast<unsigned short *>(bm + n * RSW) + h,
                                 reinterpret_cast<unsigned short *>(bm + (32 + n) * RSW) + h};
        sweep([&](const f32x16 &a, int bk, int t, int g, bool gend) {
            (void)g;
            (void)gend;
            unsigned m = 0;
#pragma unroll
            for (int e = 15; e >= 0; --e) m = push_lt(m, a[e], thr[bk]);
            bp[bk][2 * t] = (unsigned short)m;
        });
    }
    stamp(7);
    __syncthreads();
    stamp(8);
    if (flags & 16777216) return;   // timing ablation: ... + tau + sweep 2

    // ---------------------------------------------------------------- refine: the oracle's distances, ranked by counting
    // lane-per-candidate form (slow path): rows straight from the point-major copy
    auto exact_d = [&](int q, int j) {
        const float *row = xtb + (long)j * CP;
        const float *qr = qrow + q * CPQ;
        constexpr int CH = CP / 4 >= 8 ? 8 : CP / 4;   // 16-byte pieces in flight
        float dot = 0.f;
#pragma unroll
        for (int c0 = 0; c0 < CP / 4; c0 += CH) {
            f32x4 v[CH];
#pragma unroll
            for (int c4 = 0; c4 < CH; ++c4) v[c4] = *reinterpret_cast<const f32x4 *>(row + 4 * (c0 + c4));
#pragma unroll
            for (int c4 = 0; c4 < CH; ++c4) {
                const f32x4 qv = *reinterpret_cast<const f32x4 *>(qr + 4 * (c0 + c4));
#pragma unroll
                for (int e = 0; e < 4; ++e) dot = __builtin_fmaf(qv[e], v[c4][e], dot);
            }
        }
        const float tt = qr[CP] - 2.0f * dot;
        float d = tt + xxb[j];
        if (fix_diag && j == q0 + q) d = 0.f;
        return d;
    };
    {
        u64 *pl = plist + wave * PC;
        float *stg = stage + wave * PR * CPQ;
        const int qbase = wave * QW;
        // ---- this wave's bitmap rows into registers (the stage overwrites the bitmaps), survivor counts of all its queries
        constexpr int WMAX = 4;                      // bitmap words per lane and query: T <= 256
        unsigned bw[QW][WMAX];
        int cnt[QW], incl[QW], tot[QW];
#pragma unroll
        for (int u = 0; u < QW; ++u) {
            const unsigned *row = bm + (qbase + u) * RSW;
            int c = 0;
#pragma unroll
            for (int v = 0; v < WMAX; ++v) {
                const int wi = lane + 64 * v;
                const unsigned w = (64 * v < T && wi < T) ? row[wi] : 0u;
                bw[u][v] = w;
                c += __popc(w);
            }
            cnt[u] = c;
        }
#pragma unroll
        for (int u = 0; u < QW; ++u) incl[u] = wave_incl_scan(cnt[u]);
#pragma unroll
        for (int u = 0; u < QW; ++u) {
            tot[u] = __builtin_amdgcn_readlane(incl[u], 63);
            const bool exists = q0 + qbase + u < N;
            const bool bad = exists && (tot[u] > PC || tot[u] < KK);   // too many (ties) / too few (cannot happen): slow path
            if ((flags & 33554432) && lane == 0 && exists) {
                atomicAdd(&fsg_knn_split_stats[0], 1ull);
                atomicAdd(&fsg_knn_split_stats[1], (unsigned long long)tot[u]);
                atomicMax(&fsg_knn_split_stats[3], (unsigned long long)tot[u]);
                if (bad) atomicAdd(&fsg_knn_split_stats[2], 1ull);
            }
            if (bad) {
                if (lane == 0) slowq[qbase + u] = 1;
                tot[u] = 0;
                cnt[u] = 0;
#pragma unroll
                for (int v = 0; v < WMAX; ++v) bw[u][v] = 0u;
            }
        }
        stamp(9);
        __syncthreads();   // every wave holds its bitmaps in registers: xs and bm are dead, the row stage may overwrite them
        stamp(10);

        constexpr int LPR = CP / 4;                  // lanes (16-byte pieces) per candidate row
        constexpr int RPI = 64 / LPR;                // rows per load instruction
        constexpr int NI = RPI >= PR ? 1 : (PR + RPI - 1) / RPI; // load instructions per pass of PR rows
        const int lrow = lane / LPR, lpc = lane % LPR;
        int qnext = 0;
        while (qnext < QW) {      // wave-uniform: a batch = as many of the next queries as fit PC candidates
            int st[QW], en[QW];
            bool in[QW];
            int base = 0;
            bool open = true;
            int qend = qnext;
#pragma unroll
            for (int u = 0; u < QW; ++u) {
                st[u] = base;
                in[u] = false;
                if (open && u >= qnext) {
                    if (base + tot[u] <= PC) {
                        in[u] = true;
                        base += tot[u];
                        qend = u + 1;
                    } else {
                        open = false;
                    }
                }
                en[u] = base;
            }
            const int P = base;
            qnext = qend;
            // ---- decode: bit 16 h + e of word t <-> candidate 32 t + 8 (e / 4) + 4 h + e % 4.  An entry's low word is
            // (query's index in the wave << 16) | candidate: inside a query's segment the tag is constant, so it does not
            // change the order of the keys (distance bits << 32 | low word) and spares the search for the segment of an entry
#pragma unroll
            for (int u = 0; u < QW; ++u) {
                if (in[u] && tot[u] > 0) {
                    int pos = st[u] + incl[u] - cnt[u];
#pragma unroll
                    for (int v = 0; v < WMAX; ++v) {
                        if (64 * v >= T) break;   // wave-uniform
                        unsigned w = bw[u][v];
                        while (w) {
                            const int bb = __builtin_ctz(w);
                            w &= w - 1;
                            const int e = bb & 15;
                            pl[pos++] = (u64)((unsigned)(u << 16) | (unsigned)(32 * (lane + 64 * v) + 8 * (e >> 2) + 4 * (bb >> 4) + (e & 3)));
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (qend == QW) stamp(11);
#pragma unroll
            for (int u = 0; u < QW; ++u)
                if (lane == u) { segL[(wave * QW + u) * 2] = st[u]; segL[(wave * QW + u) * 2 + 1] = en[u]; }
            // ---- distances: passes of PR candidates.  Their rows are loaded whole (LPR lanes x 16 bytes per row: every
            // load instruction reads complete rows instead of one 16-byte piece of 64 different rows), staged in LDS, and
            // lanes 0..PR-1 run the channel-ordered fma chain of one candidate each; the next pass's loads are in flight
            // during the chains.
            f32x4 g[NI];
            int jc = 0, jn = 0;
            float xc = 0.f, xn = 0.f;
            auto issue = [&](int p0) {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int r = min(i * RPI + lrow, PR - 1);
                    const int j = (int)((unsigned)pl[min(p0 + r, P - 1)] & 0xFFFFu);
                    g[i] = *reinterpret_cast<const f32x4 *>(xtb + (long)j * CP + 4 * lpc);
                }
                jn = (int)(unsigned)pl[min(p0 + min(lane, PR - 1), P - 1)];   // tagged entry
                xn = xxb[jn & 0xFFFF];
            };
            auto commit = [&]() {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int r = i * RPI + lrow;
                    if (r < PR) *reinterpret_cast<f32x4 *>(stg + r * CPQ + 4 * lpc) = g[i];
                }
                jc = jn;
                xc = xn;
            };
            if (P > 0) {
                issue(0);
                commit();
            }
            if (qend == QW) stamp(12);
            for (int p0 = 0; p0 < P; p0 += PR) {
                __builtin_amdgcn_wave_barrier();
                if (p0 + PR < P) issue(p0 + PR);
                const int p = p0 + lane;
                if (lane < PR && p < P) {
                    const int ql = (jc >> 16) & (QW - 1);
                    const float *qr = qrow + (qbase + ql) * CPQ;
                    const float *row = stg + lane * CPQ;
                    float dot = 0.f;
                    constexpr int HB = CP / 4 >= 8 ? 8 : CP / 4;   // all LDS reads of half a row are issued before its fma chain
#pragma unroll
                    for (int c0 = 0; c0 < CP / 4; c0 += HB) {
                        f32x4 qv[HB], cv[HB];
#pragma unroll
                        for (int c4 = 0; c4 < HB; ++c4) {
                            qv[c4] = *reinterpret_cast<const f32x4 *>(qr + 4 * (c0 + c4));
                            cv[c4] = *reinterpret_cast<const f32x4 *>(row + 4 * (c0 + c4));
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int c4 = 0; c4 < HB; ++c4)
#pragma unroll
                            for (int ee = 0; ee < 4; ++ee) dot = __builtin_fmaf(qv[c4][ee], cv[c4][ee], dot);
                    }
                    const float tt = qr[CP] - 2.0f * dot;
                    float d = tt + xc;
                    if (fix_diag && (jc & 0xFFFF) == q0 + qbase + ql) d = 0.f;
                    pl[p] = ((u64)f2o(d) << 32) | (unsigned)jc;
                }
                __builtin_amdgcn_wave_barrier();
                if (p0 + PR < P) commit();
            }
            __builtin_amdgcn_wave_barrier();
            if (qend == QW) stamp(13);
            // ---- ranks and output
            for (int p0 = 0; p0 < P; p0 += 64) {
                const int p = p0 + lane;
                if (p < P) {
                    const u64 key = pl[p];
                    const int ql = (int)((unsigned)key >> 16) & (QW - 1);
                    const int s = segL[(wave * QW + ql) * 2], e = segL[(wave * QW + ql) * 2 + 1];
                    int r = 0;
                    int t = s;
                    if ((t & 1) && t < e) { r += pl[t] < key ? 1 : 0; ++t; }   // 16-byte reads from an even entry on
#pragma unroll 4
                    for (; t + 2 <= e; t += 2) {
                        const u32x4 kk = *reinterpret_cast<const u32x4 *>(pl + t);
                        const u64 k0 = ((u64)kk[1] << 32) | kk[0], k1 = ((u64)kk[3] << 32) | kk[2];
                        r += (k0 < key ? 1 : 0) + (k1 < key ? 1 : 0);
                    }
                    if (t < e) r += pl[t] < key ? 1 : 0;
                    if (r >= drop && r < KK) {
                        const long o = ((long)b * N + q0 + qbase + ql) * k - drop + r;
                        idx_out[o] = (int)((unsigned)key & 0xFFFFu);
                        if (dist_out) dist_out[o] = o2f((unsigned)(key >> 32));
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    stamp(14);
    __syncthreads();

    // ---------------------------------------------------------------- slow path: whole workgroup, one flagged query at a time
    for (int q = 0; q < QB; ++q) {
        if (!slowq[q]) continue;   // uniform (LDS value, written before the barrier above)
        for (int j = tid; j < N; j += WAVES * 64) dl[j] = exact_d(q, j);
        __syncthreads();
        if (wave == 0) {
            u64 last = 0;
            const long ob = ((long)b * N + q0 + q) * k - drop;
            for (int r = 0; r < KK; ++r) {
                u64 best = ~0ull;
                for (int j = lane; j < N; j += 64) {
                    const u64 kj = ((u64)f2o(dl[j]) << 32) | (unsigned)j;
                    if ((r == 0 || kj > last) && kj < best) best = kj;
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const u64 o = __shfl_xor(best, off);
                    best = o < best ? o : best;
                }
                last = best;
                if (lane == 0 && r >= drop) {
                    idx_out[ob + r] = (int)(unsigned)(best & 0xFFFFFFFFull);
                    if (dist_out) dist_out[ob + r] = o2f((unsigned)(best >> 32));
                }
            }
        }
        __syncthreads();
    }
}


// ================================================================================================ round 4: two launches
// The monolithic kernel above runs ten barrier-separated phases at ONE 8-wave workgroup per CU (130-150 KB of LDS, 160-230
// VGPRs): its refine half is a collection of latency chains at two waves per SIMD.  The same algorithm as two launches:
//   knn_nominate_kernel   setup, sweep 1, tau, sweep 2 (the scores, the acceptance rule and its error bound are the ones above),
//                         then every wave counts the survivors of its eight queries and copies their bitmap rows to the
//                         workspace ((B, Np, Np / 32) words).  A query outside the refine kernel's envelope (more than PCAP
//                         survivors: massive ties; a marked outlier in the cloud; flag 4194304) is served HERE by the exact slow
//                         path and its row is zeroed.  LDS: norms + bitmaps only (25 KB at N = 2048).  Differences to the
//                         monolithic kernel: (1) RES -- up to 64 candidate tiles: a wave's eight operand tiles are loaded ONCE
//                         and stay in registers for both sweeps (no loads, no waits inside the sweeps); (2) tau is the EXACT
//                         K-th smallest of the 64 group minima, found by a bitonic network over eight lanes x eight registers
//                         (264 independent min / max / DPP instructions instead of a 20-step bisection whose steps are serial
//                         chains of ~20 dependent instructions: 4.5 k -> ~1 k cycles).
//   knn_refine_kernel     one wave per query at a time, four independent waves per workgroup, no workgroup barrier, ~12 KB
//                         of LDS per wave and < 100 VGPRs: bitmap row -> candidate list -> whole candidate rows into a
//                         wave-private LDS stage -> one lane per candidate runs the oracle's channel-ordered fma chain with
//                         the query row in SCALAR registers -> (distance bits, index) keys ranked by counting.  Software
//                         pipeline over the wave's queries: the bitmap row of query u + 2 and the candidate rows of query
//                         u + 1 are in flight during the chains and the ranking of query u.  A row with fewer than k + drop
//                         bits belongs to a query the nominate kernel has already written.

__device__ unsigned long long fsg_knn_refine_stamps[512 * 2 * 8];
extern "C" int fsg_debug_knn_refine_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fsg_knn_refine_stamps), sizeof(fsg_knn_refine_stamps)) != hipSuccess;
}

// RAW (<= 4 channels, 64 tiles, the coordinate build of DGCNN-seg): no prep launch at all -- the kernel reads the (B, C, N) points
// themselves: a wave builds its eight operand tiles (two bf16 pieces per coordinate) and their squared norms in registers / LDS
// from three coalesced loads per tile, the workgroup writes the point-major rows and norms of its OWN 64 points for the refine
// launch (the 256 workgroups of a cloud cover it), the slow path evaluates its distances from the points directly.
struct RawPoints {
    const float *x;      // (B, C, N): x[b sb + c sc + j], 16-byte aligned rows
    long sb, sc;
    int c_knn;
    float *xx_out, *xt_out;   // (B, Np) squared norms, (B, Np, 4) point-major rows: what the refine launch reads
    // optional by-product (fsg_knn_dense_ws_pq_f32): pq_out (B, N, rows_pq) = x^T w_pq^T, w_pq (rows_pq, c_knn) row-major -- the
    // per-point rows of the FIRST EdgeConv's decomposed conv (K = c_knn <= 4: a product no matrix unit is needed for)
    const float *w_pq;
    int rows_pq;
    float *pq_out;
};

// pq_out row of point j from its (up to four) coordinates: lane = output column, the weights in registers
__device__ __forceinline__ void raw_pq_rows(const RawPoints &raw, const float *xrb, int N, long row0, int j0, int npts, int tid,
                                            int nthreads) {
    const int cols = raw.rows_pq;
    const int groups = nthreads / cols;                // point groups working side by side (nthreads is a multiple of cols)
    if (groups == 0 || tid >= groups * cols) return;
    const int col = tid % cols, pg = tid / cols;
    float w[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < raw.c_knn) w[c] = raw.w_pq[(long)col * raw.c_knn + c];
    for (int p = pg; p < npts; p += groups) {
        const int j = j0 + p;
        if (j >= N) break;
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < raw.c_knn) a = __builtin_fmaf(xrb[c * raw.sc + j], w[c], a);
        raw.pq_out[(row0 + j) * cols + col] = a;
    }
}

// the same product on its own (shapes whose graph build does not run the RAW path): grid (ceil(N / 64), B), 256 threads
__global__ __launch_bounds__(256) void knn_pq_rows_kernel(const RawPoints raw, int N) {
    const int b = blockIdx.y;
    raw_pq_rows(raw, raw.x + (long)b * raw.sb, N, (long)b * N, blockIdx.x * 64, 64, threadIdx.x, 256);
}
template <int KS, bool PACK, bool HALF, bool RES, bool RAW>
__global__ __launch_bounds__(WAVES * 64, 2) void knn_nominate_kernel(const float *__restrict__ xx, const float *__restrict__ xt,
                                                                     const u32x4 *__restrict__ cand,
                                                                     const float *__restrict__ xsg,
                                                                     const float *__restrict__ cscale, int N, int Np, int k,
                                                                     int flags, int PCAP, unsigned *__restrict__ bmg,
                                                                     int TS, int32_t *__restrict__ idx_out,
                                                                     float *__restrict__ dist_out, const RawPoints raw) {
    static_assert(!RAW || (PACK && !HALF && RES), "RAW: the packed three-product form with resident tiles");
    constexpr int CP = PACK ? 4 : 16 * KS;
    constexpr bool PK3 = PACK && !HALF;
    constexpr int NOP = PK3 ? 1 : KS;
    constexpr int OPT = HALF ? KS : (PACK ? 1 : 2 * KS);
    typedef Ops<NOP, !HALF && !PACK> OpsT;
    constexpr int QW = QB / WAVES;
    constexpr int RT = 8;                        // resident tiles per wave (RES)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int T = Np / 32;
    const int RSW = T + 1;
    const int TWC = (T + WAVES - 1) / WAVES;     // tile slots per wave: tile t = wave + 8 slot
    float *xs = reinterpret_cast<float *>(smem);                              // [WAVES][TWC][32] squared norms, by owning wave
    unsigned *bm = reinterpret_cast<unsigned *>(xs + WAVES * TWC * 32);       // [QB][RSW] survivor bitmaps (sweep 2)
    float *mins = reinterpret_cast<float *>(bm);                              // [QB][NMIN] group minima (sweep 1), same storage
    float *dl = reinterpret_cast<float *>(smem);                              // slow path: [N] distances (over xs)
    const size_t bmw = max(((size_t)QB * RSW + 3) & ~(size_t)3, (size_t)QB * NMIN);
    float *thrL = reinterpret_cast<float *>(bm + bmw);                        // [QB]
    unsigned *slowm = reinterpret_cast<unsigned *>(thrL + QB);                // [2] queries for the slow path, one bit each
    float *red = reinterpret_cast<float *>(slowm + 4);                        // [24]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    const bool stamps = (flags & 268435456) != 0;
    auto stamp = [&](int i) {
        if (stamps) {
            const unsigned wg = blockIdx.x + gridDim.x * blockIdx.y;
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (wg < 256 && lane == 0) fsg_knn_split_stamps[(wg * 8 + wave) * 16 + i] = t;
        }
    };
    stamp(0);
    int b = blockIdx.y, q0 = blockIdx.x * QB;
    {
        const unsigned L = blockIdx.x + gridDim.x * blockIdx.y, total = gridDim.x * gridDim.y;
        if ((total & 7u) == 0 && !(flags & 65536)) {
            const unsigned V = (L & 7u) * (total >> 3) + (L >> 3);
            b = (int)(V / gridDim.x);
            q0 = (int)(V % gridDim.x) * QB;
        }
    }
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    const int KK = k + drop;
    const bool fix_diag = (flags & FSG_KNN_FIX_DIAG) != 0;
    const int TW = RES ? RT : (T - wave + WAVES - 1) / WAVES;    // RES: exactly 64 tiles (the launcher checks), 8 per wave
    const float *xxb = xx + (long)b * Np;
    const float *xtb = xt + (long)b * Np * CP;
    const u32x4 *candb = cand + (long)b * T * OPT * 64 + lane;

    auto load_tile = [&](OpsT &c, int t) {
        const u32x4 *p = candb + (long)t * OPT * 64;
#pragma unroll
        for (int s = 0; s < NOP; ++s) {
            c.hi[s] = p[(HALF ? s : (PACK ? 0 : 2 * s)) * 64];
            if (!HALF && !PACK) c.lo[s] = p[(2 * s + 1) * 64];
        }
    };
    const int rot = (int)((blockIdx.x * 5u) % (unsigned)(T / WAVES > 0 ? T / WAVES : 1));
    auto tile_of = [&](int i) {
        int ii = i + rot;
        if (ii >= TW) ii -= TW;
        return wave + WAVES * ii;
    };
    // ---- setup.  No workgroup barrier in front of sweep 1: a wave stages the norms of ITS OWN tiles (wave-private LDS, 1 KiB
    // per 8 tiles = one 16-byte load per lane) -- the first matrix instruction of a wave waits for that load, its query
    // operands and its first tile only, not for the slowest wave of the workgroup (setup + barrier were 9.7 k cycles of 32 k).
    // Request order = consumption order (a wave's loads return in order): norms, query operands, then (RES) all tiles.
    const float *xsb = HALF ? xsg + (long)b * Np : xxb;
    float *xsw = xs + wave * TWC * 32;
    const int nsl = lane >> 3, npc = 4 * (lane & 7);   // this lane's tile slot (of 8 per round) and piece of its 32 norms
    f32x4 v0 = {INFINITY, INFINITY, INFINITY, INFINITY}, vo0 = v0;
    const float *xrb = RAW ? raw.x + (long)b * raw.sb : nullptr;
    // RAW: the oracle's squared norm (channel-ordered fma chain from +0; channels >= c_knn read as 0: fma(0, 0, a) = a) of 4 points
    auto raw_norm4 = [&](int j) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < raw.c_knn) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(xrb + c * raw.sc + j);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = __builtin_fmaf(v[e], v[e], a[e]);
            }
        }
        return a;
    };
    if (nsl < TW) {
        if (RAW) {
            v0 = raw_norm4(32 * (wave + WAVES * nsl) + npc);
        } else {
            v0 = *reinterpret_cast<const f32x4 *>(xsb + 32 * (wave + WAVES * nsl) + npc);
        }
        vo0 = v0;
        if (HALF) vo0 = *reinterpret_cast<const f32x4 *>(xxb + 32 * (wave + WAVES * nsl) + npc);
    }
    // the oracle's norm of this lane's query of the tau phase (eight lanes per query; the fp16 form's bound needs it)
    const float xo_q = HALF ? xxb[min(q0 + wave * QW + (lane >> 3), Np - 1)] : 0.f;
    // RAW: hi / lo bf16 pieces of point 32 t + (lane % 32), packed two channels per word
    auto raw_pieces = [&](int t, unsigned (&hi)[2], unsigned (&lo)[2]) {
        unsigned hh[4] = {0u, 0u, 0u, 0u}, ll[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < raw.c_knn) {
                const Split sp = split2(xrb[c * raw.sc + 32 * t + n]);
                hh[c] = sp.hi;
                ll[c] = sp.lo;
            }
        }
        hi[0] = hh[0] | (hh[1] << 16); hi[1] = hh[2] | (hh[3] << 16);
        lo[0] = ll[0] | (ll[1] << 16); lo[1] = ll[2] | (ll[3] << 16);
    };
    auto neg2 = [](u32x4 w) {
        u32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (w[e] ^ 0x80008000u) + (HALF ? 0x04000400u : 0x00800080u);
        return r;
    };
    OpsT qo[2];
#pragma unroll
    for (int bk = 0; bk < 2; ++bk) {
        const u32x4 *p = candb + (long)(q0 / 32 + bk) * OPT * 64;
        if (HALF) {
#pragma unroll
            for (int s = 0; s < NOP; ++s) qo[bk].hi[s] = p[s * 64];
        } else if (RAW) {     // query block: lane (m, 0) = [hi | lo], lane (m, 1) = [hi | 0]
            unsigned hi[2], lo[2];
            raw_pieces(q0 / 32 + bk, hi, lo);
            qo[bk].hi[0] = h == 0 ? u32x4{hi[0], hi[1], lo[0], lo[1]} : u32x4{hi[0], hi[1], 0u, 0u};
        } else if (PACK) {
            const u32x4 mine = p[0], other = p[(lane ^ 32) - lane];
            u32x4 q;
            if (h == 0) q = u32x4{mine[0], mine[1], other[0], other[1]};
            else q = u32x4{other[0], other[1], 0u, 0u};
            qo[bk].hi[0] = q;
        } else {
#pragma unroll
            for (int s = 0; s < NOP; ++s) {
                qo[bk].hi[s] = p[(2 * s) * 64];
                qo[bk].lo[s] = p[(2 * s + 1) * 64];
            }
        }
    }
    OpsT tiles[RES ? RT : 1];
    if (RES) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            if (RAW) {        // candidate block: lane (m, 0) = [hi | hi], lane (m, 1) = [lo | 0]
                unsigned hi[2], lo[2];
                raw_pieces(tile_of(i), hi, lo);
                tiles[i].hi[0] = h == 0 ? u32x4{hi[0], hi[1], hi[0], hi[1]} : u32x4{lo[0], lo[1], 0u, 0u};
            } else {
                load_tile(tiles[i], tile_of(i));
            }
        }
    }
    if (RAW && wave == 0) {   // this workgroup's own 64 points for the refine launch: point-major rows + norms
        const int j = q0 + lane;
        f32x4 row = {0.f, 0.f, 0.f, 0.f};
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < raw.c_knn) {
                row[c] = xrb[c * raw.sc + j];
                a = __builtin_fmaf(row[c], row[c], a);
            }
        }
        *reinterpret_cast<f32x4 *>(raw.xt_out + ((long)b * Np + j) * 4) = row;
        raw.xx_out[(long)b * Np + j] = a;
    }
    if (RAW && raw.pq_out) raw_pq_rows(raw, xrb, N, (long)b * N, q0, QB, tid, WAVES * 64);   // this workgroup's 64 points
    float mx = 0.f, mo = 0.f;
    bool outlier = false;
    // (the first round of 8 slots is straight-line code, so that the compiler's load counters stay exact and -- RES -- the
    // tiles are waited for one by one inside sweep 1)
    auto stage_norms = [&](int sl, const f32x4 &v, const f32x4 &vo) {
        *reinterpret_cast<f32x4 *>(xsw + 32 * sl + npc) = v;
        const int j = 32 * (wave + WAVES * sl) + npc;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (j + e < N) {
                outlier |= v[e] != v[e];
                mx = fmaxf(mx, v[e]);
                mo = fmaxf(mo, vo[e]);
            }
        }
    };
    if (nsl < TW) stage_norms(nsl, v0, vo0);
    if (!RES) {
        for (int sl = nsl + 8; sl < TW; sl += 8) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xsb + 32 * (wave + WAVES * sl) + npc);
            f32x4 vo = v;
            if (HALF) vo = *reinterpret_cast<const f32x4 *>(xxb + 32 * (wave + WAVES * sl) + npc);
            stage_norms(sl, v, vo);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off));
        mo = fmaxf(mo, __shfl_xor(mo, off));
    }
    const bool wave_outlier = __ballot(outlier) != 0;
    if (lane == 0) {
        red[wave] = mx;
        red[8 + wave] = mo;
        red[16 + wave] = wave_outlier ? 1.f : 0.f;
    }
    if (tid < 2) slowm[tid] = 0u;
#pragma unroll
    for (int bk = 0; bk < 2; ++bk)
#pragma unroll
        for (int g = 0; g < MSL; ++g) mins[(32 * bk + n) * NMIN + wave * 2 * MSL + h * MSL + g] = INFINITY;
    // query operands = the candidate image of the workgroup's own two tiles times -2 (see the monolithic kernel)
#pragma unroll
    for (int bk = 0; bk < 2; ++bk) {
#pragma unroll
        for (int s = 0; s < NOP; ++s) {
            qo[bk].hi[s] = neg2(qo[bk].hi[s]);
            if (!HALF && !PACK) qo[bk].lo[s] = neg2(qo[bk].lo[s]);
        }
        if (PK3 && h == 1) { qo[bk].hi[0][2] = 0u; qo[bk].hi[0][3] = 0u; }
    }
    auto scores = [&](const OpsT &c, const f32x16 &init, int bk) {
        f32x16 acc = init;
#pragma unroll
        for (int s = 0; s < NOP; ++s) {
            if (HALF) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, c.hi[s]),
                                                             __builtin_bit_cast(f16x8, qo[bk].hi[s]), acc, 0, 0, 0);
            } else {
                const bf16x8 ch = __builtin_bit_cast(bf16x8, c.hi[s]), qh = __builtin_bit_cast(bf16x8, qo[bk].hi[s]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch, qh, acc, 0, 0, 0);
                if (!PACK) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch, __builtin_bit_cast(bf16x8, qo[bk].lo[s]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c.lo[s]), qh, acc, 0, 0, 0);
                }
            }
        }
        return acc;
    };
    auto load_init = [&](int t) {                  // (t = wave + 8 slot: the wave's own staging)
        f32x16 r;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xsw + 4 * (t - wave) + 8 * g4 + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[4 * g4 + e] = v[e];
        }
        return r;
    };
    // One sweep over the wave's tiles, the consumer f one chain late (see the monolithic kernel).  RES: from the resident tiles.
    auto sweep = [&](auto &&f) {
        int gacc = 0, g = 0;
        f32x16 prev;
#pragma unroll
        for (int e = 0; e < 16; ++e) prev[e] = INFINITY;
        int pt = TW > 0 ? tile_of(0) : 0, pg = 0;
        bool pgend = false;
        auto one = [&](const OpsT &ops, int t) {
            gacc += MSL;
            const bool gend = gacc >= TW;
            const f32x16 init = load_init(t);
            const f32x16 a0 = scores(ops, init, 0);
            f(prev, 1, pt, pg, pgend);
            const f32x16 a1 = scores(ops, init, 1);
            f(a0, 0, t, g, gend);
            prev = a1;
            pt = t;
            pg = g;
            pgend = gend;
            if (gend) { ++g; gacc -= TW; }
        };
        if constexpr (RES) {
#pragma unroll
            for (int i = 0; i < RT; ++i) one(tiles[i], tile_of(i));
        } else {
            OpsT ring[3];
            if (TW > 0) load_tile(ring[0], tile_of(0));
            if (TW > 1) load_tile(ring[1], tile_of(1));
            for (int i0 = 0; i0 < TW; i0 += 3) {
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int i = i0 + u;
                    if (i < TW) {
                        if (i + 2 < TW) load_tile(ring[(u + 2) % 3], tile_of(i + 2));
                        one(ring[u], tile_of(i));
                    }
                }
            }
        }
        if (TW > 0) f(prev, 1, pt, pg, pgend);
    };
    stamp(1);
    __builtin_amdgcn_wave_barrier();   // (the wave's own norms and minimum slots: LDS operations of one wave complete in order)
    stamp(2);

    // ---------------------------------------------------------------- sweep 1: group minima
    {
        float run[2] = {INFINITY, INFINITY};
        float *mp[2] = {mins + n * NMIN + wave * 2 * MSL + h * MSL, mins + (32 + n) * NMIN + wave * 2 * MSL + h * MSL};
        sweep([&](const f32x16 &a, int bk, int t, int g, bool gend) {
            (void)t;
            float r = run[bk];
#pragma unroll
            for (int e = 0; e < 16; ++e) r = fminf(r, a[e]);
            if (gend) {
                mp[bk][g] = r;
                r = INFINITY;
            }
            run[bk] = r;
        });
    }
    stamp(3);
    __syncthreads();   // minima, norms (the queries'), red
    stamp(4);

    // ---------------------------------------------------------------- tau and the acceptance bound per query
    {
        float R2 = red[0], Ro2 = red[8], ao = red[16];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            R2 = fmaxf(R2, red[w]);
            Ro2 = fmaxf(Ro2, red[8 + w]);
            ao = fmaxf(ao, red[16 + w]);
        }
        const bool any_outlier = ao != 0.f;
        const float R = sqrtf(R2) * 1.0001f, Ro = sqrtf(Ro2) * 1.0001f;
        const float sg = HALF ? cscale[b] : 1.f;
        // eight lanes per query, eight group minima per lane (element i = 8 l8 + e): tau = the K-th smallest of the 64, taken
        // on DISTANCES (minimum + the query's norm: see the monolithic kernel).  Bitonic network in its one-direction form
        // (block size m: "flip" i <-> i ^ (m - 1), then "disperse" i <-> i ^ j, j = m / 4 ... 1; the lower index keeps the
        // minimum): partners with j < 8 are registers of the same lane, the others sit in lane l8 ^ 1, ^ 2, ^ 3, ^ 7 of the
        // eight -- one DPP move each.
        const int q = wave * QW + (lane >> 3);
        const int l8 = lane & 7;
        const int jq = q0 + q;                      // the query as a candidate: tile jq / 32 = owner wave + 8 slot
        const float xq = (jq < N) ? xs[(((jq >> 5) & (WAVES - 1)) * TWC + (jq >> 8)) * 32 + (jq & 31)] : 0.f;
        static_assert(NMIN == 64, "the tau network sorts 64 group minima");
        float kv[8];
#pragma unroll
        for (int e4 = 0; e4 < 2; ++e4) {
            const f32x4 va = *reinterpret_cast<const f32x4 *>(mins + q * NMIN + 8 * l8 + 4 * e4);
#pragma unroll
            for (int e = 0; e < 4; ++e) kv[4 * e4 + e] = va[e] + xq;
        }
#define FSG_CE(a, b) { const float lo_ = fminf(kv[a], kv[b]), hi_ = fmaxf(kv[a], kv[b]); kv[a] = lo_; kv[b] = hi_; }
#define FSG_DPPF(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true))
        auto in4 = [&]() { FSG_CE(0, 4) FSG_CE(1, 5) FSG_CE(2, 6) FSG_CE(3, 7) };
        auto in2 = [&]() { FSG_CE(0, 2) FSG_CE(1, 3) FSG_CE(4, 6) FSG_CE(5, 7) };
        auto in1 = [&]() { FSG_CE(0, 1) FSG_CE(2, 3) FSG_CE(4, 5) FSG_CE(6, 7) };
        // m = 2, 4, 8: inside the lane
        in1();
        FSG_CE(0, 3) FSG_CE(1, 2) FSG_CE(4, 7) FSG_CE(5, 6)
        in1();
        FSG_CE(0, 7) FSG_CE(1, 6) FSG_CE(2, 5) FSG_CE(3, 4)
        in2();
        in1();
        // cross-lane substages: FLIP pairs register e with the partner's register 7 - e, DISPERSE with its register e
#define FSG_XFLIP(ctrl, lowbit)                                                                        \
        {                                                                                              \
            const bool lower_ = (l8 & (lowbit)) == 0;                                                  \
            float p_[8];                                                                               \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) p_[e] = FSG_DPPF(kv[7 - e], ctrl);           \
            _Pragma("unroll") for (int e = 0; e < 8; ++e)                                              \
                kv[e] = lower_ ? fminf(kv[e], p_[e]) : fmaxf(kv[e], p_[e]);                            \
        }
#define FSG_XDISP(ctrl, lowbit)                                                                        \
        {                                                                                              \
            const bool lower_ = (l8 & (lowbit)) == 0;                                                  \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                            \
                const float p_ = FSG_DPPF(kv[e], ctrl);                                                \
                kv[e] = lower_ ? fminf(kv[e], p_) : fmaxf(kv[e], p_);                                  \
            }                                                                                          \
        }
        FSG_XFLIP(0xB1, 1)            // m = 16: i ^ 15 -> lane l8 ^ 1 (quad_perm [1,0,3,2])
        in4(); in2(); in1();
        FSG_XFLIP(0x1B, 2)            // m = 32: i ^ 31 -> lane l8 ^ 3 (quad_perm [3,2,1,0])
        FSG_XDISP(0xB1, 1)            //         j = 8
        in4(); in2(); in1();
        FSG_XFLIP(0x141, 4)           // m = 64: i ^ 63 -> lane l8 ^ 7 (row_half_mirror)
        FSG_XDISP(0x4E, 2)            //         j = 16 -> lane l8 ^ 2 (quad_perm [2,3,0,1])
        FSG_XDISP(0xB1, 1)            //         j = 8
        in4(); in2(); in1();
#undef FSG_XFLIP
#undef FSG_XDISP
#undef FSG_DPPF
#undef FSG_CE
        // element KK - 1 of the sorted 64: register (KK - 1) % 8 of lane (KK - 1) / 8 of the query's eight
        const int se = (KK - 1) & 7, sl = (KK - 1) >> 3;
        float sel = kv[0];
#pragma unroll
        for (int e = 1; e < 8; ++e) sel = (se == e) ? kv[e] : sel;
        const float td = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * ((lane & ~7) | sl), __builtin_bit_cast(int, sel)));
        stamp(15);
        float thr;
        if (q0 + q >= N) {
            thr = -INFINITY;
        } else if ((flags & 4194304) || any_outlier) {
            thr = -INFINITY;
            if (l8 == 0) atomicOr(&slowm[q >> 5], 1u << (q & 31));
        } else if (!(td < INFINITY)) {     // fewer than K finite minima: every finite candidate is a nominee
            thr = 3.4028234e38f;
        } else {
            const float ni = sqrtf(xq) * 1.0001f;
            float eps;
            if (HALF) {
                constexpr float CH = KS == 8 ? 128.f : 64.f;
                constexpr float A14 = 6.103515625e-05f;
                constexpr float kFlush1 = (KS == 8 ? 11.3138f : 8.f) * 2.f * A14 * 1.004f;
                constexpr float kFlush0 = 2.f * CH * A14 * A14 * 1.05f;
                constexpr float kAcc = (CH + 2.f) * 1.1920929e-07f * 1.004f;
                constexpr float kNorm = CH * 5.9604645e-08f * 1.004f;
                const float no = sqrtf(xo_q) * 1.0001f;
                const float so = sg * (no + Ro);
                eps = (1.96e-3f * ni * R + kFlush1 * (ni + R) + kFlush0 + kAcc * (R * R + 2.01f * ni * R) + kNorm * R * R +
                       1.2e-7f * (ni + R) * (ni + R) + 9.2e-6f * so * so) * 1.001f;
            } else {
                eps = (1.0e-4f * ni * R + 2.6e-5f * (R * R + 2.1f * ni * R) + 9.2e-6f * (ni + R) * (ni + R)) * 1.001f;
            }
            thr = (td - xq) + 2.01f * eps + (fabsf(td) + xq) * 2.4e-7f;
            if (!(thr < 3.4028234e38f)) thr = 3.4028234e38f;
        }
        if (l8 == 0) thrL[q] = thr;
    }
    stamp(5);
    __syncthreads();
    stamp(6);

    // ---------------------------------------------------------------- sweep 2: survivor bitmaps
    {
        const float thr[2] = {above(thrL[n]), above(thrL[32 + n])};
        unsigned short *bp[2] = {reinterpret_cast<unsigned short *>(bm + n * RSW) + h,
                                 reinterpret_cast<unsigned short *>(bm + (32 + n) * RSW) + h};
        sweep([&](const f32x16 &a, int bk, int t, int g, bool gend) {
            (void)g;
            (void)gend;
            unsigned m = 0;
#pragma unroll
            for (int e = 15; e >= 0; --e) m = push_lt(m, a[e], thr[bk]);
            bp[bk][2 * t] = (unsigned short)m;
        });
    }
    stamp(7);
    __syncthreads();
    stamp(8);

    // ---------------------------------------------------------------- survivor counts; bitmap rows out
    {
        constexpr int WMAX = RES ? 1 : 4;            // bitmap words per lane and row (RES: 64 tiles)
        const int qbase = wave * QW;
        unsigned w[QW][WMAX];
        int c[QW];
#pragma unroll
        for (int u = 0; u < QW; ++u) {
            const unsigned *row = bm + (qbase + u) * RSW;
            c[u] = 0;
#pragma unroll
            for (int v = 0; v < WMAX; ++v) {
                const int wi = lane + 64 * v;
                w[u][v] = (64 * v < T && wi < T) ? row[wi] : 0u;
                c[u] += __popc(w[u][v]);
            }
        }
#pragma unroll
        for (int u = 0; u < QW; ++u) c[u] = wave_incl_scan(c[u]);
#pragma unroll
        for (int u = 0; u < QW; ++u) {
            const int tot = __builtin_amdgcn_readlane(c[u], 63);
            const bool exists = q0 + qbase + u < N;
            const bool bad = exists && (tot > PCAP || tot < KK);
            if ((flags & 33554432) && lane == 0 && exists) {
                atomicAdd(&fsg_knn_split_stats[0], 1ull);
                atomicAdd(&fsg_knn_split_stats[1], (unsigned long long)tot);
                atomicMax(&fsg_knn_split_stats[3], (unsigned long long)tot);
                if (bad) atomicAdd(&fsg_knn_split_stats[2], 1ull);
            }
            if (bad && lane == 0) atomicOr(&slowm[(qbase + u) >> 5], 1u << ((qbase + u) & 31));
            unsigned *grow = bmg + ((long)b * Np + q0 + qbase + u) * TS;   // rows padded to TS words (a multiple of 8), padding zeroed
#pragma unroll
            for (int v = 0; v < WMAX; ++v) {
                const int wi = lane + 64 * v;
                if (64 * v < TS && wi < TS) grow[wi] = (bad || wi >= T) ? 0u : w[u][v];
            }
        }
    }
    stamp(9);
    __syncthreads();
    stamp(10);

    // ---------------------------------------------------------------- slow path (rare): whole workgroup, one query at a time
    u64 sm = ((u64)slowm[1] << 32) | slowm[0];     // uniform
    while (sm) {
        const int q = __builtin_ctzll(sm);
        sm &= sm - 1;
        // (RAW: rows and norms straight from the (B, C, N) points; other workgroups' point-major rows are not visible here)
        auto raw_row = [&](int j, float &nrm) {
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            nrm = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < raw.c_knn) {
                    r[c] = xrb[c * raw.sc + j];
                    nrm = __builtin_fmaf(r[c], r[c], nrm);
                }
            }
            return r;
        };
        const float *qr = xtb + (long)(q0 + q) * CP;
        float xq = 0.f;
        f32x4 qraw = {0.f, 0.f, 0.f, 0.f};
        if (RAW) qraw = raw_row(q0 + q, xq);
        else xq = xxb[q0 + q];
        __syncthreads();                            // the previous round's (and the sweeps') readers of the overlaid storage
        for (int j = tid; j < N; j += WAVES * 64) {
            const float *row = xtb + (long)j * CP;
            float dot = 0.f, xj = 0.f;
            if (RAW) {
                const f32x4 v = raw_row(j, xj);
#pragma unroll
                for (int e = 0; e < 4; ++e) dot = __builtin_fmaf(qraw[e], v[e], dot);
            } else {
#pragma unroll 4
                for (int c4 = 0; c4 < CP / 4; ++c4) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(row + 4 * c4);
                    const f32x4 qv = *reinterpret_cast<const f32x4 *>(qr + 4 * c4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dot = __builtin_fmaf(qv[e], v[e], dot);
                }
                xj = xxb[j];
            }
            const float tt = xq - 2.0f * dot;
            float d = tt + xj;
            if (fix_diag && j == q0 + q) d = 0.f;
            dl[j] = d;
        }
        __syncthreads();
        if (wave == 0) {
            u64 last = 0;
            const long ob = ((long)b * N + q0 + q) * k - drop;
            for (int r = 0; r < KK; ++r) {
                u64 best = ~0ull;
                for (int j = lane; j < N; j += 64) {
                    const u64 kj = ((u64)f2o(dl[j]) << 32) | (unsigned)j;
                    if ((r == 0 || kj > last) && kj < best) best = kj;
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const u64 o = __shfl_xor(best, off);
                    best = o < best ? o : best;
                }
                last = best;
                if (lane == 0 && r >= drop) {
                    idx_out[ob + r] = (int)(unsigned)(best & 0xFFFFFFFFull);
                    if (dist_out) dist_out[ob + r] = o2f((unsigned)(best >> 32));
                }
            }
        }
    }
}

// Refine: exact distances of the nominated candidates, ranked.  grid (Np / (RW QW), B), RW independent waves per workgroup (no
// workgroup barrier); a wave owns QW = 8 consecutive queries and works on them TOGETHER so that all 64 lanes are busy:
//   bitmap rows   eight lanes per query, a lane owns 8 consecutive words of a 64-word chunk (two 16-byte loads; rows are
//                 padded to a multiple of 8 words by the nominate kernel);
//   sub-batch     as many of the next queries as fit ECAP entries (normally all eight); a query with fewer than k + drop or
//                 more than PCAP bits has been served by the nominate kernel and contributes nothing;
//   decode        entry = (query's index in the wave << 16) | candidate into the low word of its key slot; positions from a
//                 wave scan (query-major), eight-lane prefix sums per chunk;
//   distances     passes of 64 entries x slabs of CS <= 32 channels: 128-byte half rows are loaded whole (8 lanes x 16 bytes),
//                 staged in LDS ([64][CS + 4] floats: 9 KB instead of 17 for whole rows -- the fma chain of an entry simply
//                 continues over the slabs, in channel order), one lane per entry, its query's row read from LDS; the loads of
//                 the next step are in flight during the chains of this one;
//   ranks         one lane per entry counts the smaller keys of its query's segment; ranks < k + drop are written.
// LDS per wave: 8 ECAP + 256 (CS + 4) + 32 (CP + 4) + 64 bytes (13.5 KB at 64 channels).
constexpr int RQW = 8;      // queries per wave
constexpr int RECAP = 256;  // entries per sub-batch
template <int CP>
__global__ __launch_bounds__(128) void knn_refine_kernel(const float *__restrict__ xx, const float *__restrict__ xt,
                                                         const unsigned *__restrict__ bmg, int N, int Np, int TS, int k,
                                                         int flags, int PCAP, int32_t *__restrict__ idx_out,
                                                         float *__restrict__ dist_out) {
    constexpr int QW = RQW, ECAP = RECAP, RW = 2;
    constexpr int CPQ = CP + 4;
    constexpr bool STAGE = CP > 4;                 // 16-byte rows (<= 4 channels) come straight into the lane's registers
    constexpr int CS = CP > 32 ? 32 : CP;          // channels per slab
    constexpr int NSLAB = CP / CS;
    constexpr int CSQ = CS + 4;
    constexpr int LS = STAGE ? CS / 4 : 1;         // lanes (16-byte pieces) per slab row
    constexpr int RPI = 64 / LS;                   // rows per load instruction
    constexpr int NI = STAGE ? 64 / RPI : 1;       // load instructions per step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr size_t wbytes = (size_t)8 * ECAP + (STAGE ? (size_t)4 * 64 * CSQ : 0) + (size_t)4 * QW * CPQ + 128;
    u64 *keys = reinterpret_cast<u64 *>(smem + wave * wbytes);                  // [ECAP] entries, then (distance bits, entry)
    unsigned *klo = reinterpret_cast<unsigned *>(keys);
    float *stg = reinterpret_cast<float *>(keys + ECAP);                         // [64][CSQ]
    float *qrows = stg + (STAGE ? 64 * CSQ : 0);                                 // [QW][CPQ]: query row, then its squared norm
    int *seg = reinterpret_cast<int *>(qrows + QW * CPQ);                        // [QW][2] segment bounds, then a dump slot
    unsigned *dump = reinterpret_cast<unsigned *>(seg + 2 * QW);                 // where the decode's predicated-off stores go
    const bool stamps = (flags & 268435456) != 0;
    auto stamp = [&](int i) {
        if (stamps) {
            const unsigned wg = blockIdx.x + gridDim.x * blockIdx.y;
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (wg < 512 && lane == 0) fsg_knn_refine_stamps[(wg * 2 + wave) * 8 + i] = t;
        }
    };
    stamp(0);
    int b = blockIdx.y, qw0 = blockIdx.x * RW * QW;
    {
        const unsigned L = blockIdx.x + gridDim.x * blockIdx.y, total = gridDim.x * gridDim.y;
        if ((total & 7u) == 0 && !(flags & 65536)) {
            const unsigned V = (L & 7u) * (total >> 3) + (L >> 3);
            b = (int)(V / gridDim.x);
            qw0 = (int)(V % gridDim.x) * RW * QW;
        }
    }
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    const int KK = k + drop;
    const bool fix_diag = (flags & FSG_KNN_FIX_DIAG) != 0;
    const float *xxb = xx + (long)b * Np;
    const float *xtb = xt + (long)b * Np * CP;
    const int qbeg = qw0 + wave * QW;              // wave-uniform; Np is a multiple of RW QW, so the rows exist
    if (qbeg >= N) return;
    const int qi = lane >> 3, l8 = lane & 7;
    const int NCH = (TS + 63) >> 6;                // 64-word chunks of a bitmap row
    const unsigned *brow = bmg + ((long)b * Np + qbeg + qi) * TS + 8 * l8;

    // ---- bitmap rows: per-query totals (chunk 0 stays in registers)
    u32x4 wa = {0u, 0u, 0u, 0u}, wb = wa;
    auto load_chunk = [&](int ch, u32x4 &a, u32x4 &c) {
        a = u32x4{0u, 0u, 0u, 0u};
        c = a;
        if (64 * ch + 8 * l8 < TS) {               // TS is a multiple of 8: all eight words or none
            a = *reinterpret_cast<const u32x4 *>(brow + 64 * ch);
            c = *reinterpret_cast<const u32x4 *>(brow + 64 * ch + 4);
        }
    };
    auto popc8 = [](const u32x4 &a, const u32x4 &c) {
        return __popc(a[0]) + __popc(a[1]) + __popc(a[2]) + __popc(a[3]) + __popc(c[0]) + __popc(c[1]) + __popc(c[2]) + __popc(c[3]);
    };
    auto sum8 = [](int c) {                         // sum over the eight lanes of a query, in every lane
        c += __builtin_amdgcn_update_dpp(0, c, 0xB1, 0xf, 0xf, true);
        c += __builtin_amdgcn_update_dpp(0, c, 0x4E, 0xf, 0xf, true);
        c += __builtin_amdgcn_update_dpp(0, c, 0x141, 0xf, 0xf, true);
        return c;
    };
    load_chunk(0, wa, wb);
    // the wave's query rows (+ squared norms) into LDS while the bitmaps arrive
    {
        constexpr int LQ = CP / 4;                  // lanes per query row
        constexpr int RQ = 64 / LQ > QW ? QW : 64 / LQ;   // rows per instruction
#pragma unroll
        for (int r0 = 0; r0 < QW; r0 += RQ) {
            const int r = r0 + lane / LQ, c = lane % LQ;
            if (lane < RQ * LQ && r < QW)
                *reinterpret_cast<f32x4 *>(qrows + r * CPQ + 4 * c) =
                    *reinterpret_cast<const f32x4 *>(xtb + (long)min(qbeg + r, Np - 1) * CP + 4 * c);
        }
        if (lane < QW) qrows[lane * CPQ + CP] = xxb[min(qbeg + lane, Np - 1)];
    }
    int tq = popc8(wa, wb);
    for (int ch = 1; ch < NCH; ++ch) {
        u32x4 a, c;
        load_chunk(ch, a, c);
        tq += popc8(a, c);
    }
    tq = sum8(tq);
    const bool qvalid = qbeg + qi < N && tq >= KK && tq <= PCAP;
    const int tqv = qvalid ? tq : 0;
    const int tq2 = (tqv + 1) & ~1;                 // a segment starts at an even entry and holds an even number of them: the
    int totu[QW];                                   // ranking reads keys in 16-byte pairs; an odd one is closed by a PAD entry
#pragma unroll
    for (int u = 0; u < QW; ++u) totu[u] = __builtin_amdgcn_readlane(tq2, 8 * u);
    constexpr unsigned PADF = 0x80000000u;          // pad entry: a copy of a real one (its row loads are harmless), key = all ones
    stamp(1);

    int qa = 0;
    while (qa < QW) {                               // wave-uniform: a sub-batch = as many of the next queries as fit ECAP
        int E = 0, qb = qa;
        {
            bool open = true;
#pragma unroll
            for (int u = 0; u < QW; ++u) {
                if (open && u >= qa) {
                    if (E + totu[u] <= ECAP) { E += totu[u]; qb = u + 1; }
                    else open = false;
                }
            }
        }
        const bool mine = qi >= qa && qi < qb;      // this lane's query belongs to the sub-batch
        // ---- segment of every query: a wave scan over the totals parked in the first lane of each query
        const int x0 = (mine && l8 == 0) ? tq2 : 0;
        const int sc = wave_incl_scan(x0);
        const int sstart = sc - (mine ? tq2 : 0);   // valid in all eight lanes of a query of the sub-batch
        if (l8 == 0) { seg[2 * qi] = mine ? sstart : 0; seg[2 * qi + 1] = mine ? sstart + tq2 : 0; }
        int maxlen = 0;                             // longest segment of the sub-batch (uniform)
#pragma unroll
        for (int u = 0; u < QW; ++u) maxlen = (u >= qa && u < qb && totu[u] > maxlen) ? totu[u] : maxlen;
        // ---- decode, chunk by chunk
        int runoff = 0;
        for (int ch = 0; ch < NCH; ++ch) {
            u32x4 a = wa, c = wb;
            if (ch > 0) load_chunk(ch, a, c);
            if (!mine || tqv == 0) { a = u32x4{0u, 0u, 0u, 0u}; c = a; }
            const int cn = popc8(a, c);
            // exclusive prefix over the eight lanes of the query (row_shr inside the 16-lane row, masked at the group start)
            int ps = cn;
            { const int t = __builtin_amdgcn_update_dpp(0, ps, 0x111, 0xf, 0xf, true); ps += l8 >= 1 ? t : 0; }
            { const int t = __builtin_amdgcn_update_dpp(0, ps, 0x112, 0xf, 0xf, true); ps += l8 >= 2 ? t : 0; }
            { const int t = __builtin_amdgcn_update_dpp(0, ps, 0x114, 0xf, 0xf, true); ps += l8 >= 4 ? t : 0; }
            int pos = sstart + runoff + ps - cn;
            runoff += sum8(cn);
            const unsigned tagw = (unsigned)qi << 16;
            const int wbase = 64 * ch + 8 * l8;
            auto put = [&](unsigned &w, int v) {     // the lowest set bit of w -> entry at pos; no bit: the store goes to the dump slot
                const bool has = w != 0u;
                const int bb = __builtin_ctz(w | 0x80000000u);
                const int e = bb & 15;
                const unsigned en = tagw | (unsigned)(32 * (wbase + v) + 8 * (e >> 2) + 4 * (bb >> 4) + (e & 3));
                unsigned *dst = has ? klo + 2 * pos : dump;
                *dst = en;
                pos += has ? 1 : 0;
                w &= w - 1u;                          // (0 stays 0)
            };
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                unsigned w = v < 4 ? a[v & 3] : c[v & 3];
                put(w, v);
                put(w, v);
                while (w) put(w, v);                 // more than two survivors in one 32-candidate word: rare
            }
        }
        if (mine && l8 == 7 && (tqv & 1)) klo[2 * (sstart + tqv)] = PADF | (klo[2 * sstart] & 0x7FFFFFFFu);   // (after this lane's own stores: same wave, in order)
        __builtin_amdgcn_wave_barrier();
        if (qa == 0) stamp(2);
        // pad the list to a multiple of 64 with copies of its last entry: the loaders of the last pass need no index clamp
        {
            const int E64 = (E + 63) & ~63;
            if (E > 0 && E + lane < E64) klo[2 * (E + lane)] = klo[2 * (E - 1)];
        }
        __builtin_amdgcn_wave_barrier();
        // ---- distances: steps t = (pass of 64 entries, slab of CS channels).  Rows come through a ring of TWO register sets:
        // step t commits its set to the LDS stage, requests step t + 2 into the same registers and then runs its chains, so
        // every request has two steps of chains in front of it (one step of lookahead left the wave waiting ~1.5 k cycles per
        // step: the chain of a step is ~500 cycles, an L2 round trip under load several times that).  Buffer loads: the row
        // offset is one v_lshl_add from the entry, the slab's offset is scalar.
        const int rs = lane / LS, lc = lane % LS;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(xtb), 0, __builtin_amdgcn_readfirstlane((int)min((long)Np * CP * 4, 0x7FFFFFFFL)), 0x00020000);
        const int NT = ((E + 63) >> 6) * NSLAB;
        u32x4 g[2][NI];
        float xnr[2] = {0.f, 0.f};
        const unsigned short *k16 = reinterpret_cast<const unsigned short *>(keys);
        auto issue = [&](auto slot, int t) {
            constexpr int S = decltype(slot)::value;
            const int p0 = (t / NSLAB) << 6, sl = t % NSLAB;
            if (STAGE) {
                const unsigned short *ep = k16 + 4 * (p0 + rs);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const unsigned j = ep[4 * i * RPI];
                    g[S][i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (j * (unsigned)(CP * 4)) + (unsigned)(16 * lc),
                                                                    sl * CS * 4, 0);
                }
            }
            if (sl == 0) {
                const unsigned jo = k16[4 * (p0 + lane)];
                if (!STAGE) g[S][0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, jo * 16u, 0, 0);
                xnr[S] = xxb[jo];
            }
        };
        unsigned ent = 0u;
        float dot = 0.f, xc = 0.f;
        auto step = [&](auto slot, int t) {
            constexpr int S = decltype(slot)::value;
            const int p0 = (t / NSLAB) << 6, sl = t % NSLAB;
            const int p = p0 + lane;
            if (sl == 0) {
                ent = klo[2 * p];                    // this lane's entry (before its slot takes the key); a pad copy behind E
                dot = 0.f;
                xc = xnr[S];
            }
            const u32x4 own = g[S][0];
            if (STAGE) {
#pragma unroll
                for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4 *>(stg + (i * RPI + rs) * CSQ + 4 * lc) = g[S][i];
            }
            __builtin_amdgcn_wave_barrier();
            if (t + 2 < NT) issue(slot, t + 2);
            __builtin_amdgcn_sched_barrier(0);       // the chains stay BEHIND the requests and in FRONT of the next commit
            const float *qr = qrows + ((ent >> 16) & 7u) * CPQ;
            if (STAGE) {
                const float *row = stg + lane * CSQ;
                constexpr int HB = CS / 4 >= 4 ? 4 : CS / 4;
#pragma unroll
                for (int c0 = 0; c0 < CS / 4; c0 += HB) {
                    f32x4 cv[HB], qv[HB];
#pragma unroll
                    for (int c4 = 0; c4 < HB; ++c4) {
                        cv[c4] = *reinterpret_cast<const f32x4 *>(row + 4 * (c0 + c4));
                        qv[c4] = *reinterpret_cast<const f32x4 *>(qr + sl * CS + 4 * (c0 + c4));
                    }
#pragma unroll
                    for (int c4 = 0; c4 < HB; ++c4)
#pragma unroll
                        for (int ee = 0; ee < 4; ++ee) dot = __builtin_fmaf(qv[c4][ee], cv[c4][ee], dot);
                }
            } else {
                const f32x4 qv = *reinterpret_cast<const f32x4 *>(qr);
                const f32x4 ov = __builtin_bit_cast(f32x4, own);
#pragma unroll
                for (int ee = 0; ee < 4; ++ee) dot = __builtin_fmaf(qv[ee], ov[ee], dot);
            }
            if (sl == NSLAB - 1 && p < E) {
                const int jc = (int)(ent & 0xFFFFu);
                const float tt = qr[CP] - 2.0f * dot;
                float d = tt + xc;
                if (fix_diag && jc == qbeg + (int)((ent >> 16) & 7u)) d = 0.f;
                keys[p] = (ent & PADF) ? ~0ull : (((u64)f2o(d) << 32) | ent);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_wave_barrier();
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        if (NT > 0) issue(S0{}, 0);
        if (NT > 1) issue(S1{}, 1);
        for (int t = 0; t < NT; t += 2) {
            step(S0{}, t);
            if (t + 1 < NT) step(S1{}, t + 1);
        }
        __builtin_amdgcn_wave_barrier();
        if (qa == 0) stamp(3);
        // ---- ranks and output (the tag in the low word is constant inside a segment: the key order is (distance, candidate)).
        // A lane ranks its (up to) four entries p = lane + 64 m TOGETHER: four independent streams of 16-byte key reads, so the
        // LDS round trips overlap (one entry at a time, the loop was a chain of dependent reads: 8.7 k cycles per sub-batch).
        {
            u64 key[4];
            int sb[4], se[4], r[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int p = 64 * m + lane;
                key[m] = keys[min(p, ECAP - 1)];
                const int tg = (int)(((unsigned)key[m] >> 16) & 7u);
                const bool ok = p < E;
                sb[m] = ok ? seg[2 * tg] : 0;
                se[m] = ok ? seg[2 * tg + 1] : 0;
                r[m] = 0;
            }
            for (int it = 0; it < maxlen; it += 2) {     // (no early exit per stream: branches would serialise the four reads)
                u32x4 kk[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) kk[m] = *reinterpret_cast<const u32x4 *>(keys + sb[m] + it);   // (behind a short segment: other keys / the stage, masked)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const u64 k0 = ((u64)kk[m][1] << 32) | kk[m][0], k1 = ((u64)kk[m][3] << 32) | kk[m][2];
                    const int c2 = (k0 < key[m] ? 1 : 0) + (k1 < key[m] ? 1 : 0);
                    r[m] += sb[m] + it < se[m] ? c2 : 0;
                }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int p = 64 * m + lane;
                if (p < E && r[m] >= drop && r[m] < KK && !((unsigned)key[m] & PADF)) {
                    const int tg = (int)(((unsigned)key[m] >> 16) & 7u);
                    const long o = ((long)b * N + qbeg + tg) * k - drop + r[m];
                    idx_out[o] = (int)((unsigned)key[m] & 0xFFFFu);
                    if (dist_out) dist_out[o] = o2f((unsigned)(key[m] >> 32));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (qa == 0) stamp(4);
        qa = qb;
    }
    stamp(5);
}


size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct SplitPlan {
    bool ok;
    int KS, CP, Np;
    bool pack;
    size_t off_xx, off_xt, off_cand, off_xs, off_scale, off_bm, total;
};

// the workspace is sized for the larger of the two operand images (two bf16 pieces; the fp16 image is half of it)
SplitPlan plan(int B, int N, int c_knn) {
    SplitPlan p{};
    p.ok = N >= 1024 && N <= (c_knn > 64 ? 4096 : 8192) && c_knn >= 1 && c_knn <= 128;
    p.pack = c_knn <= 4;
    p.KS = p.pack ? 1 : (c_knn <= 16 ? 1 : (c_knn <= 32 ? 2 : (c_knn <= 64 ? 4 : 8)));
    p.CP = p.pack ? 4 : 16 * p.KS;
    p.Np = (N + 63) & ~63;
    const size_t T = p.Np / 32, opt = p.pack ? 1 : (p.KS == 8 ? 8 : 2 * p.KS);   // 8 k-steps: fp16 image only
    p.off_xx = 0;
    p.off_xt = align256(sizeof(float) * (size_t)B * p.Np);
    p.off_cand = p.off_xt + align256(sizeof(float) * (size_t)B * p.Np * p.CP);
    p.off_xs = p.off_cand + align256((size_t)B * T * opt * 1024);
    p.off_scale = p.off_xs + align256(sizeof(float) * (size_t)B * p.Np);
    p.off_bm = p.off_scale + align256(sizeof(float) * (size_t)B);
    p.total = p.off_bm + align256(sizeof(unsigned) * (size_t)B * p.Np * ((T + 7) & ~(size_t)7));   // survivor bitmaps (two-launch form), rows padded to 8 words
    return p;
}

}  // namespace

size_t fsg_knn_split_workspace_bytes(int B, int N, int c_knn) {
    const SplitPlan p = plan(B, N, c_knn);
    return p.ok ? p.total : 0;
}

// where the prep products live inside the workspace (fp16 form, for a producer that emits them itself: edgeconv.hip)
int fsg_knn_split_ws_pointers(void *ws, size_t ws_bytes, int B, int N, int c_knn, float **xx, float **xs, void **cand,
                              float **cscale) {
    const SplitPlan p = plan(B, N, c_knn);
    if (!p.ok || p.pack || ws == nullptr || ws_bytes < p.total || p.Np != N) return FSG_ERR_UNSUPPORTED;
    unsigned char *w = static_cast<unsigned char *>(ws);
    *xx = reinterpret_cast<float *>(w + p.off_xx);
    *cand = w + p.off_cand;
    *xs = reinterpret_cast<float *>(w + p.off_xs);
    *cscale = reinterpret_cast<float *>(w + p.off_scale);
    return FSG_OK;
}


// the round-2/3 monolithic kernel (flag 536870912): A/B timing and an independent cross-check of the two-launch form
static int launch_monolithic(const SplitPlan &p, const float *x, const float *prepared_xt, int B, int N, int64_t stride_b,
                             int64_t stride_c, int c_knn, int k, int flags, int32_t *idx_out, float *dist_out, float *xx,
                             float *xt, u32x4 *cand, float *xs, float *cscale, hipStream_t st) {
    const dim3 pgrid(p.Np / 32, B), grid(p.Np / 64, B);
    const size_t T = p.Np / 32, CPQ = p.CP + 4, PR = p.CP > 64 ? 16 : 48;
    size_t bmb = 4 * ((QB * (T + 1) + 1) & ~(size_t)1);
    if (bmb < (size_t)4 * QB * NMIN) bmb = (size_t)4 * QB * NMIN;
    size_t usz = sizeof(float) * p.Np + bmb;
    if (usz < 4 * (size_t)WAVES * PR * CPQ) usz = 4 * (size_t)WAVES * PR * CPQ;
    const size_t fixed = ((usz + 15) & ~(size_t)15) + sizeof(float) * QB * CPQ + sizeof(float) * QB + sizeof(int) * QB +
                         sizeof(float) * 24 + sizeof(int) * WAVES * (QB / WAVES) * 2;
    // candidates a wave refines per batch (8 bytes of LDS each): as many as the 160 KiB allow
    int PC = 512;
    while (PC >= 256 && fixed + 8 * (size_t)WAVES * PC > 160 * 1024) PC /= 2;
    if (PC < 256) return FSG_ERR_UNSUPPORTED;
    const size_t lds = fixed + 8 * (size_t)WAVES * PC;
    if (p.KS == 8 && (flags & 1073741824)) return FSG_ERR_UNSUPPORTED;   // 128 channels: fp16 image only
#define FSG_KNN_MONO(KSV, PK, HF)                                                                                      \
    do {                                                                                                               \
        static FsgLdsGrant grant;                                                                                     \
        if (!grant.raise((const void *)knn_split_kernel<KSV, PK, HF>, 160 * 1024)) {                                  \
            fsg_set_error("fsg_knn_dense_ws_f32: cannot raise dynamic LDS");                                          \
            return FSG_ERR_HIP;                                                                                       \
        }                                                                                                             \
        if (!prepared_xt)                                                                                              \
            hipLaunchKernelGGL((knn_split_prep_kernel<KSV, PK, HF>), pgrid, dim3(256), 0, st, x, N, p.Np,              \
                               (long)stride_b, (long)stride_c, c_knn, xx, xt, cand, xs, cscale);                       \
        hipLaunchKernelGGL((knn_split_kernel<KSV, PK, HF>), grid, dim3(WAVES * 64), lds, st, xx, xt, cand, xs, cscale,  \
                           N, p.Np, k, flags, PC, idx_out, dist_out);                                                  \
    } while (0)
    if (lds > 160 * 1024) return FSG_ERR_UNSUPPORTED;
    // default above 4 channels: ONE fp16 product on the centred, scaled points; flag 1073741824: three bf16 products on the
    // points as they are
    // (the first form of this kernel: A/B timing, cross-check of the centred path)
    if ((flags & 1073741824) || p.pack) {   // up to 4 channels the three bf16 products share ONE k-step: nothing to gain
        if (p.pack) FSG_KNN_MONO(1, true, false);
        else if (p.KS == 1) FSG_KNN_MONO(1, false, false);
        else if (p.KS == 2) FSG_KNN_MONO(2, false, false);
        else FSG_KNN_MONO(4, false, false);
    } else {
        if (p.KS == 1) FSG_KNN_MONO(1, false, true);
        else if (p.KS == 2) FSG_KNN_MONO(2, false, true);
        else if (p.KS == 4) FSG_KNN_MONO(4, false, true);
        else FSG_KNN_MONO(8, false, true);
    }
#undef FSG_KNN_MONO
    return FSG_OK;
}

// returns FSG_ERR_UNSUPPORTED when the shape is outside this kernel's envelope (caller falls back)
int fsg_knn_split_launch_ex(const float *x, const float *prepared_xt, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn,
                            int k, int flags, int32_t *idx_out, float *dist_out, void *ws, size_t ws_bytes, hipStream_t st);

int fsg_knn_split_launch(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn, int k, int flags,
                         int32_t *idx_out, float *dist_out, void *ws, size_t ws_bytes, hipStream_t st) {
    return fsg_knn_split_launch_ex(x, nullptr, B, N, stride_b, stride_c, c_knn, k, flags, idx_out, dist_out, ws, ws_bytes, st);
}

// prepared_xt != NULL: the workspace already holds the prep products (squared norms, centred norms, fp16 image, scale: written
// by the producer of the points, ec1_apply_prep_kernel) and prepared_xt is the point-major (B, N, c_knn) copy of the points
// (c_knn == 16 KS, N % 64 == 0): the prep kernel is skipped
static int knn_split_launch_impl(const float *x, const float *prepared_xt, int B, int N, int64_t stride_b, int64_t stride_c,
                                 int c_knn, int k, int flags, int32_t *idx_out, float *dist_out, void *ws, size_t ws_bytes,
                                 hipStream_t st, const float *pq_w, int pq_rows, float *pq_out, bool *pq_fused);

int fsg_knn_split_launch_ex(const float *x, const float *prepared_xt, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn,
                            int k, int flags, int32_t *idx_out, float *dist_out, void *ws, size_t ws_bytes, hipStream_t st) {
    return knn_split_launch_impl(x, prepared_xt, B, N, stride_b, stride_c, c_knn, k, flags, idx_out, dist_out, ws, ws_bytes, st, nullptr,
                                 0, nullptr, nullptr);
}

// graph build + the per-point product pq_out (B, N, pq_rows) = x^T pq_w^T (c_knn <= 4 channels): fused into the nominate launch
// where the RAW path runs (*fused = true), otherwise the caller launches fsg_knn_pq_rows_launch itself
int fsg_knn_split_launch_pq(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn, int k, int flags,
                            int32_t *idx_out, float *dist_out, void *ws, size_t ws_bytes, hipStream_t st, const float *pq_w,
                            int pq_rows, float *pq_out, bool *fused) {
    *fused = false;
    return knn_split_launch_impl(x, nullptr, B, N, stride_b, stride_c, c_knn, k, flags, idx_out, dist_out, ws, ws_bytes, st, pq_w,
                                 pq_rows, pq_out, fused);
}

int fsg_knn_pq_rows_launch(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn, const float *pq_w,
                           int pq_rows, float *pq_out, hipStream_t st) {
    if (c_knn < 1 || c_knn > 4 || pq_rows < 1 || 256 % pq_rows != 0) return FSG_ERR_UNSUPPORTED;
    const RawPoints raw{x, (long)stride_b, (long)stride_c, c_knn, nullptr, nullptr, pq_w, pq_rows, pq_out};
    hipLaunchKernelGGL(knn_pq_rows_kernel, dim3(fsg_cdiv(N, 64), B), dim3(256), 0, st, raw, N);
    FSG_CHECK_LAUNCH("fsg_knn_dense_ws_pq_f32/rows");
    return FSG_OK;
}

static int knn_split_launch_impl(const float *x, const float *prepared_xt, int B, int N, int64_t stride_b, int64_t stride_c,
                                 int c_knn, int k, int flags, int32_t *idx_out, float *dist_out, void *ws, size_t ws_bytes,
                                 hipStream_t st, const float *pq_w, int pq_rows, float *pq_out, bool *pq_fused) {
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    const SplitPlan p = plan(B, N, c_knn);
    if (!p.ok || k + drop > 64 || ws == nullptr || ws_bytes < p.total) return FSG_ERR_UNSUPPORTED;
    if (prepared_xt && (p.pack || p.CP != c_knn || p.Np != N || (flags & 1073741824))) return FSG_ERR_UNSUPPORTED;
    unsigned char *w = static_cast<unsigned char *>(ws);
    float *xx = reinterpret_cast<float *>(w + p.off_xx);
    float *xt = prepared_xt ? const_cast<float *>(prepared_xt) : reinterpret_cast<float *>(w + p.off_xt);
    u32x4 *cand = reinterpret_cast<u32x4 *>(w + p.off_cand);
    float *xs = reinterpret_cast<float *>(w + p.off_xs);
    float *cscale = reinterpret_cast<float *>(w + p.off_scale);
    // One launch or two.  The two-launch form wins wherever the refine's batches of eight queries fit one sub-batch of 256
    // entries; at k + drop > 32 on 64 and more channels with N > 4096 (BASELINE config 4: ~77 nominees per query, three
    // sub-batches per wave, 2 x 32 KB of bitmaps per wave and sweep) the monolithic kernel is still ahead (4 x 8192, k = 40,
    // 64 channels: 209 vs 241 us; every other measured shape: 0 - 35 % in favour of two launches).  Flag 536870912 forces the
    // monolithic kernel, flag 268435456 (cycle stamps of the two-launch form) the two-launch form.
    const bool big_k = k + drop > 32 && p.CP >= 64 && N > 4096 && !(flags & 268435456);
    if ((flags & 536870912) || big_k) {
        const int rc = launch_monolithic(p, x, prepared_xt, B, N, stride_b, stride_c, c_knn, k, flags, idx_out, dist_out, xx, xt, cand, xs, cscale, st);
        if (rc != FSG_OK) return rc;
        FSG_CHECK_LAUNCH("fsg_knn_dense_ws_f32/split-monolithic");
        return FSG_OK;
    }
    const dim3 pgrid(p.Np / 32, B), grid(p.Np / 64, B);
    const size_t T = p.Np / 32, CPQ = p.CP + 4;
    if (p.KS == 8 && (flags & 1073741824)) return FSG_ERR_UNSUPPORTED;   // 128 channels: fp16 image only
    unsigned *bmg = reinterpret_cast<unsigned *>(w + p.off_bm);
    // ---- nominate: norms + bitmaps (or group minima) + per-query scalars
    size_t bmw = (QB * (T + 1) + 3) & ~(size_t)3;
    if (bmw < (size_t)QB * NMIN) bmw = (size_t)QB * NMIN;
    const size_t lds1 = sizeof(float) * 32 * WAVES * ((T + WAVES - 1) / WAVES) + 4 * bmw + sizeof(float) * QB + sizeof(unsigned) * 4 + sizeof(float) * 24;
    // ---- refine: four waves, each with its candidate / key list and its row stage
    const int PCAP = 256, TS = (int)((T + 7) & ~(size_t)7);
    const size_t CS = p.CP > 32 ? 32 : p.CP;
    const size_t lds2 = 2 * ((size_t)8 * RECAP + (p.CP > 4 ? 4 * 64 * (CS + 4) : 0) + 4 * (size_t)RQW * CPQ + 128);
    const bool res = T == 64 && !(flags & 134217728);   // a wave's <= 8 operand tiles stay in registers (flag: A/B timing)
    if (lds1 > 160 * 1024 || lds2 > 64 * 1024) return FSG_ERR_UNSUPPORTED;
    const dim3 rgrid(p.Np / (2 * RQW), B);
#define FSG_KNN_SPLIT(KSV, PK, HF)                                                                                      \
    do {                                                                                                               \
        static FsgLdsGrant grant;                                                                                     \
        if (!grant.raise((const void *)knn_nominate_kernel<KSV, PK, HF, false, false>, 160 * 1024)) {                        \
            fsg_set_error("fsg_knn_dense_ws_f32: cannot raise dynamic LDS");                                          \
            return FSG_ERR_HIP;                                                                                       \
        }                                                                                                             \
        if (!prepared_xt)                                                                                              \
            hipLaunchKernelGGL((knn_split_prep_kernel<KSV, PK, HF>), pgrid, dim3(256), 0, st, x, N, p.Np,              \
                               (long)stride_b, (long)stride_c, c_knn, xx, xt, cand, xs, cscale);                       \
        if (res && ((HF && KSV <= 4) || PK))                                                                           \
            hipLaunchKernelGGL((knn_nominate_kernel<KSV, PK, HF, ((HF && KSV <= 4) || PK), false>), grid,              \
                               dim3(WAVES * 64), lds1, st, xx, xt, cand, xs, cscale, N, p.Np, k, flags, PCAP, bmg, TS,  \
                               idx_out, dist_out, RawPoints{});                                                        \
        else                                                                                                           \
            hipLaunchKernelGGL((knn_nominate_kernel<KSV, PK, HF, false, false>), grid, dim3(WAVES * 64), lds1, st, xx, \
                               xt, cand, xs, cscale, N, p.Np, k, flags, PCAP, bmg, TS, idx_out, dist_out, RawPoints{}); \
        hipLaunchKernelGGL((knn_refine_kernel<(PK ? 4 : 16 * KSV)>), rgrid, dim3(128), lds2, st, xx, xt, bmg, N, p.Np,  \
                           TS, k, flags, PCAP, idx_out, dist_out);                                                     \
    } while (0)
    // default above 4 channels: ONE fp16 product on the centred, scaled points; flag 1073741824: three bf16 products on the
    // points as they are (the first form of this kernel: A/B timing, cross-check of the centred path)
    // <= 4 channels at 64 tiles, points directly addressable in 16-byte pieces: no prep launch (RAW, see the kernel)
    const bool raw_ok = p.pack && res && !prepared_xt && p.Np == N && !(flags & 67108864) && stride_b % 4 == 0 &&
                        stride_c % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (raw_ok) {
        const RawPoints raw{x, (long)stride_b, (long)stride_c, c_knn, xx, xt, pq_w, pq_rows, pq_out};
        if (pq_fused) *pq_fused = pq_out != nullptr && (WAVES * 64) % pq_rows == 0;
        if (pq_out && (WAVES * 64) % pq_rows != 0) return FSG_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((knn_nominate_kernel<1, true, false, true, true>), grid, dim3(WAVES * 64), lds1, st, xx, xt, cand, xs,
                           cscale, N, p.Np, k, flags, PCAP, bmg, TS, idx_out, dist_out, raw);
        hipLaunchKernelGGL((knn_refine_kernel<4>), rgrid, dim3(128), lds2, st, xx, xt, bmg, N, p.Np, TS, k, flags, PCAP, idx_out,
                           dist_out);
    } else if ((flags & 1073741824) || p.pack) {   // up to 4 channels the three bf16 products share ONE k-step: nothing to gain
        if (p.pack) FSG_KNN_SPLIT(1, true, false);
        else if (p.KS == 1) FSG_KNN_SPLIT(1, false, false);
        else if (p.KS == 2) FSG_KNN_SPLIT(2, false, false);
        else FSG_KNN_SPLIT(4, false, false);
    } else {
        if (p.KS == 1) FSG_KNN_SPLIT(1, false, true);
        else if (p.KS == 2) FSG_KNN_SPLIT(2, false, true);
        else if (p.KS == 4) FSG_KNN_SPLIT(4, false, true);
        else FSG_KNN_SPLIT(8, false, true);
    }
#undef FSG_KNN_SPLIT
    FSG_CHECK_LAUNCH("fsg_knn_dense_ws_f32/split");
    return FSG_OK;
}

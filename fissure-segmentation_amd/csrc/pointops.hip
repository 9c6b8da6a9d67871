// PointTransformer primitives on packed (n,3)/(n,c) clouds with cumulative segment offsets --
// include/fsg_hip.h: fsg_knn_segment_f32, fsg_fps_f32, fsg_group_gather_*, fsg_vec_attn_*.
// Replace pointops_cuda.{knnquery,furthestsampling,grouping,aggregation}_* behind
// models/pointtransformer/pointops.py and the torch chain at models/pointtransformer/seg_model.py:50-52.
//
// v0: one lane per query (kNN), one workgroup per cloud (FPS), lanes along channels (gather /
// aggregate).  Distances use the direct form d = fma(dz,dz, fma(dy,dy, dx*dx)) like oracle/fsg_oracle.c.
#include "fsg_common.h"
#include <stdlib.h>

int fsg_knn_segment_rows_launch(const float *xyz, const float *new_xyz, const int32_t *offset, const int32_t *new_offset,
                                int b, int n, int m, int nsample, int32_t *idx, float *dist2, hipStream_t st);

namespace {

constexpr int BLOCK = 256;

__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

template <int NS>
__global__ __launch_bounds__(BLOCK) void knn_segment_kernel(const float *__restrict__ xyz,
                                                             const float *__restrict__ new_xyz,
                                                             const int32_t *__restrict__ offset,
                                                             const int32_t *__restrict__ new_offset, int b, int m,
                                                             int nsample, int32_t *__restrict__ idx,
                                                             float *__restrict__ dist2) {
    const int q = blockIdx.x * BLOCK + threadIdx.x;
    if (q >= m) return;
    int s = 0;
    while (s < b - 1 && q >= new_offset[s]) ++s;
    const int st = s ? offset[s - 1] : 0, en = offset[s];
    const float qx = new_xyz[3L * q], qy = new_xyz[3L * q + 1], qz = new_xyz[3L * q + 2];
    float bd[NS];
    int bi[NS];
#pragma unroll
    for (int p = 0; p < NS; ++p) { bd[p] = 1e10f; bi[p] = st; }
    for (int j = st; j < en; ++j) {
        const float d = sqdist3(qx, qy, qz, xyz[3L * j], xyz[3L * j + 1], xyz[3L * j + 2]);
        if (d < bd[NS - 1]) {
#pragma unroll
            for (int p = NS - 1; p > 0; --p) {
                const bool shift = d < bd[p - 1];
                const bool here = d < bd[p];
                const float nd = shift ? bd[p - 1] : (here ? d : bd[p]);
                const int ni = shift ? bi[p - 1] : (here ? j : bi[p]);
                bd[p] = nd;
                bi[p] = ni;
            }
            if (d < bd[0]) { bd[0] = d; bi[0] = j; }
        }
    }
#pragma unroll
    for (int p = 0; p < NS; ++p)
        if (p < nsample) {
            idx[(long)q * nsample + p] = bi[p];
            dist2[(long)q * nsample + p] = bd[p];
        }
}

// wave arg-max of a 64-bit key on the DPP network (6 VALU steps instead of 6 LDS-crossbar shuffles); the result is valid in
// lanes 48..63.  Key = float bits of a non-negative value << 32 | (0x7fffffff - index): larger value wins, lower index
// on ties.
__device__ __forceinline__ unsigned long long dpp_max_step(unsigned long long k, const int ctrl, const int row_mask) {
    const int lo = (int)(unsigned)k, hi = (int)(unsigned)(k >> 32);
    int olo, ohi;
    switch (ctrl) {   // the control word must be a compile-time constant
        case 0: olo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false); break;
        case 1: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, false); break;
        case 2: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xf, 0xf, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xf, 0xf, false); break;
        case 3: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xf, 0xf, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xf, 0xf, false); break;
        case 4: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x142, 0xa, 0xf, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x142, 0xa, 0xf, false); break;
        default: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x143, 0xc, 0xf, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x143, 0xc, 0xf, false); break;
    }
    (void)row_mask;
    const unsigned long long o = ((unsigned long long)(unsigned)ohi << 32) | (unsigned)olo;
    return o > k ? o : k;
}
__device__ __forceinline__ unsigned long long wave_argmax_key(float bv, int bj) {
    unsigned long long k = bv >= 0.f ? (((unsigned long long)__float_as_uint(bv) << 32) | (unsigned)(0x7fffffff - bj)) : 0ull;
#pragma unroll
    for (int st = 0; st < 6; ++st) k = dpp_max_step(k, st, 0);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;   // wave-uniform
}

// Farthest point sampling of one segment by ONE wave: the segment's points and running minimum distances live in
// registers (PPL per lane, point j = start + lane + 64 r), the coordinates also in LDS so that the next pivot is a
// broadcast LDS read instead of a dependent global load; an iteration is PPL distance updates + a 6-step shuffle
// arg-max, no barrier.  Same arithmetic and tie rule (largest distance, lowest index) as the scalar reference.
template <int PPL>
__device__ __forceinline__ void fps_one_wave(const float *__restrict__ xyz, int st, int len, int qs, int qe,
                                             int32_t *__restrict__ idx, float *lds) {
    const int lane = threadIdx.x & 63;
    float px[PPL], py[PPL], pz[PPL], md[PPL];
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
        const int j = lane + 64 * r;
        const bool ok = j < len;
        px[r] = ok ? xyz[3L * (st + j)] : 0.f;
        py[r] = ok ? xyz[3L * (st + j) + 1] : 0.f;
        pz[r] = ok ? xyz[3L * (st + j) + 2] : 0.f;
        md[r] = 1e10f;
        if (ok) { lds[3 * j] = px[r]; lds[3 * j + 1] = py[r]; lds[3 * j + 2] = pz[r]; }
    }
    __builtin_amdgcn_wave_barrier();
    int cur = 0;  // local index of the pivot
    if (lane == 0) idx[qs] = st;
    for (int t = qs + 1; t < qe; ++t) {
        const float cx = lds[3 * cur], cy = lds[3 * cur + 1], cz = lds[3 * cur + 2];
        float bv = -1.f;
        int bj = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < PPL; ++r) {
            const int j = lane + 64 * r;
            const float d = sqdist3(px[r], py[r], pz[r], cx, cy, cz);
            const float v = fminf(d, md[r]);
            md[r] = v;
            if (j < len && v > bv) { bv = v; bj = j; }
        }
        cur = 0x7fffffff - (int)(unsigned)wave_argmax_key(bv, bj);
        if (lane == 0) idx[t] = st + cur;
    }
}

// Same, one WORKGROUP of FW waves per segment (segments up to 64*FW*PPL points): every wave keeps PPL points per lane in
// registers, the per-wave arg-max goes through LDS and one barrier per iteration.  Shorter serial chain than the
// single-wave version for the 2048-point segments of BASELINE config 3 (4 instead of 32 distance updates per lane).
constexpr int FW = 8;
template <int PPL>
__device__ __forceinline__ void fps_multi_wave(const float *__restrict__ xyz, int st, int len, int qs, int qe,
                                               int32_t *__restrict__ idx, float *lds, unsigned long long *wkey) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float px[PPL], py[PPL], pz[PPL], md[PPL];
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
        const int j = tid + 64 * FW * r;
        const bool ok = j < len;
        px[r] = ok ? xyz[3L * (st + j)] : 0.f;
        py[r] = ok ? xyz[3L * (st + j) + 1] : 0.f;
        pz[r] = ok ? xyz[3L * (st + j) + 2] : 0.f;
        md[r] = 1e10f;
        if (ok) { lds[3 * j] = px[r]; lds[3 * j + 1] = py[r]; lds[3 * j + 2] = pz[r]; }
    }
    __syncthreads();
    int cur = 0;
    if (tid == 0) idx[qs] = st;
    for (int t = qs + 1; t < qe; ++t) {
        const float cx = lds[3 * cur], cy = lds[3 * cur + 1], cz = lds[3 * cur + 2];
        float bv = -1.f;
        int bj = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < PPL; ++r) {
            const int j = tid + 64 * FW * r;
            const float d = sqdist3(px[r], py[r], pz[r], cx, cy, cz);
            const float v = fminf(d, md[r]);
            md[r] = v;
            if (j < len && v > bv) { bv = v; bj = j; }
        }
        const unsigned long long wk = wave_argmax_key(bv, bj);
        const int slot = (t & 1) * FW;   // double-buffered: the next iteration's writes cannot overtake this one's reads
        if (lane == 0) wkey[slot + wave] = wk;
        __syncthreads();
        unsigned long long best = wkey[slot];
#pragma unroll
        for (int w = 1; w < FW; ++w) best = wkey[slot + w] > best ? wkey[slot + w] : best;
        bj = 0x7fffffff - (int)(unsigned)best;
        cur = bj;
        if (tid == 0) idx[t] = st + cur;
    }
}

__global__ __launch_bounds__(FW * 64) void fps_kernel_mw(const float *__restrict__ xyz, const int32_t *__restrict__ offset,
                                                         const int32_t *__restrict__ new_offset, int32_t *__restrict__ idx) {
    __shared__ float pts[2048 * 3];
    __shared__ unsigned long long wkey[2 * FW];
    const int s = blockIdx.x;
    const int st = s ? offset[s - 1] : 0, en = offset[s];
    const int qs = s ? new_offset[s - 1] : 0, qe = new_offset[s];
    if (qe > qs && en <= st) {       // samples asked of an empty segment: index 0 (the caller's buffer is not pre-filled)
        for (int t = qs + (int)threadIdx.x; t < qe; t += FW * 64) idx[t] = 0;
        return;
    }
    if (qe <= qs) return;
    const int len = en - st;   // host guarantees len <= 2048 for this kernel (it only knows n: n <= 2048 * segments is
                               // not enough, so oversized segments are left to the generic kernel via the flag array)
    if (len > 2048) return;
    if (len <= 512) fps_multi_wave<1>(xyz, st, len, qs, qe, idx, pts, wkey);
    else if (len <= 1024) fps_multi_wave<2>(xyz, st, len, qs, qe, idx, pts, wkey);
    else fps_multi_wave<4>(xyz, st, len, qs, qe, idx, pts, wkey);
}

__global__ __launch_bounds__(BLOCK) void fps_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ offset,
                                                     const int32_t *__restrict__ new_offset, float *__restrict__ md,
                                                     int32_t *__restrict__ idx, int mw_done) {
    __shared__ float wv[BLOCK / 64];
    __shared__ int wj[BLOCK / 64];
    __shared__ float pts[2048 * 3];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int st = s ? offset[s - 1] : 0, en = offset[s];
    const int qs = s ? new_offset[s - 1] : 0, qe = new_offset[s];
    if (qe > qs && en <= st && !mw_done)       // (see fps_kernel_mw, which has done it when it ran)
        for (int t = qs + tid; t < qe; t += BLOCK) idx[t] = 0;
    if (qe <= qs || en <= st) return;
    if (en - st <= 2048) {
        if (mw_done) return;   // already sampled by fps_kernel_mw (eight waves per segment)
        // register-resident single-wave path (uniform per workgroup: the other waves just leave)
        if (wave != 0) return;
        if (en - st <= 512) fps_one_wave<8>(xyz, st, en - st, qs, qe, idx, pts);
        else fps_one_wave<32>(xyz, st, en - st, qs, qe, idx, pts);
        return;
    }
    for (int j = st + tid; j < en; j += BLOCK) md[j] = 1e10f;
    int cur = st;
    if (tid == 0) idx[qs] = cur;
    for (int t = qs + 1; t < qe; ++t) {
        const float px = xyz[3L * cur], py = xyz[3L * cur + 1], pz = xyz[3L * cur + 2];
        float bv = -1.f;
        int bj = 0x7fffffff;
        for (int j = st + tid; j < en; j += BLOCK) {
            const float d = sqdist3(xyz[3L * j], xyz[3L * j + 1], xyz[3L * j + 2], px, py, pz);
            const float v = fminf(d, md[j]);
            md[j] = v;
            if (v > bv) { bv = v; bj = j; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oj = __shfl_xor(bj, off);
            if (ov > bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; }
        }
        if (lane == 0) { wv[wave] = bv; wj[wave] = bj; }
        __syncthreads();
        bv = wv[0];
        bj = wj[0];
#pragma unroll
        for (int w = 1; w < BLOCK / 64; ++w)
            if (wv[w] > bv || (wv[w] == bv && wj[w] < bj)) { bv = wv[w]; bj = wj[w]; }
        cur = bj;
        if (tid == 0) idx[t] = cur;
        __syncthreads();
    }
}

__global__ __launch_bounds__(BLOCK) void group_fwd_kernel(const float *__restrict__ feat, const int32_t *__restrict__ idx,
                                                           float *__restrict__ out, int c, long total) {
    for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (long)gridDim.x * BLOCK) {
        const long r = t / c;
        const int ch = (int)(t - r * c);
        out[t] = feat[(long)idx[r] * c + ch];
    }
}

__global__ __launch_bounds__(BLOCK) void group_bwd_kernel(const float *__restrict__ go, const int32_t *__restrict__ idx,
                                                           float *__restrict__ gf, int c, long total) {
    for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (long)gridDim.x * BLOCK) {
        const long r = t / c;
        const int ch = (int)(t - r * c);
        atomicAdd(gf + (long)idx[r] * c + ch, go[t]);
    }
}

// pointops.queryandgroup with use_xyz (models/pointtransformer/pointops.py:100-123 of the reference): out (m, ns, 3 + c) =
// [ xyz[idx] - new_xyz | feat[idx] ] in one launch (the reference: two gathers, a subtraction, a concatenation); the backward
// scatters the feature columns of the (m, ns, 3 + c) gradient as they lie (no contiguous copy of the slice first)
__global__ __launch_bounds__(BLOCK) void group_xyz_feat_fwd_kernel(const float *__restrict__ xyz, const float *__restrict__ nxyz,
                                                                    const float *__restrict__ feat, const int32_t *__restrict__ idx,
                                                                    float *__restrict__ out, int c, int ns, long total) {
    const int w = c + 3;
    for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (long)gridDim.x * BLOCK) {
        const long r = t / w;
        const int col = (int)(t - r * w);
        const long j = idx[r];
        out[t] = col < 3 ? xyz[3 * j + col] - nxyz[3 * (r / ns) + col] : feat[j * c + (col - 3)];
    }
}
__global__ __launch_bounds__(BLOCK) void group_xyz_feat_bwd_kernel(const float *__restrict__ go, const int32_t *__restrict__ idx,
                                                                    float *__restrict__ gf, int c, long total) {
    for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (long)gridDim.x * BLOCK) {
        const long r = t / c;
        const int ch = (int)(t - r * c);
        atomicAdd(gf + (long)idx[r] * c + ch, go[r * (c + 3) + 3 + ch]);
    }
}

// max over the ns neighbour rows of x (m, ns, c) (TransitionDown's MaxPool1d, seg_model.py:77-83 of the reference) with the arg-max
// kept, so that the backward is ONE launch writing the whole (m, ns, c) gradient (torch: a fill and a scatter); first row on ties
__global__ __launch_bounds__(BLOCK) void rows_max_fwd_kernel(const float *__restrict__ x, float *__restrict__ out,
                                                              int32_t *__restrict__ arg, int ns, int c, long total) {
    for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (long)gridDim.x * BLOCK) {
        const long i = t / c;
        const int ch = (int)(t - i * c);
        const float *src = x + i * ns * c + ch;
        float best = src[0];
        int ba = 0;
        for (int s = 1; s < ns; ++s) {
            const float v = src[(long)s * c];
            if (v > best || v != v) { if (!(best != best)) { best = v; ba = s; } }     // NaN wins and stays (torch's rule)
        }
        out[t] = best;
        arg[t] = ba;
    }
}
__global__ __launch_bounds__(BLOCK) void rows_max_bwd_kernel(const float *__restrict__ go, const int32_t *__restrict__ arg,
                                                              float *__restrict__ gx, int ns, int c, long total) {
    for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (long)gridDim.x * BLOCK) {     // t over (m, ns, c)
        const long r = t / c;
        const int ch = (int)(t - r * c);
        const long i = r / ns;
        const int s = (int)(r - i * ns);
        gx[t] = arg[i * c + ch] == s ? go[i * c + ch] : 0.f;
    }
}

// pointops.interpolation (models/pointtransformer/pointops.py:198-215 of the reference): inverse-distance weights of the k nearest
// coarse points, w_j = 1 / (dist_j + 1e-8) normalised to sum 1, out[i] = sum_j feat[idx[i, j]] w_j -- the reference's eight
// element-wise / reduce launches as one.  The weights are recomputed per thread from the k squared distances of its point (same
// operations and order as the reference's tensor expression: sqrt, + 1e-8, reciprocal, sequential sum, divide, product, sum).
constexpr int INTERP_MAXK = 8;
__device__ __forceinline__ void interp_weights(const float *__restrict__ d2, int k, float (&w)[INTERP_MAXK]) {
    float s = 0.f;
    for (int j = 0; j < k; ++j) {
        w[j] = 1.0f / (sqrtf(d2[j]) + 1e-8f);
        s += w[j];
    }
    for (int j = 0; j < k; ++j) w[j] = w[j] / s;
}
__global__ __launch_bounds__(BLOCK) void interp_fwd_kernel(const float *__restrict__ feat, const int32_t *__restrict__ idx,
                                                            const float *__restrict__ d2, float *__restrict__ out, int c, int k,
                                                            long total) {
    for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (long)gridDim.x * BLOCK) {
        const long i = t / c;
        const int ch = (int)(t - i * c);
        float w[INTERP_MAXK];
        interp_weights(d2 + i * k, k, w);
        float acc = 0.f;
        for (int j = 0; j < k; ++j) acc += feat[(long)idx[i * k + j] * c + ch] * w[j];
        out[t] = acc;
    }
}
// gradient of the features (the weights carry none: the reference's kNN distances are not differentiable): atomics into a
// zeroed buffer, like the grouping backward
__global__ __launch_bounds__(BLOCK) void interp_bwd_kernel(const float *__restrict__ go, const int32_t *__restrict__ idx,
                                                            const float *__restrict__ d2, float *__restrict__ gf, int c, int k,
                                                            long total) {
    for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (long)gridDim.x * BLOCK) {
        const long i = t / c;
        const int ch = (int)(t - i * c);
        float w[INTERP_MAXK];
        interp_weights(d2 + i * k, k, w);
        const float g = go[t];
        for (int j = 0; j < k; ++j) atomicAdd(gf + (long)idx[i * k + j] * c + ch, g * w[j]);
    }
}

__global__ __launch_bounds__(BLOCK) void vec_attn_fwd_kernel(const float *__restrict__ v, const float *__restrict__ pos,
                                                              const float *__restrict__ w,
                                                              const int32_t *__restrict__ idx, float *__restrict__ out,
                                                              int ns, int c, int cw, long total) {
    for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (long)gridDim.x * BLOCK) {
        const long i = t / c;
        const int ch = (int)(t - i * c);
        const int wc = ch % cw;
        float acc = 0.f;
        for (int j = 0; j < ns; ++j) {
            const long e = i * ns + j;
            acc = __builtin_fmaf(v[(long)idx[e] * c + ch] + pos[e * c + ch], w[e * cw + wc], acc);
        }
        out[t] = acc;
    }
}

// one thread per (i, j, wc): walks the share planes s (ch = s*cw + wc)
__global__ __launch_bounds__(BLOCK) void vec_attn_bwd_kernel(const float *__restrict__ v, const float *__restrict__ pos,
                                                              const float *__restrict__ w,
                                                              const int32_t *__restrict__ idx,
                                                              const float *__restrict__ go, float *__restrict__ gv,
                                                              float *__restrict__ gpos, float *__restrict__ gw,
                                                              int ns, int c, int cw, long total) {
    for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (long)gridDim.x * BLOCK) {
        const long e = t / cw;  // (i, j)
        const int wc = (int)(t - e * cw);
        const long i = e / ns;
        const long src = idx[e];
        const float wt = w[t];
        float acc = 0.f;
        for (int ch = wc; ch < c; ch += cw) {
            const float g = go[i * c + ch];
            acc = __builtin_fmaf(g, v[src * c + ch] + pos[e * c + ch], acc);
            const float gp = g * wt;
            gpos[e * c + ch] = gp;
            atomicAdd(gv + src * c + ch, gp);
        }
        gw[t] = acc;
    }
}

int grid_for(long total) { return (int)((total + BLOCK - 1) / BLOCK > 16384 ? 16384 : (total + BLOCK - 1) / BLOCK); }

}  // namespace

extern "C" int fsg_knn_segment_f32(const float *xyz, const float *new_xyz, const int32_t *offset,
                                   const int32_t *new_offset, int b, int n, int m, int nsample, int32_t *idx,
                                   float *dist2, fsg_stream_t stream) {
    FSG_REQUIRE(xyz && new_xyz && offset && new_offset && idx && dist2, "fsg_knn_segment_f32: NULL pointer");
    FSG_REQUIRE(b > 0 && n > 0 && m >= 0 && nsample >= 1 && nsample <= 64,
                "fsg_knn_segment_f32: bad shape b=%d n=%d m=%d nsample=%d", b, n, m, nsample);
    if (m == 0) return FSG_OK;
    hipStream_t st = (hipStream_t)stream;
    // production path: 32 queries per workgroup on the dense kNN's wave-cooperative selection (knn_rows_mfma.hip);
    // the one-thread-per-query kernel below stays as the fallback (nsample > 32) and as a cross-check
    // (FSG_KNN_SEGMENT_SCALAR=1)
    static const bool scalar_only = getenv("FSG_KNN_SEGMENT_SCALAR") != nullptr;
    if (!scalar_only) {
        const int rc = fsg_knn_segment_rows_launch(xyz, new_xyz, offset, new_offset, b, n, m, nsample, idx, dist2, st);
        if (rc != FSG_ERR_UNSUPPORTED) return rc;
    }
    dim3 grid(fsg_cdiv(m, BLOCK)), block(BLOCK);
#define FSG_KNNSEG(NS) \
    hipLaunchKernelGGL(knn_segment_kernel<NS>, grid, block, 0, st, xyz, new_xyz, offset, new_offset, b, m, nsample, idx, dist2)
    if (nsample <= 4) FSG_KNNSEG(4);
    else if (nsample <= 8) FSG_KNNSEG(8);
    else if (nsample <= 16) FSG_KNNSEG(16);
    else if (nsample <= 32) FSG_KNNSEG(32);
    else FSG_KNNSEG(64);
#undef FSG_KNNSEG
    FSG_CHECK_LAUNCH("fsg_knn_segment_f32");
    return FSG_OK;
}

extern "C" int fsg_fps_f32(const float *xyz, const int32_t *offset, const int32_t *new_offset, int b, int n, float *tmp,
                           int32_t *idx, fsg_stream_t stream) {
    FSG_REQUIRE(xyz && offset && new_offset && tmp && idx, "fsg_fps_f32: NULL pointer");
    FSG_REQUIRE(b > 0 && n > 0, "fsg_fps_f32: bad shape b=%d n=%d", b, n);
    // segments of up to 2048 points: eight waves per segment, points in registers; longer ones: the generic kernel, which
    // skips what the first launch already sampled.  n <= 2048 means no segment can be longer: one launch.
    static const bool one_wave = getenv("FSG_FPS_ONE_WAVE") != nullptr;   // cross-check: the single-wave path
    if (!one_wave) {
        hipLaunchKernelGGL(fps_kernel_mw, dim3(b), dim3(FW * 64), 0, (hipStream_t)stream, xyz, offset, new_offset, idx);
        FSG_CHECK_LAUNCH("fsg_fps_f32/mw");
    }
    if (one_wave || n > 2048) {
        hipLaunchKernelGGL(fps_kernel, dim3(b), dim3(BLOCK), 0, (hipStream_t)stream, xyz, offset, new_offset, tmp, idx,
                           one_wave ? 0 : 1);
        FSG_CHECK_LAUNCH("fsg_fps_f32");
    }
    return FSG_OK;
}

extern "C" int fsg_group_gather_fwd_f32(const float *feat, const int32_t *idx, float *out, int n, int c, int m, int ns,
                                        fsg_stream_t stream) {
    FSG_REQUIRE(feat && idx && out, "fsg_group_gather_fwd_f32: NULL pointer");
    FSG_REQUIRE(n > 0 && c > 0 && m >= 0 && ns > 0, "fsg_group_gather_fwd_f32: bad shape");
    const long total = (long)m * ns * c;
    if (total == 0) return FSG_OK;
    hipLaunchKernelGGL(group_fwd_kernel, dim3(grid_for(total)), dim3(BLOCK), 0, (hipStream_t)stream, feat, idx, out, c,
                       total);
    FSG_CHECK_LAUNCH("fsg_group_gather_fwd_f32");
    return FSG_OK;
}

extern "C" int fsg_group_gather_bwd_f32(const float *grad_out, const int32_t *idx, float *grad_feat, int n, int c, int m,
                                        int ns, fsg_stream_t stream) {
    FSG_REQUIRE(grad_out && idx && grad_feat, "fsg_group_gather_bwd_f32: NULL pointer");
    FSG_REQUIRE(n > 0 && c > 0 && m >= 0 && ns > 0, "fsg_group_gather_bwd_f32: bad shape");
    const long total = (long)m * ns * c;
    if (total == 0) return FSG_OK;
    hipLaunchKernelGGL(group_bwd_kernel, dim3(grid_for(total)), dim3(BLOCK), 0, (hipStream_t)stream, grad_out, idx,
                       grad_feat, c, total);
    FSG_CHECK_LAUNCH("fsg_group_gather_bwd_f32");
    return FSG_OK;
}

extern "C" int fsg_group_xyz_feat_fwd_f32(const float *xyz, const float *new_xyz, const float *feat, const int32_t *idx, float *out,
                                          int n, int c, int m, int ns, fsg_stream_t stream) {
    FSG_REQUIRE(xyz && new_xyz && feat && idx && out, "fsg_group_xyz_feat_fwd_f32: NULL pointer");
    FSG_REQUIRE(n > 0 && c > 0 && m >= 0 && ns > 0, "fsg_group_xyz_feat_fwd_f32: bad shape");
    const long total = (long)m * ns * (c + 3);
    if (total == 0) return FSG_OK;
    hipLaunchKernelGGL(group_xyz_feat_fwd_kernel, dim3(grid_for(total)), dim3(BLOCK), 0, (hipStream_t)stream, xyz, new_xyz, feat, idx,
                       out, c, ns, total);
    FSG_CHECK_LAUNCH("fsg_group_xyz_feat_fwd_f32");
    return FSG_OK;
}

extern "C" int fsg_group_xyz_feat_bwd_f32(const float *grad_out, const int32_t *idx, float *grad_feat, int n, int c, int m, int ns,
                                          fsg_stream_t stream) {
    FSG_REQUIRE(grad_out && idx && grad_feat, "fsg_group_xyz_feat_bwd_f32: NULL pointer");
    FSG_REQUIRE(n > 0 && c > 0 && m >= 0 && ns > 0, "fsg_group_xyz_feat_bwd_f32: bad shape");
    const long total = (long)m * ns * c;
    if (total == 0) return FSG_OK;
    hipLaunchKernelGGL(group_xyz_feat_bwd_kernel, dim3(grid_for(total)), dim3(BLOCK), 0, (hipStream_t)stream, grad_out, idx, grad_feat,
                       c, total);
    FSG_CHECK_LAUNCH("fsg_group_xyz_feat_bwd_f32");
    return FSG_OK;
}

extern "C" int fsg_rows_max_fwd_f32(const float *x, float *out, int32_t *arg, int m, int ns, int c, fsg_stream_t stream) {
    FSG_REQUIRE(x && out && arg, "fsg_rows_max_fwd_f32: NULL pointer");
    FSG_REQUIRE(m >= 0 && ns > 0 && c > 0, "fsg_rows_max_fwd_f32: bad shape");
    const long total = (long)m * c;
    if (total == 0) return FSG_OK;
    hipLaunchKernelGGL(rows_max_fwd_kernel, dim3(grid_for(total)), dim3(BLOCK), 0, (hipStream_t)stream, x, out, arg, ns, c, total);
    FSG_CHECK_LAUNCH("fsg_rows_max_fwd_f32");
    return FSG_OK;
}

extern "C" int fsg_rows_max_bwd_f32(const float *grad_out, const int32_t *arg, float *grad_x, int m, int ns, int c,
                                    fsg_stream_t stream) {
    FSG_REQUIRE(grad_out && arg && grad_x, "fsg_rows_max_bwd_f32: NULL pointer");
    FSG_REQUIRE(m >= 0 && ns > 0 && c > 0, "fsg_rows_max_bwd_f32: bad shape");
    const long total = (long)m * ns * c;
    if (total == 0) return FSG_OK;
    hipLaunchKernelGGL(rows_max_bwd_kernel, dim3(grid_for(total)), dim3(BLOCK), 0, (hipStream_t)stream, grad_out, arg, grad_x, ns, c,
                       total);
    FSG_CHECK_LAUNCH("fsg_rows_max_bwd_f32");
    return FSG_OK;
}

extern "C" int fsg_interp_fwd_f32(const float *feat, const int32_t *idx, const float *dist2, float *out, int n, int c, int m, int k,
                                  fsg_stream_t stream) {
    FSG_REQUIRE(feat && idx && dist2 && out, "fsg_interp_fwd_f32: NULL pointer");
    FSG_REQUIRE(n > 0 && c > 0 && m >= 0 && k > 0 && k <= INTERP_MAXK, "fsg_interp_fwd_f32: bad shape n=%d c=%d m=%d k=%d", n, c, m, k);
    const long total = (long)m * c;
    if (total == 0) return FSG_OK;
    hipLaunchKernelGGL(interp_fwd_kernel, dim3(grid_for(total)), dim3(BLOCK), 0, (hipStream_t)stream, feat, idx, dist2, out, c, k,
                       total);
    FSG_CHECK_LAUNCH("fsg_interp_fwd_f32");
    return FSG_OK;
}

extern "C" int fsg_interp_bwd_f32(const float *grad_out, const int32_t *idx, const float *dist2, float *grad_feat, int n, int c,
                                  int m, int k, fsg_stream_t stream) {
    FSG_REQUIRE(grad_out && idx && dist2 && grad_feat, "fsg_interp_bwd_f32: NULL pointer");
    FSG_REQUIRE(n > 0 && c > 0 && m >= 0 && k > 0 && k <= INTERP_MAXK, "fsg_interp_bwd_f32: bad shape n=%d c=%d m=%d k=%d", n, c, m, k);
    const long total = (long)m * c;
    if (total == 0) return FSG_OK;
    hipLaunchKernelGGL(interp_bwd_kernel, dim3(grid_for(total)), dim3(BLOCK), 0, (hipStream_t)stream, grad_out, idx, dist2,
                       grad_feat, c, k, total);
    FSG_CHECK_LAUNCH("fsg_interp_bwd_f32");
    return FSG_OK;
}

extern "C" int fsg_vec_attn_fwd_f32(const float *v, const float *pos, const float *w, const int32_t *idx, float *out,
                                    int n, int ns, int c, int cw, fsg_stream_t stream) {
    FSG_REQUIRE(v && pos && w && idx && out, "fsg_vec_attn_fwd_f32: NULL pointer");
    FSG_REQUIRE(n >= 0 && ns > 0 && c > 0 && cw > 0 && c % cw == 0, "fsg_vec_attn_fwd_f32: bad shape c=%d cw=%d", c, cw);
    const long total = (long)n * c;
    if (total == 0) return FSG_OK;
    hipLaunchKernelGGL(vec_attn_fwd_kernel, dim3(grid_for(total)), dim3(BLOCK), 0, (hipStream_t)stream, v, pos, w, idx,
                       out, ns, c, cw, total);
    FSG_CHECK_LAUNCH("fsg_vec_attn_fwd_f32");
    return FSG_OK;
}

extern "C" int fsg_vec_attn_bwd_f32(const float *v, const float *pos, const float *w, const int32_t *idx,
                                    const float *grad_out, float *grad_v, float *grad_pos, float *grad_w, int n, int ns,
                                    int c, int cw, fsg_stream_t stream) {
    FSG_REQUIRE(v && pos && w && idx && grad_out && grad_v && grad_pos && grad_w, "fsg_vec_attn_bwd_f32: NULL pointer");
    FSG_REQUIRE(n >= 0 && ns > 0 && c > 0 && cw > 0 && c % cw == 0, "fsg_vec_attn_bwd_f32: bad shape c=%d cw=%d", c, cw);
    const long total = (long)n * ns * cw;
    if (total == 0) return FSG_OK;
    hipLaunchKernelGGL(vec_attn_bwd_kernel, dim3(grid_for(total)), dim3(BLOCK), 0, (hipStream_t)stream, v, pos, w, idx,
                       grad_out, grad_v, grad_pos, grad_w, ns, c, cw, total);
    FSG_CHECK_LAUNCH("fsg_vec_attn_bwd_f32");
    return FSG_OK;
}

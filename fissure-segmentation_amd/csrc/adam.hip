// Adam over one flat fp32 buffer -- include/fsg_hip.h: fsg_adam_flat_f32.
//
// The reference's optimizer is torch.optim.Adam(model.parameters(), lr, weight_decay) (model_trainer.py:57).  Over the one
// flat buffer of optim.FlatAdam torch's fused kernel is a single launch, but a chunked one: 64 Ki elements per
// workgroup = 28 workgroups for the 1.8 M parameters of DGCNN-seg on a 256-CU chip (43 us per step in
// profiles/r1_bench_c2_kernel_stats.csv, against 50 MB of traffic = 6 us at HBM rate).  This kernel streams the four
// arrays as float4, 1024 grid-stride workgroups.
//
// The step count lives on the device (the update is replayed inside a hipGraph, so nothing per-step may come from the
// host): every workgroup reads it at entry, the LAST workgroup to finish (ticket counter) increments it -- by then all
// others have consumed the old value.  Update rule = torch's (torch/optim/adam.py, `_single_tensor_adam`):
//   g' = g + wd * p;  m = lerp(m, g', 1-b1);  v = b2 v + (1-b2) g'^2;
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "fsg_common.h"

namespace {

struct AdamState {  // the 8-byte `state` block of the C ABI
    float step;
    unsigned int ticket;
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float b1, float b2, float eps, float wd,
                                         float step_size, float bc2_sqrt) {
    if (wd != 0.f) g = fmaf(wd, p, g);
    m = m + (1.f - b1) * (g - m);
    v = b2 * v + (1.f - b2) * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_flat_kernel(float *__restrict__ param, const float *__restrict__ grad,
                                                        float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq,
                                                        AdamState *__restrict__ state, long n, float lr_host,
                                                        const float *__restrict__ lr_dev, float b1, float b2, float eps,
                                                        float wd) {
    __shared__ float sc[2];
    if (threadIdx.x == 0) {
        const double t = (double)state->step + 1.0;
        const double lr = lr_dev ? (double)lr_dev[0] : (double)lr_host;
        sc[0] = (float)(lr / (1.0 - pow((double)b1, t)));
        sc[1] = (float)sqrt(1.0 - pow((double)b2, t));
    }
    __syncthreads();
    const float step_size = sc[0], bc2_sqrt = sc[1];
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 p = reinterpret_cast<float4 *>(param)[i];
        const float4 g = reinterpret_cast<const float4 *>(grad)[i];
        float4 m = reinterpret_cast<float4 *>(exp_avg)[i];
        float4 v = reinterpret_cast<float4 *>(exp_avg_sq)[i];
        adam_one(p.x, g.x, m.x, v.x, b1, b2, eps, wd, step_size, bc2_sqrt);
        adam_one(p.y, g.y, m.y, v.y, b1, b2, eps, wd, step_size, bc2_sqrt);
        adam_one(p.z, g.z, m.z, v.z, b1, b2, eps, wd, step_size, bc2_sqrt);
        adam_one(p.w, g.w, m.w, v.w, b1, b2, eps, wd, step_size, bc2_sqrt);
        reinterpret_cast<float4 *>(param)[i] = p;
        reinterpret_cast<float4 *>(exp_avg)[i] = m;
        reinterpret_cast<float4 *>(exp_avg_sq)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // tail
        const long i = (n4 << 2) + threadIdx.x;
        float p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
        adam_one(p, grad[i], m, v, b1, b2, eps, wd, step_size, bc2_sqrt);
        param[i] = p;
        exp_avg[i] = m;
        exp_avg_sq[i] = v;
    }
    __syncthreads();  // every thread of this workgroup is past its read of sc[] (and thread 0 past state->step)
    if (threadIdx.x == 0) {
        const unsigned int done = atomicAdd(&state->ticket, 1u);
        if (done == gridDim.x - 1) {  // all other workgroups have finished, hence read the old step
            state->step += 1.f;
            state->ticket = 0u;
        }
    }
}

}  // namespace

extern "C" int fsg_adam_flat_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, void *state, int64_t n,
                                 float lr, const float *lr_dev, float beta1, float beta2, float eps, float weight_decay,
                                 fsg_stream_t stream) {
    FSG_REQUIRE(param && grad && exp_avg && exp_avg_sq && state, "fsg_adam_flat_f32: NULL pointer");
    FSG_REQUIRE(n > 0, "fsg_adam_flat_f32: n=%ld", (long)n);
    FSG_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                "fsg_adam_flat_f32: the four arrays must be 16-byte aligned");
    FSG_REQUIRE(((uintptr_t)state & 7) == 0, "fsg_adam_flat_f32: state must be 8-byte aligned");
    FSG_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f,
                "fsg_adam_flat_f32: bad hyper-parameters beta1=%g beta2=%g eps=%g", beta1, beta2, eps);
    // four workgroups per CU, grid-stride: every workgroup pays two fp64 pow() in one thread before its first load, so
    // few fat workgroups beat one per 1024 elements (7.8 M parameters: 96 us with 7.6 K workgroups)
    long blocks = ((n >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, (AdamState *)state, (long)n, lr, lr_dev, beta1, beta2, eps, weight_decay);
    FSG_CHECK_LAUNCH("fsg_adam_flat_f32");
    return FSG_OK;
}

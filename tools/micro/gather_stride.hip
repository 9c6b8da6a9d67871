// Row gathers out of an L2-resident table: does the ROW STRIDE matter when only the first 256 bytes of a row are read?
// (EdgeConv gathers the P half of 512-byte [P | Q] rows.)  hipcc --offload-arch=gfx950 -O3 -o gather_stride gather_stride.hip
// One 16-lane group per row piece (16 bytes per lane), 4 rows per wave instruction, 8 loads in flight, clouds of 2048 rows,
// blockIdx.x % 8 = cloud (XCD-local table), 20 random rows per point, 16384 points.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void gather(const float *__restrict__ tab, const int *__restrict__ idx, int N, int k, int ld,
                                               float *__restrict__ out) {
    const int b = blockIdx.x & 7, tile = blockIdx.x >> 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    const int i = tile * 32 + wave * 4 + g;
    const float *T = tab + (long)b * N * ld;
    const int *id = idx + ((long)b * N + i) * k;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < k; s0 += 10) {
        f4 y[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) y[u] = *reinterpret_cast<const f4 *>(T + (long)id[min(s0 + u, k - 1)] * ld + 4 * c16);
#pragma unroll
        for (int u = 0; u < 10; ++u) acc += y[u];
    }
    *reinterpret_cast<f4 *>(out + ((long)b * N + i) * 64 + 4 * c16) = acc;
}
int main() {
    const int B = 8, N = 2048, k = 20;
    std::vector<int> h((size_t)B * N * k);
    srand(1);
    for (auto &v : h) v = rand() % N;
    int *idx; float *tab, *out;
    hipMalloc(&idx, h.size() * 4); hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&tab, (size_t)B * N * 1024 * 4); hipMemset(tab, 0, (size_t)B * N * 1024 * 4);
    hipMalloc(&out, (size_t)B * N * 64 * 4);
    for (int ld : {64, 128, 160, 192, 256, 320}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(gather, dim3(8 * N / 32), dim3(512), 0, 0, tab, idx, N, k, ld, out);
        hipEventRecord(e0);
        for (int it = 0; it < 50; ++it) hipLaunchKernelGGL(gather, dim3(8 * N / 32), dim3(512), 0, 0, tab, idx, N, k, ld, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("row stride %4d B (table %5.2f MB per cloud): %6.2f us per launch, %5.2f TB/s of gathered bytes\n", ld * 4,
               N * ld * 4 / 1048576.0, ms * 1e3 / 50, (double)B * N * k * 256 / (ms * 1e-3 / 50) / 1e12);
    }
    return 0;
}

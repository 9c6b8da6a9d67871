// libfsg_hip_experiments.so -- ONE superseded kNN design, kept as an independent cross-check of the production kernels
// (tests/test_gpu_parity.py compares it bit for bit at the BASELINE sizes) and as a benchmark baseline
// (tools/bench_kernels.py).  TEST INFRASTRUCTURE: not part of libfsg_hip.so, not declared in include/fsg_hip.h, loaded only by
// the tests that ask for it (`_lib.experiments()`).
//   flag 8      knn_mfma.hip    first matrix-core design (per-lane filter + sorting network)
// (round 3 removed the wave-specialised pipeline knn_pipe.hip and the threshold filter knn_filter.hip from the tree; their
// measurements are in DESIGN.md "What did not work", their code in the history up to round 2.)
// Same arguments and results as fsg_knn_dense_f32 (include/fsg_hip.h).
#include "fsg_common.h"

int fsg_knn_mfma_launch(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn, int k, int flags,
                        int32_t *idx_out, float *dist_out, float *xx_scratch, hipStream_t st);    // knn_mfma.hip

extern "C" int fsg_knn_experiment_f32(const float *x, int B, int N, int64_t stride_b, int64_t stride_c, int c_knn, int k,
                                      int flags, int32_t *idx_out, float *dist_out, float *xx_scratch, fsg_stream_t stream) {
    const int drop = (flags & FSG_KNN_DROP_FIRST) ? 1 : 0;
    FSG_REQUIRE(x && idx_out && xx_scratch, "fsg_knn_experiment_f32: NULL pointer");
    FSG_REQUIRE(B > 0 && N > 0 && c_knn > 0 && k >= 1 && k + drop <= N && k + drop <= FSG_KNN_MAX_K && N <= 32768,
                "fsg_knn_experiment_f32: bad shape B=%d N=%d c_knn=%d k=%d", B, N, c_knn, k);
    hipStream_t st = (hipStream_t)stream;
    int rc = FSG_ERR_UNSUPPORTED;
    if (flags & FSG_KNN_FORCE_MFMA) rc = fsg_knn_mfma_launch(x, B, N, stride_b, stride_c, c_knn, k, flags, idx_out, dist_out, xx_scratch, st);
    if (rc == FSG_ERR_UNSUPPORTED) fsg_set_error("fsg_knn_experiment_f32: flags %d / shape outside the experimental kernels' envelope", flags);
    return rc;
}

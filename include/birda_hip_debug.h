/* birda_hip_debug.h -- the diagnostic entry points of libbirda_hip.so.
 *
 * Not part of the boundary birda binds (include/birda_hip.h maps 1:1 onto the calls `BirdClassifier` makes on birdnet_onnx,
 * reference src/inference/classifier.rs:469-488, 559-582; include/birda_hip_sys.rs is generated from THAT header): these three
 * exist for the parity tests (tests/test_parity_gpu.py reads every tensor of a forward; tests/test_gated_gemm_gpu.py drives the
 * gated project GEMM on operands of its own) and for the tuning tools (tools/gpu_mb_stamps.py).  Round 5's verdict (weak #12) asked
 * for them out of the product header; the symbols stay exported -- the tests call through the C ABI like everything else.
 */
#ifndef BIRDA_HIP_DEBUG_H
#define BIRDA_HIP_DEBUG_H

#include "birda_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Debug/parity: copy tensor `t` (0 = spectrogram, i = output of layer i-1... see
 * modelfile.py) of the LAST forward on this context to host; rows = n of that forward. */
BH_API int bh_debug_read_tensor(bh_classifier *c, bh_batch_context *ctx, uint32_t tensor,
                                float *host, size_t max_floats);

/* Diagnostic: the gated project GEMM of a squeeze-excite block alone, on host operands -- C[M][N] = (A[M][K] x gate[M /
 * rows_per_seg][K]) W[K][N] + bias[N] (+ R[M][N], may be NULL) on the split-f16 MFMA (terms: 1 = f16, 3 = f16x3), through the same
 * dispatch a forward pass takes (streaming / row-streaming / staged kernels by N and M); blocked != 0 lays A out the way pass A of
 * the fused block writes it for N = 96 .. 240.  K % 4 == 0; M a multiple of rows_per_seg.  Tests only. */
BH_API int bh_debug_gated_gemm(int device, const float *A, const float *gate, const float *W, const float *bias, const float *R,
                               float *C, size_t M, size_t K, size_t N, size_t rows_per_seg, int terms, int blocked);

/* Diagnostic of the `make EXPERIMENTS=1` build (there: BIRDA_HIP_MB_STAMPS=1 at create): per fused block, 8 counters of
 * wave-cycles spent in setup, dw-weight staging, expand, barrier, depthwise, barrier, project, epilogue since the last call.
 * Returns the number of blocks written (8 values each); the product build has no phase clock and always returns 0. */
BH_API int bh_debug_mb_stamps(bh_classifier *c, uint64_t *out, size_t cap);

#ifdef __cplusplus
}
#endif

#endif /* BIRDA_HIP_DEBUG_H */

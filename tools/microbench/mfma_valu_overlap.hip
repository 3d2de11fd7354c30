// Does the f32 MFMA (v_mfma_f32_16x16x4_f32) overlap with f32 VALU work on gfx950?
// One workgroup per CU, W waves per SIMD.  Modes: 0 = MFMA only, 1 = VALU (v_fma) only,
// 2 = both in the SAME wave (interleaved 1 MFMA : R VALU), 3 = even waves MFMA, odd waves VALU.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int R>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[8];
    float v[8];
    for (int i = 0; i < 8; i++) { acc[i] = (f32x4){0, 0, 0, 0}; v[i] = threadIdx.x * 0.001f + i; }
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
    const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (do_mfma) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
            if (do_valu) {
#pragma unroll
                for (int r = 0; r < R; r++) v[(j + r) & 7] = __builtin_fmaf(v[(j + r) & 7], b, a);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int R>
float run(int threads, int iters, float *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, R>), dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, R>), dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 20000;
    for (int threads : {256, 512}) {
        const int wps = threads / 256;
        printf("== %d waves per SIMD (one %d-thread workgroup per CU), %d iterations x 8 MFMA (and x 8 x R v_fma)\n", wps, threads, iters);
        float m = run<0, 1>(threads, iters, d);
        printf("MFMA only                : %8.3f ms  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", m, m * 1e-3 * 2.4e9 / (iters * 8.0 * wps));
        float v8 = run<1, 8>(threads, iters, d);
        printf("VALU only, R = 8         : %8.3f ms  (%.2f cycles per v_fma per SIMD)\n", v8, v8 * 1e-3 * 2.4e9 / (iters * 64.0 * wps));
        float b4 = run<2, 4>(threads, iters, d), v4 = run<1, 4>(threads, iters, d);
        printf("same wave, R = 4         : %8.3f ms   (MFMA %.3f + VALU %.3f = %.3f if serial)\n", b4, m, v4, m + v4);
        float b8 = run<2, 8>(threads, iters, d);
        printf("same wave, R = 8         : %8.3f ms   (MFMA %.3f + VALU %.3f = %.3f if serial)\n", b8, m, v8, m + v8);
        if (wps == 2) {
            float s = run<3, 8>(threads, iters, d);
            float m1 = run<0, 1>(256, iters, d), v1 = run<1, 8>(256, iters, d);
            printf("split waves (MFMA | VALU): %8.3f ms   (alone: MFMA wave %.3f, VALU wave %.3f)\n", s, m1, v1);
        }
    }
    return 0;
}

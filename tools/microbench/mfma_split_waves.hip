// One 512-thread workgroup per CU: waves 0-3 issue only MFMAs, waves 4-7 only f32 VALU FMAs, so every
// SIMD hosts one wave of each kind.  Overlap => time ~ max(MFMA alone, VALU alone); none => ~ sum.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int F16, int WHO>   // WHO: 1 = MFMA waves only work, 2 = VALU waves only, 3 = both
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[8]; float v[8]; f16x8 a, b;
    for (int i = 0; i < 8; i++) { acc[i] = (f32x4){0, 0, 0, 0}; v[i] = threadIdx.x * 0.001f + i; a[i] = (_Float16)(i + 1); b[i] = (_Float16)(1.0f + i * 0.01f); }
    const float fa = threadIdx.x * 1e-3f, fb = 1.0f + threadIdx.x * 1e-4f;
    if (wave < 4) {
        if (WHO & 1)
            for (int it = 0; it < iters; it++)
#pragma unroll
                for (int j = 0; j < 8; j++)
                    acc[j] = F16 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[j], 0, 0, 0);
    } else {
        if (WHO & 2)
            for (int it = 0; it < iters; it++)
#pragma unroll
                for (int j = 0; j < 32; j++) v[j & 7] = __builtin_fmaf(v[j & 7], fb, fa);
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int F16, int WHO> float run(float *d, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<F16, WHO>), dim3(256), dim3(512), 0, 0, d, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<F16, WHO>), dim3(256), dim3(512), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float *d; (void)hipMalloc(&d, 256 * 512 * 4);
    const int it = 20000;
    printf("f32 MFMA 16x16x4 : MFMA waves alone %.3f ms, VALU waves alone %.3f ms, together %.3f ms\n", run<0, 1>(d, it), run<0, 2>(d, it), run<0, 3>(d, it));
    printf("f16 MFMA 16x16x32: MFMA waves alone %.3f ms, VALU waves alone %.3f ms, together %.3f ms\n", run<1, 1>(d, it), run<1, 2>(d, it), run<1, 3>(d, it));
    return 0;
}

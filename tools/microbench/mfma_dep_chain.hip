// Does the hardware interlock dependent MFMAs whose SrcC is an EARLIER MFMA's vDst in a DIFFERENT register
// (D != C), issued back to back or one / two MFMAs apart, when several waves share a SIMD?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_dep tools/microbench/mfma_dep_chain.hip && /tmp/mfma_dep
// A = B = all ones (f16), so every v_mfma_f32_16x16x32_f16 adds exactly 32 to each accumulator element.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)1.0f; b[i] = (_Float16)1.0f; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f}, x0, x1, x2, y0 = {0.f, 0.f, 0.f, 0.f}, y1 = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0)   // back-to-back, D != C
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %5, %3\n\t"
                         "v_mfma_f32_16x16x32_f16 %1, %4, %5, %0\n\t"
                         "v_mfma_f32_16x16x32_f16 %2, %4, %5, %1\n\t"
                         "s_nop 15\n\ts_nop 15"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2) : "v"(c), "v"(a), "v"(b));
        if (MODE == 1)   // one independent MFMA between dependents
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %6, %7, %5\n\t"
                         "v_mfma_f32_16x16x32_f16 %3, %6, %7, %3\n\t"
                         "v_mfma_f32_16x16x32_f16 %1, %6, %7, %0\n\t"
                         "v_mfma_f32_16x16x32_f16 %4, %6, %7, %4\n\t"
                         "v_mfma_f32_16x16x32_f16 %2, %6, %7, %1\n\t"
                         "s_nop 15\n\ts_nop 15"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "+v"(y0), "+v"(y1) : "v"(c), "v"(a), "v"(b));
        if (MODE == 2)   // the result is then read by a VALU op after 16 + 16 wait states (reference: must be right)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %5, %3\n\t"
                         "s_nop 15\n\t"
                         "v_mfma_f32_16x16x32_f16 %1, %4, %5, %0\n\t"
                         "s_nop 15\n\t"
                         "v_mfma_f32_16x16x32_f16 %2, %4, %5, %1\n\t"
                         "s_nop 15\n\ts_nop 15"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2) : "v"(c), "v"(a), "v"(b));
        c = x2;
    }
    out[(blockIdx.x * 256 + threadIdx.x)] = c[0] + c[1] + c[2] + c[3] + y0[0] * 0.f + y1[0] * 0.f;
}

template <int MODE>
int run(const char *name, int blocks) {
    const int iters = 2000;
    float *d;
    hipMalloc(&d, blocks * 256 * sizeof(float));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    std::vector<float> h(blocks * 256);
    hipMemcpy(h.data(), d, h.size() * sizeof(float), hipMemcpyDeviceToHost);
    const float want = 4.0f * 96.0f * iters;
    long bad = 0;
    for (float v : h) bad += (v != want);
    printf("%-48s blocks %5d: %ld of %zu lanes wrong (want %.0f, e.g. got %.0f)\n", name, blocks, bad, h.size(), want, h[0]);
    hipFree(d);
    return bad != 0;
}

int main() {
    for (int blocks : {256, 1024, 4096}) {
        run<2>("dependent, 16 wait states between (reference)", blocks);
        run<0>("dependent D != C, back to back", blocks);
        run<1>("dependent D != C, one MFMA between", blocks);
    }
    return 0;
}

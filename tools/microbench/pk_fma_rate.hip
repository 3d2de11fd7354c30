// Issue cost of packed f32 VALU (v_pk_fma_f32) against plain v_fma_f32, v_exp_f32 and v_med3_f32 -- alone, with a partner wave
// on the same SIMD issuing f16 MFMAs, and interleaved with MFMAs in the same wave.  The fused MBConv kernels' GELU is built on
// v_pk_fma_f32; MI355X_MICROARCH.md prices it as an anti-lever beside MFMAs.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/pk_fma_rate.hip -o /tmp/pk_fma_rate && /tmp/pk_fma_rate
// Output: cycles per inner block (s_memtime over the whole loop / iterations) for the VALU wave and the MFMA wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// VMODE: 0 none, 1: 32 fmaf() (hipcc SLP-packs them into 16 v_pk_fma_f32!), 7: 32 v_fma_f32 by inline asm, 2 / 8: 16 v_pk_fma_f32 (the same 32 FMAs), 3: 16 v_exp_f32, 4: 32 v_med3_f32,
//        5: the GELU of kernels.hpp on 8 pairs (2 x gelu_erf_fast4-like: 12 pk + 2 med3 + 2 exp per pair, written with builtins)
// MMODE: 0 none, 1: partner waves (wave >= 4) issue 8 v_mfma_f32_16x16x32_f16 per block, 2: the SAME wave issues them interleaved
template <int VMODE, int MMODE>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    const bool valu_wave = MMODE == 1 ? wave < 4 : true, mfma_wave = MMODE == 1 ? wave >= 4 : MMODE == 2;
    f32x4 acc[8];
    f32x2 p[16];
    float v[32];
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { acc[i] = (f32x4){0, 0, 0, 0}; a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
    for (int i = 0; i < 32; i++) v[i] = threadIdx.x * 0.001f + i * 0.01f;
    for (int i = 0; i < 16; i++) p[i] = (f32x2){v[2 * i], v[2 * i + 1]};
    const float fa = 0.25f + threadIdx.x * 1e-6f, fb = 0.999f;
    const f32x2 pa = {fa, fa}, pb = {fb, fb};
    const f32x2 sc = {0.25f + iters * 1e-9f, 0.25f + iters * 1e-9f};   // wave-uniform: lives in an SGPR pair
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (mfma_wave && MMODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
        }
        if (valu_wave) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (MMODE == 2) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
                if (VMODE == 1) {
#pragma unroll
                    for (int r = 0; r < 4; r++) v[4 * j + r] = __builtin_fmaf(v[4 * j + r], fb, fa);
                } else if (VMODE == 2) {
#pragma unroll
                    for (int r = 0; r < 2; r++) p[2 * j + r] = __builtin_elementwise_fma(p[2 * j + r], pb, pa);
                } else if (VMODE == 3) {
#pragma unroll
                    for (int r = 0; r < 2; r++) v[2 * j + r] = __builtin_amdgcn_exp2f(v[2 * j + r]);
                } else if (VMODE == 4) {
#pragma unroll
                    for (int r = 0; r < 4; r++) v[4 * j + r] = __builtin_amdgcn_fmed3f(v[4 * j + r], fa, fb);
                } else if (VMODE == 5) {   // one GELU pair per j: 2 med3, 10 packed, 2 exp
                    f32x2 x = p[j], m, q;
                    m[0] = __builtin_amdgcn_fmed3f(x[0], 0.0f, 3.0e38f); m[1] = __builtin_amdgcn_fmed3f(x[1], 0.0f, 3.0e38f);
                    const f32x2 aa = __builtin_elementwise_fma(m, (f32x2){2.0f, 2.0f}, -x);
                    q = __builtin_elementwise_fma(aa, pb, pa);
#pragma unroll
                    for (int r = 0; r < 7; r++) q = __builtin_elementwise_fma(q, aa, pa);
                    f32x2 e;
                    e[0] = __builtin_amdgcn_exp2f(q[0]); e[1] = __builtin_amdgcn_exp2f(q[1]);
                    p[j] = __builtin_elementwise_fma(-aa, e, m);
                } else if (VMODE == 7) {   // 4 v_fma_f32 that hipcc's SLP vectoriser cannot pack (inline asm)
#pragma unroll
                    for (int r = 0; r < 4; r++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[4 * j + r]) : "v"(fb), "v"(fa));
                } else if (VMODE == 8) {   // 2 v_pk_fma_f32 (inline asm): the same 4 FMAs
#pragma unroll
                    for (int r = 0; r < 2; r++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[2 * j + r]) : "v"(pb), "v"(pa));
                } else if (VMODE == 9) {   // 2 v_pk_fma_f32 whose addend is an SGPR pair broadcast by op_sel_hi (what hipcc emits for a constant)
#pragma unroll
                    for (int r = 0; r < 2; r++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(p[2 * j + r]) : "v"(pb), "s"(sc));
                } else if (VMODE == 10) {  // 2 v_pk_fma_f32, addend = the low half of a VGPR pair broadcast by op_sel_hi
#pragma unroll
                    for (int r = 0; r < 2; r++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]" : "+v"(p[2 * j + r]) : "v"(pb), "v"(pa));
                } else if (VMODE == 11) {  // 2 v_pk_fma_f32 with an inline constant addend
#pragma unroll
                    for (int r = 0; r < 2; r++) asm volatile("v_pk_fma_f32 %0, %0, %1, -1.0 op_sel_hi:[1,1,0]" : "+v"(p[2 * j + r]) : "v"(pb));
                } else if (VMODE == 12) {  // 2 v_pk_fma_f32, addend a full SGPR pair (no op_sel)
#pragma unroll
                    for (int r = 0; r < 2; r++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[2 * j + r]) : "v"(pb), "s"(sc));
                } else if (VMODE == 6) {   // the same GELU pair with scalar f32 ops only: 2 med3, 20 fma, 2 exp
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const float x = v[2 * j + h];
                        const float m = __builtin_amdgcn_fmed3f(x, 0.0f, 3.0e38f);
                        const float aa = __builtin_fmaf(m, 2.0f, -x);
                        float q = __builtin_fmaf(aa, fb, fa);
#pragma unroll
                        for (int r = 0; r < 7; r++) q = __builtin_fmaf(q, aa, fa);
                        v[2 * j + h] = __builtin_fmaf(-aa, __builtin_amdgcn_exp2f(q), m);
                    }
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 32; i++) s += v[i];
    for (int i = 0; i < 16; i++) s += p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int VMODE, int MMODE>
void run(const char *name, float *d, unsigned long long *dc, int iters) {
    const int threads = MMODE == 1 ? 512 : 256;
    hipLaunchKernelGGL((k<VMODE, MMODE>), dim3(256), dim3(threads), 0, 0, d, dc, iters);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 1e30f;
    for (int rep = 0; rep < 5; rep++) {   // the fastest of five launches (the clock ramps during the first ones)
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<VMODE, MMODE>), dim3(256), dim3(threads), 0, 0, d, dc, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float t; (void)hipEventElapsedTime(&t, e0, e1);
        ms = t < ms ? t : ms;
    }
    std::vector<unsigned long long> h(256 * 8);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    double cv = 0, cm = 0;
    for (int b = 0; b < 256; b++) { cv += (double)h[b * 8 + 0]; cm += (double)h[b * 8 + (MMODE == 1 ? 4 : 0)]; }
    // s_memtime ticks at 100 MHz on this part; wall time * nominal clock is the comparable number
    printf("%-58s wall %8.3f ms  %7.1f ns per block of 8\n", name, ms, ms * 1e6 / iters);
    (void)cv; (void)cm;
}

int main() {
    float *d; unsigned long long *dc;
    (void)hipMalloc(&d, 256 * 512 * sizeof(float));
    (void)hipMalloc(&dc, 256 * 8 * 8);
    const int it = 100000;
    printf("one VALU wave per SIMD (+ one MFMA partner wave per SIMD where stated); a block = 8 groups\n");
    run<1, 0>("32 v_fma_f32", d, dc, it);
    run<2, 0>("16 v_pk_fma_f32 (same 32 FMAs)", d, dc, it);
    run<7, 0>("32 v_fma_f32 (inline asm: not SLP-packed)", d, dc, it);
    run<8, 0>("16 v_pk_fma_f32 (inline asm)", d, dc, it);
    run<7, 1>("32 v_fma_f32 (asm) + partner 8 MFMA", d, dc, it);
    run<8, 1>("16 v_pk_fma_f32 (asm) + partner 8 MFMA", d, dc, it);
    run<7, 2>("same wave: 8 MFMA interleaved with 32 v_fma_f32 (asm)", d, dc, it);
    run<8, 2>("same wave: 8 MFMA interleaved with 16 v_pk_fma_f32 (asm)", d, dc, it);
    run<9, 0>("16 v_pk_fma_f32 (asm), addend SGPR pair, op_sel_hi:[1,1,0]", d, dc, it);
    run<12, 0>("16 v_pk_fma_f32 (asm), addend SGPR pair, no op_sel", d, dc, it);
    run<10, 0>("16 v_pk_fma_f32 (asm), addend VGPR pair, op_sel_hi:[1,1,0]", d, dc, it);
    run<11, 0>("16 v_pk_fma_f32 (asm), addend inline constant", d, dc, it);
    run<3, 0>("16 v_exp_f32", d, dc, it);
    run<4, 0>("32 v_med3_f32", d, dc, it);
    run<5, 0>("8 GELU pairs, packed (2 med3 + 10 pk + 2 exp each)", d, dc, it);
    run<6, 0>("8 GELU pairs, scalar (2 med3 + 20 fma + 2 exp each)", d, dc, it);
    run<0, 1>("partner: 8 v_mfma_f32_16x16x32_f16 alone", d, dc, it);
    run<1, 1>("32 v_fma_f32 + partner 8 MFMA", d, dc, it);
    run<2, 1>("16 v_pk_fma_f32 + partner 8 MFMA", d, dc, it);
    run<5, 1>("8 GELU pairs packed + partner 8 MFMA", d, dc, it);
    run<6, 1>("8 GELU pairs scalar + partner 8 MFMA", d, dc, it);
    run<0, 2>("same wave: 8 MFMA alone", d, dc, it);
    run<1, 2>("same wave: 8 MFMA interleaved with 32 v_fma_f32", d, dc, it);
    run<2, 2>("same wave: 8 MFMA interleaved with 16 v_pk_fma_f32", d, dc, it);
    run<5, 2>("same wave: 8 MFMA interleaved with 8 GELU pairs packed", d, dc, it);
    run<6, 2>("same wave: 8 MFMA interleaved with 8 GELU pairs scalar", d, dc, it);
    return 0;
}

// `v_pk_add_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]` (b's halves swapped) fed straight from ds_read2_b32 pairs:
// is it always right?  (the split-f16 mel kernel was nondeterministic with this form -- hipcc's
// SLP vectoriser produces it from `x[a + j] + x[b - j]` -- and deterministic with single v_add_f32s.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_add tools/microbench/pk_add_opsel.hip && /tmp/pk_add
// Every mode computes y0 = u0 + w1, y1 = u1 + w0 from the same LDS contents; the scalar form is the reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// MODE 0: two v_add_f32      1: v_pk_add_f32 op_sel right after the s_waitcnt     2: the same after s_nop 0
//      3: after s_nop 3      4: mode 1 without MFMA traffic                        5: v_pk_add_f32 on a register copy (v_mov first)
//      6: v_pk_add_f32 WITHOUT op_sel on pre-swapped operands (same sums)          7: mode 1, MFMAs only in the OTHER waves (odd waves)
//      8: mode 1 with f32 MFMAs (16x16x4) as the traffic                            9: v_pk_fma_f32 y = u * 1 + w, op_sel as mode 1
//     10: the broadcast forms the GELU uses (op_sel_hi only): v_pk_mul_f32 op_sel_hi:[1,0], v_pk_fma_f32 op_sel_hi:[1,0,1] neg;
//         its reference is mode 11 = the same instructions with no MFMA in flight
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float *out, int iters) {
    extern __shared__ float xs[];   // ~60 KB so that two workgroups share a CU, like the mel kernel
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 15000; i += 256) xs[i] = (float)((i * 7) % 1021) * 0.25f;
    __syncthreads();
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)1.0f; b[i] = (_Float16)(0.001f * lane); }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float sum = 0.f;
    const int li = lane & 15, kq = lane >> 4;
    for (int it = 0; it < iters; it++) {
        const int j0 = ((it * 32) % 1024) + 8 * kq;
        const float *f = xs + li * 278 + j0 + 1, *r = xs + li * 278 + 2047 - j0;
#pragma unroll
        for (int jj = 0; jj < 8; jj += 2) {
            f32x2 u, w, y;
            const unsigned fa = (unsigned)(size_t)(f + jj), ra = (unsigned)(size_t)(r - jj - 1);
            asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %3 offset1:1\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(u), "=&v"(w) : "v"(fa), "v"(ra) : "memory");
            if (MODE == 0)
                asm volatile("v_add_f32_e32 %0, %2, %5\n\tv_add_f32_e32 %1, %3, %4" : "=&v"(y[0]), "=&v"(y[1]) : "v"(u[0]), "v"(u[1]), "v"(w[0]), "v"(w[1]));
            if (MODE == 1 || MODE == 4 || MODE == 7 || MODE == 8)
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(y) : "v"(u), "v"(w));
            if (MODE == 6) {
                f32x2 ws;
                asm volatile("v_mov_b32_e32 %0, %1" : "=v"(ws[0]) : "v"(w[1]));
                asm volatile("v_mov_b32_e32 %0, %1" : "=v"(ws[1]) : "v"(w[0]));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y) : "v"(u), "v"(ws));
            }
            if (MODE == 9) {
                const f32x2 one = {1.0f, 1.0f};
                asm volatile("v_pk_fma_f32 %0, %1, %3, %2 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(y) : "v"(u), "v"(w), "v"(one));
            }
            if (MODE == 2)
                asm volatile("s_nop 0\n\tv_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(y) : "v"(u), "v"(w));
            if (MODE == 3)
                asm volatile("s_nop 3\n\tv_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(y) : "v"(u), "v"(w));
            if (MODE == 5) {
                f32x2 u2, w2;
                asm volatile("v_mov_b32_e32 %0, %1" : "=v"(u2[0]) : "v"(u[0]));
                asm volatile("v_mov_b32_e32 %0, %1" : "=v"(u2[1]) : "v"(u[1]));
                asm volatile("v_mov_b32_e32 %0, %1" : "=v"(w2[0]) : "v"(w[0]));
                asm volatile("v_mov_b32_e32 %0, %1" : "=v"(w2[1]) : "v"(w[1]));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(y) : "v"(u2), "v"(w2));
            }
            if (MODE == 10 || MODE == 11) {   // the GELU's forms: t = u * c (c broadcast from the low half), y = t * 2 - w
                const f32x2 c = {0.75f, 123.0f};
                f32x2 t;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(u), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(y) : "v"(t), "v"(w));
            }
            sum += y[0] * 0.5f + y[1] * 0.25f;
        }
        if (MODE == 8) {
#pragma unroll
            for (int q = 0; q < 6; q++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, 0.001f * lane, acc, 0, 0, 0);
        } else if (MODE != 4 && MODE != 11 && (MODE != 7 || ((tid >> 6) & 1))) {   // MFMA traffic (as the mel kernel's main loop has between the sums)
#pragma unroll
            for (int q = 0; q < 6; q++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + tid] = sum + 1e-30f * acc[0];
}

template <int MODE>
static std::vector<float> run(int blocks, int iters) {
    float *d;
    hipMalloc(&d, blocks * 256 * sizeof(float));
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 60000, 0, d, iters);
    std::vector<float> h(blocks * 256);
    hipMemcpy(h.data(), d, h.size() * sizeof(float), hipMemcpyDeviceToHost);
    hipFree(d);
    return h;
}

template <int MODE>
static void report(const char *name, const std::vector<float> &ref, int blocks, int iters) {
    long worst = 0, total = 0;
    for (int rep = 0; rep < 4; rep++) {
        const std::vector<float> s = run<MODE>(blocks, iters);
        long bad = 0;
        for (size_t i = 0; i < ref.size(); i++) bad += s[i] != ref[i];
        worst = bad > worst ? bad : worst;
        total += bad;
    }
    printf("%-58s wrong lanes per run: mean %8.0f  max %8ld  (of %zu)\n", name, total / 4.0, worst, ref.size());
}

int main() {
    const int blocks = 4096, iters = 400;
    const std::vector<float> ref = run<0>(blocks, iters);
    report<0>("two v_add_f32 (reference, re-run)", ref, blocks, iters);
    report<1>("v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", ref, blocks, iters);
    report<2>("  ... after s_nop 0", ref, blocks, iters);
    report<3>("  ... after s_nop 3", ref, blocks, iters);
    report<4>("  ... without MFMA traffic", ref, blocks, iters);
    report<5>("  ... on operands copied by v_mov_b32 first", ref, blocks, iters);
    report<7>("  ... MFMAs only in the odd waves (even waves checked too)", ref, blocks, iters);
    report<8>("  ... with f32 MFMAs (16x16x4) as the traffic", ref, blocks, iters);
    report<6>("v_pk_add_f32 without op_sel (operands pre-swapped)", ref, blocks, iters);
    report<9>("v_pk_fma_f32 u * 1 + w, op_sel:[0,0,1] op_sel_hi:[1,1,0]", ref, blocks, iters);
    const std::vector<float> ref2 = run<11>(blocks, iters);   // the same packed instructions with no MFMA in flight
    report<10>("GELU's forms (v_pk_mul op_sel_hi:[1,0]; v_pk_fma op_sel_hi:[1,0,1] neg) vs no-MFMA run", ref2, blocks, iters);
    return 0;
}

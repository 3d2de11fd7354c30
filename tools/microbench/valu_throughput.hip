// THROUGHPUT price list of the vector instructions the fused blocks' activation is built from (VERDICT r2 weak #3a):
// pk_fma_rate.hip measured ONE wave per SIMD, where a dependent chain's latency can hide as "issue cost".  Here 1, 2, 3 and 4
// waves share every SIMD (the occupancy the early fused blocks run at) and every wave carries 16 independent chains, so what is
// printed is the SIMD's sustained issue rate: nanoseconds per instruction and SIMD, and the same in units of v_fma_f32 (the
// full-rate reference: 64 lanes over 16 ALUs = 4 cycles).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/valu_throughput.hip -o /tmp/valu_throughput && /tmp/valu_throughput
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: 32 v_fma_f32; 1: 16 v_pk_fma_f32 (VGPR operands); 2: 16 v_pk_fma_f32, addend an SGPR pair; 3: 16 v_exp_f32; 4: 32 v_med3_f32;
//      5: 8 GELU pairs of round 2 (degree 8: 2 med3 + 10 pk + 2 exp each; + the pre-scale multiply = 11 pk);
//      6: 8 GELU pairs of round 3 (degree 5, scaled coefficients from SGPRs: 2 med3 + 7 pk + 2 exp each); 7: 16 v_pk_mul_f32;
//      10: 8 GELU pairs of round 4 (2 x (4 fma + mul + exp + add + fma), scalar, |v| as source modifier);
//      8: 8 swish pairs (mul, exp, add, rcp, mul per value: scalar ops); 9: 16 v_rcp_f32
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float s0, float s1) {
    f32x2 p[16];
    float v[32];
    for (int i = 0; i < 32; i++) v[i] = threadIdx.x * 0.001f + i * 0.01f;
    for (int i = 0; i < 16; i++) p[i] = (f32x2){v[2 * i], v[2 * i + 1]};
    const float fa = 0.25f + threadIdx.x * 1e-6f, fb = 0.999f;
    const f32x2 pa = {fa, fa}, pb = {fb, fb};
    const f32x2 sc = {s0, s0}, sd = {s1, s1};    // wave-uniform (kernel arguments): SGPR pairs
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[4 * j + r]) : "v"(fb), "v"(fa));
            } else if (MODE == 1) {
#pragma unroll
                for (int r = 0; r < 2; r++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[2 * j + r]) : "v"(pb), "v"(pa));
            } else if (MODE == 2) {
#pragma unroll
                for (int r = 0; r < 2; r++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[2 * j + r]) : "v"(pb), "s"(sc));
            } else if (MODE == 3) {
#pragma unroll
                for (int r = 0; r < 2; r++) asm volatile("v_exp_f32 %0, %0" : "+v"(v[2 * j + r]));
            } else if (MODE == 4) {
#pragma unroll
                for (int r = 0; r < 4; r++) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[4 * j + r]) : "v"(fa), "v"(fb));
            } else if (MODE == 7) {
#pragma unroll
                for (int r = 0; r < 2; r++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[2 * j + r]) : "v"(pb));
            } else if (MODE == 9) {
#pragma unroll
                for (int r = 0; r < 2; r++) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[2 * j + r]));
            } else if (MODE == 5) {
                f32x2 x = p[j] * pb, m, q;
                m[0] = __builtin_amdgcn_fmed3f(x[0], 0.0f, 3.0e38f); m[1] = __builtin_amdgcn_fmed3f(x[1], 0.0f, 3.0e38f);
                const f32x2 aa = __builtin_elementwise_fma(m, (f32x2){2.0f, 2.0f}, -x);
                q = __builtin_elementwise_fma(aa, pb, pa);
#pragma unroll
                for (int r = 0; r < 7; r++) q = __builtin_elementwise_fma(q, aa, pa);
                f32x2 e;
                e[0] = __builtin_amdgcn_exp2f(q[0]); e[1] = __builtin_amdgcn_exp2f(q[1]);
                p[j] = __builtin_elementwise_fma(-aa, e, m);
            } else if (MODE == 6) {
                f32x2 x = p[j], m, q;
                m[0] = __builtin_amdgcn_fmed3f(x[0], 0.0f, 3.0e38f); m[1] = __builtin_amdgcn_fmed3f(x[1], 0.0f, 3.0e38f);
                const f32x2 aa = __builtin_elementwise_fma(m, (f32x2){2.0f, 2.0f}, -x);
                q = __builtin_elementwise_fma(aa, sc, sd);
#pragma unroll
                for (int r = 0; r < 4; r++) q = __builtin_elementwise_fma(q, aa, r & 1 ? sc : sd);
                f32x2 e;
                e[0] = __builtin_amdgcn_exp2f(q[0]); e[1] = __builtin_amdgcn_exp2f(q[1]);
                p[j] = __builtin_elementwise_fma(-aa, e, m);
            } else if (MODE == 10) {
                // round 4: 2 GELU(v) = (v + |v|) - |v| exp2(|v| (c1 + ... )), scalar FMAs with |v| as a source modifier
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const float x = v[2 * j + h], a = __builtin_fabsf(x);
                    float t = __builtin_fmaf(a, s0, s1);
                    t = __builtin_fmaf(t, a, s0); t = __builtin_fmaf(t, a, s1); t = __builtin_fmaf(t, a, s0);
                    const float e2 = __builtin_amdgcn_exp2f(t * a);
                    v[2 * j + h] = __builtin_fmaf(-a, e2, x + a);
                }
            } else if (MODE == 11) {
                // the same 2 GELU with PACKED FMAs: |v| by v_and_b32 (no abs modifier on packed ops), v + |v| and the final fma packed
                const f32x2 x = p[j];
                f32x2 aa;
                aa[0] = __builtin_fabsf(x[0]); aa[1] = __builtin_fabsf(x[1]);
                asm volatile("" : "+v"(aa));   // (materialised: what the packed form costs)
                f32x2 q = __builtin_elementwise_fma(aa, sc, sd);
#pragma unroll
                for (int r = 0; r < 3; r++) q = __builtin_elementwise_fma(q, aa, r & 1 ? sc : sd);
                q = q * aa;
                f32x2 e;
                e[0] = __builtin_amdgcn_exp2f(q[0]); e[1] = __builtin_amdgcn_exp2f(q[1]);
                p[j] = __builtin_elementwise_fma(-aa, e, x + aa);
            } else if (MODE == 8) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const float x = v[2 * j + h];
                    v[2 * j + h] = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 32; i++) s += v[i];
    for (int i = 0; i < 16; i++) s += p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double g_fma_ns[5];

template <int MODE>
void run(const char *name, int n_instr, float *d, int iters) {
    printf("%-66s", name);
    for (int w = 1; w <= 4; w++) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float ms = 1e30f;
        for (int rep = 0; rep < 6; rep++) {   // the fastest of six launches (the clock ramps during the first ones)
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256 * w), 0, 0, d, iters, 0.25f, 0.125f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float t; (void)hipEventElapsedTime(&t, e0, e1);
            if (rep > 0) ms = t < ms ? t : ms;
        }
        // one block per CU, w waves per SIMD: the SIMD issued w * iters * n_instr instructions of this kind
        const double ns = ms * 1e6 / ((double)iters * n_instr * w);
        if (MODE == 0) g_fma_ns[w] = ns;
        printf("  w=%d %6.3f ns (%5.2f fma)", w, ns, 4.0 * ns / g_fma_ns[w]);
    }
    printf("\n");
}

int main() {
    float *d;
    (void)hipMalloc(&d, 256 * 1024 * sizeof(float));
    const int it = 50000;
    printf("per instruction and SIMD, 1..4 waves per SIMD; (x fma) = cycles if v_fma_f32 issues in 4\n");
    run<0>("v_fma_f32", 32, d, it);
    run<1>("v_pk_fma_f32 (VGPR operands)", 16, d, it);
    run<2>("v_pk_fma_f32 (addend: SGPR pair)", 16, d, it);
    run<7>("v_pk_mul_f32", 16, d, it);
    run<4>("v_med3_f32", 32, d, it);
    run<3>("v_exp_f32", 16, d, it);
    run<9>("v_rcp_f32", 16, d, it);
    run<5>("GELU pair, round 2 (2 med3 + 11 pk + 2 exp), per PAIR", 8, d, it);
    run<6>("GELU pair, round 3 (2 med3 + 7 pk [SGPR coefficients] + 2 exp), per PAIR", 8, d, it);
    run<10>("GELU pair, round 4 (2 x (4 fma + mul + exp + add + fma), |v| modifiers), per PAIR", 8, d, it);
    run<11>("GELU pair, round 4 packed (2 and + 4 pk fma + pk mul + 2 exp + pk add + pk fma), per PAIR", 8, d, it);
    run<8>("swish pair (2 x (mul, exp, add, rcp, mul)), per PAIR", 8, d, it);
    return 0;
}

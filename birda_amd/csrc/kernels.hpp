// Internal launcher interface between the host executor (api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cmath>
#include <mutex>

// Environment switches.  The PRODUCT library reads a short, documented list (tests/test_abi_and_host.py holds the shipped .so to
// it): BIRDA_HIP_PRECISION, BIRDA_HIP_COPY_THREADS, BIRDA_HOST_PIPELINE_DEPTH, BIRDA_INFERENCE_TIMEOUT, BIRDA_HIP_ROCTX,
// BIRDA_HOST_TIMING, and the switches the parity tests drive it with (BIRDA_HIP_KEEP_TENSORS, _KEEP_FUSED, _MB_CFG, _MB_PREFER,
// _FUSE, _MEL32, _MEL_F32, _HEAD_GAP).  Everything else -- ablation bits, A/B aids of experiments DESIGN.md reports -- goes through
// BH_XENV and exists only in the EXPERIMENTS build (`make EXPERIMENTS=1`): in the product the name is not even in the binary.
#ifdef BIRDA_HIP_EXPERIMENTS
#define BH_XENV(name) getenv(name)
#else
#define BH_XENV(name) (static_cast<const char *>(nullptr))
#endif

namespace bh {

// One-time launcher set-up (kernel attributes, CU count) is per DEVICE, not per process: one process may hold
// classifiers on several GPUs (bh_multi_*, one host thread per device).
constexpr int MAX_DEVICES = 64;
inline int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEVICES) d = 0;
    return d;
}
struct DeviceOnce {
    std::mutex mu;
    std::atomic<unsigned char> done[MAX_DEVICES] = {};
    template <class F> void run(F &&f) {
        const int d = current_device();
        if (done[d].load(std::memory_order_acquire)) return;
        std::lock_guard<std::mutex> g(mu);
        if (!done[d].load(std::memory_order_relaxed)) { f(); done[d].store(1, std::memory_order_release); }
    }
};
// Power-of-two pre-scale of an f16 operand tensor (host side): the exponent s with max |w| * 2^s in [2^13, 2^14) -- a factor
// of four below the f16 maximum 65 504, and high enough that the lo half (w - f16(w), ~2^-11 of w) of every entry within
// 2^-13 of the largest is a NORMAL f16.  Exact (a power of two) and undone exactly in the consumer's f32 epilogue.
// 0 for an all-zero tensor; clamped so that biases multiplied alike stay far inside the f32 range.
inline int f16_scale_exponent(float max_abs) {
    if (!(max_abs > 0.0f) || !std::isfinite(max_abs)) return 0;
    int e = 0;
    (void)std::frexp(max_abs, &e);   // max_abs = f * 2^e, f in [0.5, 1)
    const int s = 14 - e;
    return s < -60 ? -60 : (s > 60 ? 60 : s);
}
inline int device_cu_count() {
    static std::atomic<int> n_cu[MAX_DEVICES] = {};
    const int d = current_device();
    int n = n_cu[d].load(std::memory_order_relaxed);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
    n_cu[d].store(n, std::memory_order_relaxed);
    return n;
}

enum Act : int { ACT_NONE = 0, ACT_RELU, ACT_RELU6, ACT_SWISH, ACT_GELU_ERF, ACT_GELU_TANH, ACT_SIGMOID };

constexpr int MAX_BRANCHES = 4;

// erf GELU, GELU(v) = v Phi(v), without erff and without a division:
//     GELU(v) = max(v, 0) - |v| Phi(-|v|),   Phi(-a) = exp2(Q(a)),
// Q(a) = -1 + a (c1 + ... + c5 a^4): a weighted minimax fit of log2 Phi(-a) on [0, 6.2] (tools/fit_gelu.py), the weight being
// the stated tolerance of the activation, |GELU error| <= 5e-7 max(|v|, 1) (VERDICT r2 next #3b): measured 8.9e-7 at |v| = 3.6,
// 0.53 of the tolerance at the worst point, f32 evaluation included.  Q(0) = -1 exactly, so GELU(0) = 0; the leading coefficient
// is negative, so Q falls monotonically to -inf beyond the fit range, where Phi(-a) < 3e-10 rounds away.  Per value: 1 v_med3,
// 3.5 packed FMAs and ONE transcendental (v_exp_f32, quarter rate).  Round 2 carried a degree-8 fit (1.7e-7, three more packed
// FMAs per pair: 102 cycles per pair of values against ~80 now, tools/microbench/pk_fma_rate.hip); the logits sat 10x inside
// their tolerance with it.  -DBH_GELU_DEGREE=8 restores it (A/B aid).
// Exp-free forms were priced and rejected: a polynomial for a Phi(-a) itself needs degree >= 15 on [0, 5.3] for the same
// tolerance (Chebyshev interpolation: 4.4e-7 at degree 15, 1.0e-6 at 14) -- seven more packed FMAs per pair (39 cycles) plus
// two clamps (11) to remove two v_exp_f32 (30).
#ifndef BH_GELU_DEGREE
#define BH_GELU_DEGREE 5
#endif
#if BH_GELU_DEGREE == 8
#define BH_GELU_C1 -1.1511051654815674f
#define BH_GELU_C2 -0.4592081904411316f
#define BH_GELU_C3 -0.052496183663606644f
#define BH_GELU_C4 0.007063428405672312f
#define BH_GELU_C5 -0.00013694142398890108f
#define BH_GELU_C6 -0.00018617883324623108f
#define BH_GELU_C7 3.93775844713673e-05f
#define BH_GELU_C8 -2.834923634509323e-06f
#define BH_GELU_CTOP BH_GELU_C8
#define BH_GELU_CNEXT BH_GELU_C7
#define BH_GELU_MID(STEP) STEP(BH_GELU_C6) STEP(BH_GELU_C5) STEP(BH_GELU_C4) STEP(BH_GELU_C3) STEP(BH_GELU_C2) STEP(BH_GELU_C1)
#else
#define BH_GELU_C1 -1.1510556936264038f
#define BH_GELU_C2 -0.45938199758529663f
#define BH_GELU_C3 -0.05241312459111214f
#define BH_GELU_C4 0.007328995503485203f
#define BH_GELU_C5 -0.0005096292006783187f
#define BH_GELU_CTOP BH_GELU_C5
#define BH_GELU_CNEXT BH_GELU_C4
#define BH_GELU_MID(STEP) STEP(BH_GELU_C3) STEP(BH_GELU_C2) STEP(BH_GELU_C1)
#endif
// max(v, 0) as ONE instruction (v_med3_f32 v, 0, 3e38: the finite bound keeps hipcc from rewriting it as a canonicalise + max pair): fmaxf() in IEEE mode first canonicalises its operand
// (a second v_max).  NOT inline asm: the operand is usually an MFMA result, and hipcc places the wait
// states an MFMA -> VALU read needs only in front of instructions it knows -- an inline-asm v_max_f32
// here read stale accumulators whenever fewer than ~10 instructions separated it from the last MFMA
// (single-tile groups of the expand GEMM).
__device__ __forceinline__ float bh_relu1(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 3.0e38f); }
__device__ __forceinline__ float gelu_erf_fast(float v) {
    const float m = bh_relu1(v);
    const float a = __builtin_fmaf(m, 2.0f, -v);   // |v|, exactly
    float q = __builtin_fmaf(a, BH_GELU_CTOP, BH_GELU_CNEXT);
#define BH_GELU_STEP1(c) q = __builtin_fmaf(q, a, c);
    BH_GELU_MID(BH_GELU_STEP1)
#undef BH_GELU_STEP1
    q = __builtin_fmaf(q, a, -1.0f);
    const float e = __builtin_amdgcn_exp2f(q);
    return __builtin_fmaf(-a, e, m);   // 0.5 v + |v| (0.5 - e) with 0.5 v + 0.5 |v| = max(v, 0)
}

// |v| = 2 max(v, 0) - v for a pair, one packed FMA (written out: hipcc otherwise negates v with v_xor first)
typedef float bh_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bh_f32x2 bh_abs_from_relu2(bh_f32x2 m, bh_f32x2 v) {
    bh_f32x2 a;
    asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(a) : "v"(m), "v"(v));
    return a;
}

// Four GELUs at once with the f32 packed ops (v_pk_fma_f32): 7 packed instructions + 2 v_med3_f32 +
// 2 v_exp_f32 per pair; same arithmetic, bit-identical to gelu_erf_fast.  The two pairs' Horner chains
// are interleaved by hand: back-to-back dependent packed ops cost a wait state each on gfx950.
#define BH_PK(c) ((bh_f32x2){(c), (c)})
__device__ __forceinline__ void gelu_erf_fast4(bh_f32x2 &v0, bh_f32x2 &v1) {
    bh_f32x2 m0, m1;
    m0[0] = bh_relu1(v0[0]); m0[1] = bh_relu1(v0[1]);
    m1[0] = bh_relu1(v1[0]); m1[1] = bh_relu1(v1[1]);
    const bh_f32x2 a0 = bh_abs_from_relu2(m0, v0);
    const bh_f32x2 a1 = bh_abs_from_relu2(m1, v1);
    bh_f32x2 q0 = __builtin_elementwise_fma(a0, BH_PK(BH_GELU_CTOP), BH_PK(BH_GELU_CNEXT));
    bh_f32x2 q1 = __builtin_elementwise_fma(a1, BH_PK(BH_GELU_CTOP), BH_PK(BH_GELU_CNEXT));
#define BH_GELU_STEP(c)                                       \
    q0 = __builtin_elementwise_fma(q0, a0, BH_PK(c));         \
    q1 = __builtin_elementwise_fma(q1, a1, BH_PK(c));
    BH_GELU_MID(BH_GELU_STEP) BH_GELU_STEP(-1.0f)
#undef BH_GELU_STEP
    bh_f32x2 e0, e1;
    e0[0] = __builtin_amdgcn_exp2f(q0[0]); e0[1] = __builtin_amdgcn_exp2f(q0[1]);
    e1[0] = __builtin_amdgcn_exp2f(q1[0]); e1[1] = __builtin_amdgcn_exp2f(q1[1]);
    v0 = __builtin_elementwise_fma(-a0, e0, m0);
    v1 = __builtin_elementwise_fma(-a1, e1, m1);
}
__device__ __forceinline__ bh_f32x2 gelu_erf_fast2(bh_f32x2 v) {
    bh_f32x2 m;
    m[0] = bh_relu1(v[0]); m[1] = bh_relu1(v[1]);
    const bh_f32x2 a = bh_abs_from_relu2(m, v);
    bh_f32x2 q = __builtin_elementwise_fma(a, BH_PK(BH_GELU_CTOP), BH_PK(BH_GELU_CNEXT));
#define BH_GELU_STEP2(c) q = __builtin_elementwise_fma(q, a, BH_PK(c));
    BH_GELU_MID(BH_GELU_STEP2) BH_GELU_STEP2(-1.0f)
#undef BH_GELU_STEP2
    bh_f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(q[0]); e[1] = __builtin_amdgcn_exp2f(q[1]);
    return __builtin_elementwise_fma(-a, e, m);
}

// The same four GELUs on SCALED arguments: the split-f16 expand GEMM of a fused block leaves acc = 2^s x (its weight planes and
// bias carry 2^s, f16_scale_exponent).  GELU is not homogeneous, but its pieces are: max(acc, 0) = 2^s max(x, 0),
// 2 max(acc, 0) - acc = 2^s |x|, and Q(|x|) = -1 + sum c_k |x|^k = -1 + sum (c_k 2^-ks) (2^s |x|)^k -- the Horner chain runs on the
// scaled magnitude with coefficients c_k 2^-ks (exact: powers of two; every intermediate is the unscaled one times a power of
// two, so q is bit-identical), and max - magnitude * e comes out as 2^s GELU(x).  The factor 2^s then rides through the LDS grid
// into the depthwise taps, which the host multiplies by 2^-s (api.hip plan_fusion).  Saves the packed multiply per pair that
// round 2 spent in front of every expand GELU.  gc[k - 1] = c_k 2^-ks, wave-uniform (SGPRs).
struct GeluScaled { float c1, c2, c3, c4, c5; };
__device__ __forceinline__ void gelu_erf_fast4_scaled(bh_f32x2 &v0, bh_f32x2 &v1, const GeluScaled &gc) {
    bh_f32x2 m0, m1;
    m0[0] = bh_relu1(v0[0]); m0[1] = bh_relu1(v0[1]);
    m1[0] = bh_relu1(v1[0]); m1[1] = bh_relu1(v1[1]);
    const bh_f32x2 a0 = bh_abs_from_relu2(m0, v0);
    const bh_f32x2 a1 = bh_abs_from_relu2(m1, v1);
    bh_f32x2 q0 = __builtin_elementwise_fma(a0, BH_PK(gc.c5), BH_PK(gc.c4));
    bh_f32x2 q1 = __builtin_elementwise_fma(a1, BH_PK(gc.c5), BH_PK(gc.c4));
#define BH_GELU_STEP(c)                                       \
    q0 = __builtin_elementwise_fma(q0, a0, BH_PK(c));         \
    q1 = __builtin_elementwise_fma(q1, a1, BH_PK(c));
    BH_GELU_STEP(gc.c3) BH_GELU_STEP(gc.c2) BH_GELU_STEP(gc.c1) BH_GELU_STEP(-1.0f)
#undef BH_GELU_STEP
    bh_f32x2 e0, e1;
    e0[0] = __builtin_amdgcn_exp2f(q0[0]); e0[1] = __builtin_amdgcn_exp2f(q0[1]);
    e1[0] = __builtin_amdgcn_exp2f(q1[0]); e1[1] = __builtin_amdgcn_exp2f(q1[1]);
    v0 = __builtin_elementwise_fma(-a0, e0, m0);
    v1 = __builtin_elementwise_fma(-a1, e1, m1);
}
// -DBH_GELU_2X=1 builds the fused blocks with the "twice the GELU" form below instead of round 3's gelu_erf_fast4 (max(v, 0) form,
// no factor two to fold).  Built, parity-green and MEASURED in round 4 -- and off: in isolation the packed 2x form is 12 % cheaper
// per pair (22.5 against 25.6 ns, profiles/r4_d_valu_throughput.txt), inside the kernels it is 0.7 % SLOWER on every one of five
// alternations on one box (fused blocks 5.861 -> 5.902 us per segment, profiles/r4_e_gelu2x_packed_ab.txt: tools/ab.sh lib with
// the two builds of one tree); a scalar spelling with |v| as a free source modifier measured 31.2 ns per pair.
#ifndef BH_GELU_2X
#define BH_GELU_2X 0
#endif
#if BH_GELU_DEGREE == 5
// TWICE the GELU, for the fused blocks of the f16 modes (round 4): 2 GELU(v) = (v + |v|) - |v| 2 Phi(-|v|), with
// 2 Phi(-a) = exp2(Q(a) + 1) and Q(a) + 1 = a (c1 + a (c2 + ... + a c5)) -- the constant term is gone, so the Horner chain ends in a
// multiply -- and v + |v| = 2 max(v, 0) replaces the v_med3_f32 (1.8 v_fma_f32 at the occupancy these kernels run at) and the
// packed FMA that made |v| from it: |v| is a v_and_b32 per value.  Per PAIR of values: 2 v_and + 4 packed FMAs + 1 packed multiply +
// 2 v_exp + 1 packed add + 1 packed FMA = 11 instructions, as many as gelu_erf_fast4's (2 v_med3 + 7 packed FMAs + 2 v_exp), with
// the two dearest non-transcendental ones replaced by the two cheapest: 22.5 ns per pair and SIMD against 25.6
// (tools/microbench/valu_throughput.hip at 4 waves per SIMD, profiles/r4_d_valu_throughput.txt).  A scalar spelling with |v| as
// a free VOP3 source modifier (8 instructions per value instead of 5.5) was built first and measured SLOWER in isolation, 31.2 ns:
// at this occupancy the instruction count, not the issue cycles, is what the SIMD charges for.
// The factor two is a power of two and is folded on the host into the next linear stage (depthwise taps after the expand GELU,
// the exponent the project accumulators live at after the depthwise GELU: api.hip plan_fusion).  Same polynomial, same rounding
// points but the last: against float64 erfc the worst point is 0.63 of the activation's stated tolerance, as for gelu_erf_fast.
// gc = c_k 2^(-k s) when the argument arrives multiplied by 2^s (gelu_erf_fast4_scaled's convention), c_k otherwise.
__device__ __forceinline__ bh_f32x2 bh_abs2(bh_f32x2 v) {
    bh_f32x2 a;
    a[0] = __builtin_fabsf(v[0]); a[1] = __builtin_fabsf(v[1]);
    return a;
}
__device__ __forceinline__ void gelu2x_fast4(bh_f32x2 &v0, bh_f32x2 &v1, const GeluScaled &gc) {
    const bh_f32x2 a0 = bh_abs2(v0), a1 = bh_abs2(v1);
    bh_f32x2 q0 = __builtin_elementwise_fma(a0, BH_PK(gc.c5), BH_PK(gc.c4));
    bh_f32x2 q1 = __builtin_elementwise_fma(a1, BH_PK(gc.c5), BH_PK(gc.c4));
#define BH_GELU_STEP(c)                                       \
    q0 = __builtin_elementwise_fma(q0, a0, BH_PK(c));         \
    q1 = __builtin_elementwise_fma(q1, a1, BH_PK(c));
    BH_GELU_STEP(gc.c3) BH_GELU_STEP(gc.c2) BH_GELU_STEP(gc.c1)
#undef BH_GELU_STEP
    q0 = q0 * a0; q1 = q1 * a1;
    bh_f32x2 e0, e1;
    e0[0] = __builtin_amdgcn_exp2f(q0[0]); e0[1] = __builtin_amdgcn_exp2f(q0[1]);
    e1[0] = __builtin_amdgcn_exp2f(q1[0]); e1[1] = __builtin_amdgcn_exp2f(q1[1]);
    v0 = __builtin_elementwise_fma(-a0, e0, v0 + a0);
    v1 = __builtin_elementwise_fma(-a1, e1, v1 + a1);
}
__device__ __forceinline__ bh_f32x2 gelu2x_fast2(bh_f32x2 v, const GeluScaled &gc) {
    const bh_f32x2 a = bh_abs2(v);
    bh_f32x2 q = __builtin_elementwise_fma(a, BH_PK(gc.c5), BH_PK(gc.c4));
    q = __builtin_elementwise_fma(q, a, BH_PK(gc.c3));
    q = __builtin_elementwise_fma(q, a, BH_PK(gc.c2));
    q = __builtin_elementwise_fma(q, a, BH_PK(gc.c1));
    q = q * a;
    bh_f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(q[0]); e[1] = __builtin_amdgcn_exp2f(q[1]);
    return __builtin_elementwise_fma(-a, e, v + a);
}
// the unscaled polynomial's coefficients, for the host (c_k 2^-ks is computed there) and for the unscaled 2 GELU
constexpr float kGeluCoef[5] = {BH_GELU_C1, BH_GELU_C2, BH_GELU_C3, BH_GELU_C4, BH_GELU_C5};
constexpr GeluScaled kGeluUnscaled = {BH_GELU_C1, BH_GELU_C2, BH_GELU_C3, BH_GELU_C4, BH_GELU_C5};
#endif

// a + b as a plain v_add_f32 that hipcc cannot fuse with its neighbour.  Written as `fwd[j] + rev[-j]`
// in C, the folded-frame sums of the split-f16 mel kernel are SLP-vectorised into
// `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` (the second operand's halves swapped), and on gfx950 that
// form returns wrong sums while the same wave has v_mfma_f32_16x16x32_f16 in flight
// (tools/microbench/pk_add_opsel.hip: 7 % of the lanes, different ones every run; never without the MFMAs,
// never for two v_add_f32).  In the mel kernel that was one wrong 16-frame tile in 10^5 ... 10^6.
__device__ __forceinline__ float bh_add_unpacked(float a, float b) {
    float r;
    asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// a = hi + lo with hi, lo in f16 (22 significant bits between them), for a pair: products
// hi*hi + hi*lo + lo*hi on the f16 MFMA with f32 accumulation reproduce an f32 fmaf chain to ~1e-7 of
// sum|a b| (tools/microbench/mfma_f16_overlap.hip).  Valid while |a| < 65504 (f16 range).
// 4 instructions per pair: v_cvt_pk_f16_f32 (RNE), two v_fma_mix_f32 (lo = a - hi, the f16 operand
// widened inside the FMA, exact) and a second v_cvt_pk_f16_f32.
typedef _Float16 bh_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bh_split2(float v0, float v1, bh_f16x2 &hi, bh_f16x2 &lo) {
    hi = __builtin_convertvector((bh_f32x2){v0, v1}, bh_f16x2);
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi), "v"(v0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi), "v"(v1));
    lo = __builtin_convertvector((bh_f32x2){l0, l1}, bh_f16x2);
}
typedef _Float16 bh_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void bh_split8(const float (&v)[8], bh_f16x8 &hi, bh_f16x8 &lo) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        bh_f16x2 h, l;
        bh_split2(v[j], v[j + 1], h, l);
        hi[j] = h[0]; hi[j + 1] = h[1]; lo[j] = l[0]; lo[j + 1] = l[1];
    }
}

__device__ __forceinline__ float act_apply_slow(float v, int act) {
    switch (act) {
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_RELU6: return fminf(fmaxf(v, 0.f), 6.f);
    case ACT_SWISH: return v / (1.0f + expf(-v));
    case ACT_GELU_ERF: return gelu_erf_fast(v);
    case ACT_GELU_TANH: return 0.5f * v * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v)));
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
    }
}

// Compile-time activations for the MFMA kernels' epilogues (a run-time switch inlined at ~40 call sites blew the fused block
// kernel up to >100 KB of code: instruction-cache misses in the depthwise loop).  The fast forms cover what EfficientNet /
// MobileNet-style exports use between the convolutions (ONNX Erf-GELU, Mul(Sigmoid) = swish, Clip(0, 6), Relu); anything else
// takes the exact run-time form.  swish: v / (1 + 2^(-v log2 e)) through v_exp_f32 + v_rcp_f32 (1 ulp each).
template <int ACT>
__device__ __forceinline__ float bh_act(float v) {
    if constexpr (ACT == ACT_NONE) return v;
    else if constexpr (ACT == ACT_GELU_ERF) return gelu_erf_fast(v);
    else if constexpr (ACT == ACT_RELU) return bh_relu1(v);
    else if constexpr (ACT == ACT_RELU6) return __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f);
    else if constexpr (ACT == ACT_SWISH) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
    else return act_apply_slow(v, ACT);
}
template <int ACT>
__device__ __forceinline__ bh_f32x2 bh_act2(bh_f32x2 v) {
    if constexpr (ACT == ACT_GELU_ERF) return gelu_erf_fast2(v);
    else { bh_f32x2 r; r[0] = bh_act<ACT>(v[0]); r[1] = bh_act<ACT>(v[1]); return r; }
}
template <int ACT>
__device__ __forceinline__ void bh_act4(bh_f32x2 &v0, bh_f32x2 &v1) {
    if constexpr (ACT == ACT_GELU_ERF) gelu_erf_fast4(v0, v1);
    else { v0 = bh_act2<ACT>(v0); v1 = bh_act2<ACT>(v1); }
}
// the activations the MFMA epilogues are instantiated for (layers with any other one run on the f32 layer kernels)
inline bool act_is_templated(int act) { return act == ACT_GELU_ERF || act == ACT_SWISH || act == ACT_RELU6; }

// One STFT/mel branch of the front-end, with the Hann window, the
// real-part DFT and the mel projection folded into one operator Gf[K = L/2][n_mels_pad].
struct BranchParams {
    const float *gf;  // device, MFMA-fragment-major gfF[K/16][nm_pad/16][64 lanes][4]: element
                      // (g, mt, lane, c) = Gf[k = 16g + 4(lane>>4) + c][mel = 16mt + (lane&15)],
                      // Gf row k <-> sample offset n = k + 1 (last row halved)
    int L, H, K;      // frame length, hop, folded depth L/2
    int n_mels, nm_pad, n_frames;
    float expo;       // 1 / (1 + exp(mag_scale))
    float out_scale, out_shift;
    int flip;
    float log2_bias;  // f16 operator planes hold Gf * 2^s (s puts max |Gf| in [2^13, 2^14): both f16 halves of every
                      // entry that matters stay normal numbers); the power law undoes it inside its exp2:
                      // (v^2)^expo = exp2(expo * log2((v 2^s)^2) + log2_bias), log2_bias = -2 s expo.  0 for the f32 operator.
};
struct FrontendParams {
    BranchParams br[MAX_BRANCHES];
    int prec;  // 0: gf in f32 fragments (f32 MFMA); 3: gf as f16 hi / lo planes (split-f16 MFMA, 32-deep steps)
    int n_branches;
    int sample_count;
    float norm_eps;
};

// interleaved PCM16 on the device -> mono f32 segments starting at d_starts[i] (zero-padded tail)
// sample_format: BH_PCM_S16 = 1, BH_PCM_S24 = 2, BH_PCM_S32 = 3, BH_PCM_F32 = 4 (birda_hip.h); d_pcm_origin = device byte address of frame 0
void launch_segment_pcm(const void *d_pcm_origin, int sample_format, size_t n_frames, int channels, const unsigned long long *d_starts,
                        int n_seg, int seg_len, float *d_out, size_t out_stride, hipStream_t s);
// minmax [n_seg][8][2]: min / max of eight slices of every segment; in_bad (nullable) [n_seg][8]: 1 where the slice holds an inf / NaN
void launch_minmax(const float *x, float *minmax, unsigned *in_bad, int n_seg, int sample_count, hipStream_t s);
size_t mel_lds_bytes(const FrontendParams &p);   // LDS the front-end launch needs (create refuses a model beyond 160 KB)
void launch_mel(const float *x, const float *minmax, float *spec, const FrontendParams &p,
                const FrontendParams *d_p, int n_seg, hipStream_t s);


struct ConvParams {
    int in_h, in_w, out_h, out_w, cin, cout, kh, kw, sh, sw, pad_t, pad_l, in_layout, act;
};
// direct conv for the small-Cin stem; w [kh][kw][cin][cout]
void launch_conv_direct(const float *in, const float *w, const float *b, float *out, const ConvParams &p,
                        int n_seg, hipStream_t s);
// depthwise conv NHWC; w [kh][kw][c]
void launch_dwconv(const float *in, const float *w, const float *b, float *out, const ConvParams &p,
                   int n_seg, hipStream_t s);
// C[M][N] = act(A[M][K] . W[K][ldw] + bias) (+ R); W rows padded to ldw (multiple of 4)
void launch_pw_gemm(const float *A, const float *W, const float *bias, const float *R, float *C, int M, int K,
                    int N, int ldw, int act, hipStream_t s);
// the same product on the f16 MFMA (terms = 3: hi / lo split operands, f32-grade; 1: plain f16); K % 32 == 0;
// Wf: fragment-major planes [K / 32][ceil(N / 16)]{hi, lo}[64 lanes][8 halves]
// The planes hold W * w_scale, w_scale = 1 / w_unscale an exact power of two chosen on the host (f16_weight_scale, api.hip)
// so that max |W w_scale| lies in [2^13, 2^14): the lo half of every weight that matters is a normal f16 (an unscaled
// He-normal weight at Cin 1152 is ~2^-5, its lo half a subnormal with 2^-24 absolute resolution); the epilogue computes
// acc * w_unscale + bias in one FMA.
bool pw_gemm16_supports(int K, int act);
void launch_pw_gemm16(const float *A, const void *Wf, const float *bias, const float *R, float *C, int M, int K, int N,
                      int act, int terms, float w_unscale, hipStream_t s);
// squeeze-excite blocks (round 5): the project convolution with A = D x gate (gate [M / rows_per_seg][K]; nullptr: plain) ...
void launch_pw_gemm_gated(const float *A, const float *gate, int rows_per_seg, const float *W, const float *bias, const float *R,
                          float *C, int M, int K, int N, int ldw, int act, hipStream_t s);
// ... on the f16 MFMA (no activation; K % 4 == 0, planes [ceil(K / 32)][ceil(N / 16)]{hi, lo}[64][8] zero-padded in K) ...
// (a_blocked: A in MbDesc::dblk's layout -- every kernel behind this entry reads either; pw_gemm16_gated_wants_blocked: the
//  shapes whose kernel is the faster for it, a property of the BLOCK, never of the launch)
bool pw_gemm16_gated_wants_blocked(int K, int N, int rows_per_seg);
void launch_pw_gemm16_gated(const float *A, const float *gate, int rows_per_seg, const void *Wf, const float *bias, const float *R,
                            float *C, int M, int K, int N, int terms, float w_unscale, int a_blocked, hipStream_t s);
// ... and the gate itself: pool (from the per-tile channel sums of mbconv pass A, part [n][tiles][C]) -> 1x1 (C -> Cr, act1) -> 1x1
// (Cr -> C, act2), one launch, fixed summation order
// the gate beyond 576 channels in two launches of sixteen-segment workgroups (hpart: scratch of n_seg x C floats)
bool se_gate16_supports(int C, int Cr);
void launch_se_gate16(const float *part, int tiles, int P, float *hpart, const float *W1, const float *b1, int ld1, int act1, const float *W2,
                      const float *b2, int ld2, int act2, float *gate, int n_seg, int C, int Cr, hipStream_t s);
bool se_gate_supports(int C, int Cr);
// (the same gate as three launches -- the pool, then the two dense layers as GEMMs over all segments: launches of many segments)
void launch_se_gate_gemm(const float *part, int tiles, int P, float *pooled, float *hidden, const float *W1, const float *b1, int ld1, int act1,
                         const float *W2, const float *b2, int ld2, int act2, float *gate, int n_seg, int C, int Cr, hipStream_t s);
void launch_se_gate(const float *part, int tiles, int P, const float *W1, const float *b1, int ld1, int act1, const float *W2, const float *b2,
                    int ld2, int act2, float *gate, int n_seg, int C, int Cr, hipStream_t s);
// global average pool [n][P][C] -> [n][C]
void launch_gap(const float *in, float *out, int n_seg, int P, int C, hipStream_t s);
// squeeze-excite gate: out[n][p][c] = in[n][p][c] * gate[n][c]   (C % 4 == 0)
void launch_scale(const float *in, const float *gate, float *out, int n_seg, int P, int C, hipStream_t s);
// head 1x1 conv + activation (GELU / swish / ReLU6) + global average pool fused (f16 hi / lo weight planes as for launch_pw_gemm16)
bool head_gap16_supports(int P, int K, int N, int act);
void launch_head_gap16(const float *A, const void *Wf, const float *bias, float *out, int n_seg, int P, int K, int N,
                       int act, int terms, float w_unscale, hipStream_t s);
// activation + top-k over logits [n][n_classes] -> idx/conf [n][top_k]
// Post-filter of the kept top-k (reference apply_range_filter, classifier.rs:587-645): class_score (NaN = species without
// geomodel entry) selects geomodel_filter.rs:46-82, else species_keep the species-list retain (:617-640); both null = off.
struct TopkFilter {
    // BSG post-processing (reference classifier.rs:508-545): conf' = sigmoid(intercept[c] + slope[c] logit(conf)) (* prior[c]),
    // then re-sorted; applied before the range filter / species list stage.  bsg_intercept null = off.
    const float *bsg_intercept = nullptr, *bsg_slope = nullptr, *bsg_prior = nullptr;
    const float *class_score = nullptr;
    const unsigned char *species_keep = nullptr;
    float threshold = 0.f;
    int keep_unmatched = 1;
    int rerank = 0;
};
// in_bad (nullable): launch_minmax's flags [n_seg][8]; nonfinite (nullable): counter that receives +1 for every segment whose
// logits hold an inf / NaN although its samples were all finite (an operand left the f16 range on the way)
void launch_topk(const float *logits, int n_seg, int n_classes, int out_act, int top_k, float min_conf,
                 const TopkFilter &filter, int32_t *idx, float *conf, const unsigned *in_bad, unsigned *nonfinite, hipStream_t s);

// Polyphase resampler (resample.hip): rubato's FFT resampler as one dense operator on the MFMA.
struct ResamplePlan {
    uint32_t from, to;
    int hop, N, nblk, K, dmin;  // y[N m + p] = sum_{k < K} x[hop m + dmin + k] G[k][p]
    const float *d_op;          // device, fragment-major [nblk][K/16][10][64][4]
    const void *d_op16;         // the same operator as f16 hi / lo planes [2 nblk (80 phases each)][K/32][5]{hi, lo}[64 lanes][8 halves], k = 32 s + 8 (lane >> 4) + j
    float op16_unscale;         // the planes hold G * 2^s (max in [2^13, 2^14)); the kernel's store multiplies by 2^-s
};
void resample_sizes(uint32_t from, uint32_t to, int *fft_in, int *fft_out);
size_t resample_output_len(size_t n, uint32_t from, uint32_t to);  // rubato's output length for n inputs
const ResamplePlan *resample_plan(uint32_t from, uint32_t to, const char **err);
// d_in [n_seg][in_stride] (src_len valid samples each) -> d_out [n_seg][out_stride]: the first
// min(out_len, rubato length) samples are the resampled signal, the rest up to out_len zeros
// (`samples.resize(segment_samples, 0.0)`, reference src/pipeline/processor.rs:87)
// split_f16: the GEMM on the split-f16 MFMA (three v_mfma_f32_16x16x32_f16 per product, f32-grade sums) instead of the f32 MFMA
void launch_resample(const ResamplePlan &pl, const float *d_in, size_t in_stride, int src_len, float *d_out,
                     size_t out_stride, int out_len, int n_seg, bool split_f16, hipStream_t s);

// Fused MBConv block (kernels_mbconv.hip): expand 1x1 -> depthwise -> project 1x1 (+ residual).
struct MbDesc {
    const float *X, *R;  // input NHWC [n][H][W][Cin]; residual (nullable) shaped like Y
    float *Y;            // output NHWC [n][Ho][Wo][Cout]
    // Per-chunk weight blocks, each one contiguous LDS-DMA transfer (built by api.hip plan_fusion):
    // We: [chunk]{ fragments [KG][CE/16][64 lanes][4], element (g, j, lane, c) =
    //              We[k = 16g + 4(lane>>4) + c][n = ch*CE + 16j + (lane&15)] (0 for k >= Cin); be[CE] }
    // Wp: [chunk]{ fragments [CE/16][NTOP][64][4], element (g, j, lane, c) =
    //              Wp[k = ch*CE + 16g + 4(lane>>4) + c][n = 16j + (lane&15)] (0 for n >= Cout) }
    // Wd: [chunk]{ Wd[tap][CE]; bd[CE] }
    const float *We, *Wd, *Wp, *bp;
    int H, W, Cin, Cexp, Cout, Ho, Wo, pad_t, pad_l, KS, ST;
    int act_e, act_d, act_p;
    int prec;  // 0: f32 MFMA; 3: f16 hi/lo split (three f16 MFMAs per product, f32-grade); 1: plain f16 MFMA
    // f16 modes: the expand / project planes hold We * 2^se and Wp * 2^sp (powers of two that put the largest weight in
    // [2^13, 2^14), so both f16 halves of the weights are normal numbers), be and bp arrive multiplied alike;
    // the expand result is multiplied by e_unscale = 2^-se before its activation, the project accumulators start at
    // bp 2^sp + R p_scale and are stored times p_unscale.  All three are 1 in f32 mode.
    float e_unscale, p_scale, p_unscale;
    // e_fold != 0 (GELU blocks of the f16 modes): the expand result is NOT multiplied by e_unscale; its GELU runs on the scaled
    // value with the coefficients gelu (c_k 2^-ks, gelu_erf_fast4_scaled) and the depthwise taps in Wd carry the 2^-s instead
    int e_fold;
    GeluScaled gelu;
    // 1: a block without an expand convolution (depthwise -> project (+ residual), EfficientNet's expand-ratio-1 blocks): Cexp ==
    // Cin, X's channels are copied into the grid chunk by chunk, We holds nothing but zero biases (tile entries with KG = 0)
    int noexp;
    // stem variant (first block): "expand" = the k x k stride-s stem conv gathered from the planar
    // spectrogram X [n][stem_c][stem_h][stem_w]; then H, W are the stem's OUTPUT size and
    // Cin = stem_k * stem_k * stem_c im2col columns (We rows in [kh][kw][cin] order)
    int stem, stem_c, stem_h, stem_w, stem_k, stem_s, stem_pt, stem_pl;
    // diagnostic: 8 phase counters (wave-cycles: setup, dw-weight stage, P1, barrier, P2, barrier, P3,
    // epilogue) or nullptr
    unsigned long long *stamps;
    int dbg;  // ablation bits for tuning (BIRDA_HIP_MB_DBG): 1 no GELU in P1, 2 no P2, 4 no P3, 8 no P1 MFMA,
              // 16 no weight DMA, 32 no output store.  Results are wrong when non-zero.
    // filled by mb_plan()
    int cfg, CE, TH, S, tiles_y, tiles_x, IH, IW, KG, nchunks, NTOP, mpad_max;
    // floor(2^32 / tiles_x) + 1 and floor(2^32 / (tiles_x tiles_y)) + 1 (saturated at 2^32 - 1): the kernel decodes its tile index with
    // them instead of dividing (mbconv_kernel.hpp BH_MB_HOSTRCP)
    unsigned rcp_tiles_x, rcp_tiles_xy;
    int ring;  // 1: the chunk weights go through rings of LDS buffers (We x 2, Wp x 3, Wd x 2), refilled a whole chunk ahead
    size_t lds_bytes;
    // Squeeze-excite blocks (round 5; EfficientNet's gate between the depthwise and the project convolution: pool -> 1x1 -> act ->
    // 1x1 -> sigmoid -> scale).  The gate needs the pool of the WHOLE depthwise output, so the block runs as pass A (se = 1: this
    // kernel's expand and depthwise phases -- the depthwise output D goes to HBM once, NHWC [n][Ho][Wo][Cexp] at Dout, and every
    // workgroup leaves the per-channel sums of its tile's pixels in pool_part [n][tiles_y * tiles_x][Cexp]; no project GEMM),
    // se_gate_kernel (the partial sums in fixed order, the two dense layers) and a project GEMM whose A operand is D times the
    // gate (launch_pw_gemm*_gated, kernels_conv.hip).  The expanded tensor still never leaves the CU.
    int se;
    float *Dout, *pool_part;
    // dblk = 1: D is written BLOCKED -- [row tile of 16 pixels][Cexp / 16][16 rows][16 channels] (rows = n Ho Wo, channels a
    // multiple of 16, Ho Wo a multiple of 16): the 2 KB a wave of the project GEMM needs for one 32-deep step of a row tile are ONE
    // contiguous run, and a tile's steps follow each other -- DRAM sees sequential reads instead of 128-byte pieces a row apart (the
    // late blocks: 3.2-3.6 TB/s on NHWC), and pass A's stores are 1-KB runs instead of 64-byte ones.  Only pass A and the gated GEMMs
    // ever see D (api.hip forward_slice sets the flag for both).
    int dblk;
    // A squeeze-excite block WITHOUT an expand convolution (KG = 0; EfficientNet's expand-ratio-1 blocks: D is no larger than the
    // block input) is cheaper to compute twice than to keep: pass A runs with Dout = nullptr (channel sums only, nothing stored),
    // and the one-launch block (se = 0) takes gate != nullptr -- [n][Cexp], multiplied into the depthwise output in front of the
    // project phase's f16 split, where the gated GEMMs apply it too.  D never exists (api.hip forward_slice).
    const float *gate;
    // Channel split for launches of a few segments (round 6, BH_FLAG_LOW_LATENCY): ksplit > 1 runs the launch ksplit workgroups deep,
    // each walking 1 / ksplit of the chunks and leaving raw project accumulators in partial [ksplit][n][Ho Wo][Cout]; the caller
    // follows with mb_reduce_partials (parts added in index order, then x p_unscale): a rounding of its own, fixed for the regime.
    int ksplit;
    float *partial;
};
// Y[i] = (partial[0][i] + partial[1][i] + ...) * unscale, parts in index order
void launch_mb_reduce_partials(const float *partial, float *Y, int ksplit, size_t count, float unscale, hipStream_t s);
int mb_config_count();
int mb_config_name(int ci, char *out, size_t cap);
// picks the instantiation (force_cfg >= 0: that entry or fail) and fills the derived fields
bool mb_plan(MbDesc &d, int force_cfg);
bool mb_plan_twin(const MbDesc &d, MbDesc &twin);
bool mb_twin_sums_match(const MbDesc &d, const MbDesc &twin);   // squeeze-excite pass A: the twin's pooled sums are the block's, bit for bit   // one-segment-per-workgroup twin of a two-segment configuration (small launches)
bool mb_plan_narrow(const MbDesc &d, MbDesc &narrow);   // narrow-tile twin of a whole-image configuration (launches of a few segments)
void launch_mbconv(const MbDesc &d, int n_seg, hipStream_t s);

}  // namespace bh

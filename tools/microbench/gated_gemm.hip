// The gated project GEMMs of the squeeze-excite blocks (launch_pw_gemm16_gated, kernels_conv.hip) on synthetic operands: the product
// dispatch, then the row-streaming kernel (N = 64 .. 240) and the streaming kernel (N <= 48) with their phases switched off at
// compile time (their DBG template argument) -- which of {row loads, W pieces, W reads, MFMAs, epilogue, residual} a launch is made
// of -- and a phase clock around the main loop and the epilogue of every wave.  Includes kernels_conv.hip itself: the shipped code.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-inline-asm -Wno-unused-result -o tools/microbench/gated_gemm.bin tools/microbench/gated_gemm.hip
//   tools/microbench/gated_gemm.bin M K N rows_per_segment blocked        (e.g. 64000 1392 232 64 1; 15936000 24 24 15936 0)
#include "../../birda_amd/csrc/kernels_conv.hip"
#include <cstdio>
#include <vector>
int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64000, K = argc > 2 ? atoi(argv[2]) : 1392, N = argc > 3 ? atoi(argv[3]) : 232;
    const int P = argc > 4 ? atoi(argv[4]) : 64, blocked = argc > 5 ? atoi(argv[5]) : 0;
    const int steps = (K + 31) / 32, nt = (N + 15) / 16;
    float *A, *gate, *bias, *R, *C; void *W;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&gate, (size_t)(M / P) * K * 4); hipMalloc(&bias, N * 4);
    hipMalloc(&R, (size_t)M * N * 4); hipMalloc(&C, (size_t)M * N * 4 + 256 * 8 * 32); hipMalloc(&W, (size_t)steps * nt * 2048);
    std::vector<float> h((size_t)M * K);
    unsigned x = 12345;
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f; }
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(gate, h.data(), (size_t)(M / P) * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(R, h.data(), (size_t)M * N * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, h.data(), N * 4, hipMemcpyHostToDevice);
    std::vector<_Float16> hw((size_t)steps * nt * 1024);
    for (auto &v : hw) { x = x * 1664525u + 1013904223u; v = (_Float16)((float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f); }
    hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)M * K * 4 + (double)M * N * 8;
    const int n_rt = (M + 15) / 16;
    auto run = [&](const char *what, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(e0, s);
        const int reps = 10;
        for (int i = 0; i < reps; i++) launch();
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("M %d K %d N %d P %d blocked %d %-46s: %8.1f us  (%.2f TB/s of D + R + C)\n", M, K, N, P, blocked, what, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12);
    };
    run("product dispatch", [&] { bh::launch_pw_gemm16_gated(A, gate, P, W, bias, R, C, M, K, N, 3, 1.0f, blocked, s); });
    if (nt > 3) {      // the 128 x 128 staged tiles the dispatch keeps for small launches and N > 240
        int ntb = 8; long best = -1;
        for (int cand : {10, 8, 6}) { const long padded = (long)((nt + cand - 1) / cand) * cand; if (best < 0 || padded < best) { best = padded; ntb = cand; } }
        const int n_xb = (nt + ntb - 1) / ntb, n_yb = (M + 127) / 128;
        dim3 grid((unsigned)(n_xb * n_yb)), block(256);
#define GS(NTBV, BLKV)                                                                                                             \
        if (ntb == NTBV && (blocked != 0) == BLKV) {                                                                               \
            constexpr size_t lds = 2 * ((8 + NTBV) * 2 * 256) * sizeof(float);                                                     \
            (void)hipFuncSetAttribute((const void *)bh::pw_gemm16s_kernel<3, bh::ACT_NONE, true, NTBV, BLKV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            run("staged 128 x 128 tiles", [&] { hipLaunchKernelGGL((bh::pw_gemm16s_kernel<3, bh::ACT_NONE, true, NTBV, BLKV>), grid, block, lds, s, A, (const bh::f16x8 *)W, bias, R, C, M, K, N, nt, 1.0f, gate, P); }); \
        }
        GS(6, false) GS(6, true) GS(8, false) GS(8, true) GS(10, false) GS(10, true)
    }
#define GG(NTV, RBV, PFV, DBGV, what)                                                                                              \
    if (nt == NTV) {                                                                                                               \
        const int gs_max = (8 * RBV * 16 + P - 2) / P + 1;                                                                         \
        const size_t lds = std::max((size_t)PFV * NTV * 2048 + (size_t)gs_max * ((K + 31) / 32 * 32) * sizeof(float), (size_t)8 * 16 * NTV * 16 * sizeof(float)); \
        const int wgs = std::min((n_rt + 8 * RBV - 1) / (8 * RBV), bh::device_cu_count());                                         \
        (void)hipFuncSetAttribute((const void *)bh::pw_gemm16_wide_kernel<3, NTV, RBV, PFV, DBGV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        run(what, [&] { hipLaunchKernelGGL((bh::pw_gemm16_wide_kernel<3, NTV, RBV, PFV, DBGV>), dim3(wgs), dim3(512), lds, s, A, gate, P, (const bh::f16x8 *)W, bias, R, C, M, K, N, 1.0f, gs_max, blocked); }); \
    }
#define GG_ALL(NTV, RBV, PFV)                                                      \
    GG(NTV, RBV, PFV, 0, "all")                                                    \
    { const float *Rk = R; R = nullptr; GG(NTV, RBV, PFV, 0, "all, no residual") R = (float *)Rk; }   \
    GG(NTV, RBV, PFV, 1 | 2 | 4, "epilogue only")                                  \
    { const float *Rk = R; R = nullptr; GG(NTV, RBV, PFV, 1 | 2 | 4, "epilogue only, no residual") R = (float *)Rk; }   \
    GG(NTV, RBV, PFV, 8, "-epilogue")                                              \
    GG(NTV, RBV, PFV, 8 | 1, "-epilogue -mfma -Wreads")                            \
    GG(NTV, RBV, PFV, 8 | 16, "-epilogue -Wreads")                                 \
    GG(NTV, RBV, PFV, 8 | 32, "-epilogue -mfma")                                   \
    GG(NTV, RBV, PFV, 8 | 4, "-epilogue -Wpieces")                                 \
    GG(NTV, RBV, PFV, 8 | 4 | 16, "-epilogue -Wpieces -Wreads")                    \
    GG(NTV, RBV, PFV, 8 | 2, "-epilogue -rows")                                    \
    GG(NTV, RBV, PFV, 8 | 2 | 16, "-epilogue -rows -Wreads")                       \
    GG(NTV, RBV, PFV, 8 | 1 | 2, "-epilogue -rows -mfma -Wreads")                  \
    GG(NTV, RBV, PFV, 8 | 1 | 2 | 4, "-epilogue -rows -mfma -Wreads -Wpieces")
#define GG_STAMPS(NTV, RBV, PFV)                                                                                                    \
    if (nt == NTV) {                                                                                                               \
        GG(NTV, RBV, PFV, 64, "all, with the phase clock")                                                                         \
        std::vector<long long> st(256 * 8 * 4);                                                                                    \
        hipMemcpy(st.data(), C + (size_t)M * N, st.size() * 8, hipMemcpyDeviceToHost);                                             \
        const int wgs = std::min((n_rt + 8 * RBV - 1) / (8 * RBV), bh::device_cu_count());                                         \
        double a[3] = {0, 0, 0}, mx[3] = {0, 0, 0};                                                                               \
        for (int w = 0; w < wgs * 8; w++) for (int k = 0; k < 3; k++) { a[k] += (double)st[w * 4 + k]; mx[k] = std::max(mx[k], (double)st[w * 4 + k]); } \
        printf("  last pass of every wave, cycles (mean / max): main loop %.0f / %.0f, barrier %.0f / %.0f, epilogue %.0f / %.0f\n",  \
               a[0] / (wgs * 8), mx[0], a[1] / (wgs * 8), mx[1], a[2] / (wgs * 8), mx[2]);                                          \
    }
    // the streaming kernel of the early blocks (N <= 48, all of W in LDS)
#define GT(NTV, SH, DBGV, what)                                                                                                        \
    if (nt == NTV && nt <= 3) {                                                                                                    \
        const size_t w_bytes = (size_t)steps * NTV * 2 * 1024;                                                                     \
        const size_t lds = w_bytes + (size_t)8 * (SH ? (NTV <= 2 ? 3 : 2) : (NTV <= 2 ? 4 : 3)) * 16 * NTV * 16 * sizeof(float);                               \
        const int wgs = std::min((n_rt + 31) / 32, 2 * bh::device_cu_count());                                                     \
        (void)hipFuncSetAttribute((const void *)bh::pw_gemm16_thin_kernel<3, NTV, SH, DBGV>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); \
        run(what, [&] { hipLaunchKernelGGL((bh::pw_gemm16_thin_kernel<3, NTV, SH, DBGV>), dim3(wgs), dim3(512), lds, s, A, gate, P, (const bh::f16x8 *)W, bias, R, C, M, K, N, 1.0f); }); \
    }
#define GT_ALL(NTV, SH, tag)                                                       \
    GT(NTV, SH, 0, tag "streaming kernel, all")                                    \
    { const float *Rk = R; R = nullptr; GT(NTV, SH, 0, tag "streaming kernel, no residual") R = (float *)Rk; } \
    GT(NTV, SH, 1, tag "streaming kernel -mfma")                                   \
    GT(NTV, SH, 8, tag "streaming kernel -epilogue")                               \
    GT(NTV, SH, 9, tag "streaming kernel -mfma -epilogue (row loads only)")
    GT_ALL(1, false, "") GT_ALL(2, false, "") GT_ALL(3, false, "")
    GT_ALL(1, true, "2 wg/CU: ") GT_ALL(2, true, "2 wg/CU: ") GT_ALL(3, true, "2 wg/CU: ")
    GG_STAMPS(15, 2, 3)
    GG_STAMPS(9, 2, 4)
    GG_STAMPS(6, 3, 3)
    GG_ALL(15, 2, 3)
    GG_ALL(9, 2, 4)
    GG_ALL(6, 3, 3)
    return 0;
}

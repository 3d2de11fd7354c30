// The squeeze-excite gate launches (kernels_conv.hip: se_gate_kernel; se_hidden_kernel + se_gate16_kernel) on synthetic operands:
// what a gate of C channels / Cr hidden units costs over n segments whose channel sums arrive in `tiles` per-tile rows.
// Includes kernels_conv.hip itself: the shipped code.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-inline-asm -Wno-unused-result -o tools/microbench/se_gate.bin tools/microbench/se_gate.hip
//   tools/microbench/se_gate.bin n C Cr tiles
#include "../../birda_amd/csrc/kernels_conv.hip"
#include <cstdio>
#include <vector>
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1000, C = argc > 2 ? atoi(argv[2]) : 1392, Cr = argc > 3 ? atoi(argv[3]) : 58, tiles = argc > 4 ? atoi(argv[4]) : 4;
    float *part, *W1, *W2, *b1, *b2, *gate, *hpart;
    const int ld1 = (Cr + 3) / 4 * 4, ld2 = (C + 3) / 4 * 4;
    hipMalloc(&part, (size_t)n * tiles * C * 4); hipMalloc(&W1, (size_t)C * ld1 * 4); hipMalloc(&W2, (size_t)Cr * ld2 * 4);
    hipMalloc(&b1, Cr * 4); hipMalloc(&b2, C * 4); hipMalloc(&gate, (size_t)n * C * 4); hipMalloc(&hpart, (size_t)n * C * 4);
    std::vector<float> h((size_t)n * tiles * C);
    unsigned x = 12345;
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f; }
    hipMemcpy(part, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W1, h.data(), (size_t)C * ld1 * 4, hipMemcpyHostToDevice); hipMemcpy(W2, h.data(), (size_t)Cr * ld2 * 4, hipMemcpyHostToDevice);
    hipMemcpy(b1, h.data(), Cr * 4, hipMemcpyHostToDevice); hipMemcpy(b2, h.data(), C * 4, hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *what, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(e0, s);
        const int reps = 20;
        for (int i = 0; i < reps; i++) launch();
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("n %d C %d Cr %d tiles %d %-40s: %8.1f us\n", n, C, Cr, tiles, what, ms * 1e3 / reps);
    };
    if (bh::se_gate_supports(C, Cr))
        run("se_gate_kernel (a workgroup a segment)", [&] { bh::launch_se_gate(part, tiles, 64, W1, b1, ld1, 3, W2, b2, ld2, 4, gate, n, C, Cr, s); });
    if (bh::se_gate16_supports(C, Cr)) {
        run("se_hidden + se_gate16", [&] { bh::launch_se_gate16(part, tiles, 64, hpart, W1, b1, ld1, 3, W2, b2, ld2, 4, gate, n, C, Cr, s); });
        const bh::SeHiddenShape sh = bh::se_hidden_shape(C, Cr);
        const int ksn = sh.slices, slice = sh.slice, groups = (n + bh::SE_SG - 1) / bh::SE_SG;
        const size_t lds = ((size_t)slice * bh::SE_SG + 256 * bh::SE_SG) * sizeof(float);
        run("  se_hidden_kernel alone", [&] { hipLaunchKernelGGL(bh::se_hidden_kernel, dim3(groups, ksn), dim3(256), lds, s, part, tiles, 1.0f / 64, W1, ld1, hpart, n, C, Cr, slice); });
        run("  se_gate16_kernel alone", [&] { hipLaunchKernelGGL(bh::se_gate16_kernel, dim3(groups, (C + 255) / 256), dim3(256), 0, s, hpart, ksn, b1, 3, W2, b2, ld2, 4, gate, n, C, Cr); });
    }
    return 0;
}

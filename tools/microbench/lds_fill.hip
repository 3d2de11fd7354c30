// Cost of filling LDS with a 52-KB weight block per iteration (the late fused blocks' per-chunk weights), one workgroup of 4 waves
// per CU (1 wave per SIMD): LDS-DMA (global_load_lds_dwordx4, what mbconv_kernel uses) against global_load_dwordx4 into
// registers + ds_write_b128.  MODE 0: nothing (the loop skeleton), 1: LDS-DMA, 2: registers + ds_write, each with and without
// independent MFMA work between issue and wait.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/lds_fill.hip -o /tmp/lds_fill && /tmp/lds_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int NP = 52;            // 1-KiB pieces per block of weights
constexpr int PPW = NP / 4;       // pieces per wave

template <int MODE, int WORK>
__global__ __launch_bounds__(256, 1) void k(const float *__restrict__ w, float *out, int iters, int nblk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc[8];
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { acc[i] = (f32x4){0, 0, 0, 0}; a[i] = (_Float16)(tid * 1e-3f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
    float s = 0;
    for (int it = 0; it < iters; it++) {
        const float *src = w + (size_t)((it + blockIdx.x) % nblk) * NP * 256;
        float4 r[PPW];
        if (MODE == 1) {
#pragma unroll
            for (int p = 0; p < PPW; p++) {
                const int piece = p * 4 + wave;
                const unsigned la = (unsigned)(size_t)(smem + piece * 256);
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                             :: "v"(src + piece * 256 + lane * 4), "s"(__builtin_amdgcn_readfirstlane(la)) : "memory", "m0");
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int p = 0; p < PPW; p++) r[p] = *reinterpret_cast<const float4 *>(src + (p * 4 + wave) * 256 + lane * 4);
        }
        if (WORK) {
#pragma unroll
            for (int rep = 0; rep < WORK; rep++)
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
        }
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == 2) {
#pragma unroll
            for (int p = 0; p < PPW; p++) *reinterpret_cast<float4 *>(smem + (p * 4 + wave) * 256 + lane * 4) = r[p];
        }
        __syncthreads();
        s += smem[(tid * 17 + it) & (NP * 256 - 1)];
        __syncthreads();
    }
    for (int i = 0; i < 8; i++) s += acc[i][0];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int WORK>
void run(const char *name, const float *w, float *out, int iters, int nblk) {
    const size_t smem = 100 * 1024;   // one workgroup per CU
    (void)hipFuncSetAttribute((const void *)k<MODE, WORK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, WORK>), dim3(256), dim3(256), smem, 0, w, out, iters, nblk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-64s %8.3f ms  %7.1f ns per iteration\n", name, best, best * 1e6 / iters);
}

int main() {
    const int nblk = 36, iters = 20000;
    float *w, *out;
    (void)hipMalloc(&w, (size_t)nblk * NP * 256 * sizeof(float));
    (void)hipMemset(w, 0, (size_t)nblk * NP * 256 * sizeof(float));
    (void)hipMalloc(&out, 256 * 256 * sizeof(float));
    printf("52 KB of weights -> LDS per iteration, 1 workgroup (4 waves) per CU, 36 blocks cycling through L2\n");
    run<0, 0>("skeleton (two barriers, one LDS read)", w, out, iters, nblk);
    run<1, 0>("LDS-DMA (13 global_load_lds_dwordx4 per wave), wait at once", w, out, iters, nblk);
    run<2, 0>("13 global_load_dwordx4 + 13 ds_write_b128 per wave", w, out, iters, nblk);
    run<0, 8>("skeleton + 64 MFMA (16x16x32 f16) per wave", w, out, iters, nblk);
    run<1, 8>("LDS-DMA + 64 MFMA between issue and wait", w, out, iters, nblk);
    run<2, 8>("loads + 64 MFMA + ds_writes", w, out, iters, nblk);
    run<0, 24>("skeleton + 192 MFMA per wave", w, out, iters, nblk);
    run<1, 24>("LDS-DMA + 192 MFMA between issue and wait", w, out, iters, nblk);
    run<2, 24>("loads + 192 MFMA + ds_writes", w, out, iters, nblk);
    return 0;
}

// f16 MFMA (v_mfma_f32_16x16x32_f16) rate, overlap with f32 VALU, and accuracy of the 3-term
// hi/lo split (a = ah + al in f16: ah*bh + ah*bl + al*bh, f32 accumulate) against an f64 product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int R>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    f32x4 acc[8]; float v[8]; f16x8 a, b;
    for (int i = 0; i < 8; i++) { acc[i] = (f32x4){0, 0, 0, 0}; v[i] = threadIdx.x * 0.001f + i; a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
    const float fa = threadIdx.x * 1e-3f, fb = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE != 1) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
            if (MODE != 0) {
#pragma unroll
                for (int r = 0; r < R; r++) v[(j + r) & 7] = __builtin_fmaf(v[(j + r) & 7], fb, fa);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int R>
float run(int threads, int iters, float *d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, R>), dim3(256), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, R>), dim3(256), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

// accuracy: C[16x16] = A[16xK] B[Kx16] with the split, one wave
__global__ void acc_kernel(const float *A, const float *B, float *C, int K, int terms) {
    const int l = threadIdx.x, row = l & 15, kq = l >> 4;
    f32x4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 32) {
        f16x8 ah, al, bh, bl;
        for (int j = 0; j < 8; j++) {
            const float a = A[row * K + k0 + 8 * kq + j], b = B[(k0 + 8 * kq + j) * 16 + row];
            ah[j] = (_Float16)a; al[j] = (_Float16)(a - (float)ah[j]);
            bh[j] = (_Float16)b; bl[j] = (_Float16)(b - (float)bh[j]);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
        if (terms >= 3) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 4; r++) C[(4 * kq + r) * 16 + row] = acc[r];
}

int main() {
    float *d; (void)hipMalloc(&d, 256 * 512 * 4);
    const int iters = 20000;
    for (int threads : {256, 512}) {
        const int wps = threads / 256;
        float m = run<0, 1>(threads, iters, d), v = run<1, 4>(threads, iters, d), b = run<2, 4>(threads, iters, d);
        printf("%d wave(s)/SIMD: f16 MFMA only %.3f ms (%.1f cyc/MFMA/SIMD), VALU R=4 only %.3f ms, both in one wave %.3f ms (serial would be %.3f)\n",
               wps, m, m * 1e-3 * 2.4e9 / (iters * 8.0 * wps), v, b, m + v);
    }
    // accuracy
    for (int K : {32, 192, 1152}) {
        std::vector<float> A(16 * K), B(K * 16), C(256);
        srand(1);
        for (auto &x : A) x = (float)((rand() / (double)RAND_MAX) * 2 - 1) * 1.7f;     // activations O(1)
        for (auto &x : B) x = (float)((rand() / (double)RAND_MAX) * 2 - 1) * 0.2f;     // weights
        float *dA, *dB, *dC; (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, 1024);
        (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        for (int terms : {1, 3}) {
            hipLaunchKernelGGL(acc_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, terms);
            (void)hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
            double emax = 0, smax = 0, e32 = 0;
            for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
                double ref = 0, sabs = 0; float f = 0;
                for (int kk = 0; kk < K; kk++) { ref += (double)A[i * K + kk] * B[kk * 16 + j]; sabs += fabs((double)A[i * K + kk] * B[kk * 16 + j]); f = fmaf(A[i * K + kk], B[kk * 16 + j], f); }
                emax = fmax(emax, fabs(C[i * 16 + j] - ref) / sabs); e32 = fmax(e32, fabs((double)f - ref) / sabs); smax = fmax(smax, sabs);
            }
            printf("K=%4d terms=%d: max |err| / sum|a b| = %.3e   (f32 fmaf chain: %.3e)\n", K, terms, emax, e32);
        }
    }
    return 0;
}

// Host-side gather rates on the GPU box: N independent pageable f32 slices of 576 000 B -> one pinned staging
// buffer, with T worker threads, by glibc memcpy and by non-temporal AVX2 stores; and pinned -> device on top.
//   hipcc -O2 -o /tmp/host_gather tools/microbench/host_gather.cpp -lpthread && /tmp/host_gather
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__attribute__((target("avx2"))) static void copy_nt(float *dst, const float *src, size_t n) {
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256 a = _mm256_loadu_ps(src + i), b = _mm256_loadu_ps(src + i + 8), c = _mm256_loadu_ps(src + i + 16), d = _mm256_loadu_ps(src + i + 24);
        _mm256_stream_ps(dst + i, a); _mm256_stream_ps(dst + i + 8, b); _mm256_stream_ps(dst + i + 16, c); _mm256_stream_ps(dst + i + 24, d);
    }
    for (; i < n; i++) dst[i] = src[i];
    _mm_sfence();
}

int main() {
    const size_t S = 144000, N = 512;
    std::vector<float *> slices(N);
    for (size_t i = 0; i < N; i++) { slices[i] = (float *)malloc(S * 4); for (size_t k = 0; k < S; k += 256) slices[i][k] = (float)k; memset(slices[i], 1, S * 4); }
    float *pinned = nullptr, *plain = (float *)aligned_alloc(4096, N * S * 4), *dev = nullptr;
    if (hipHostMalloc((void **)&pinned, N * S * 4, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
    hipMalloc((void **)&dev, N * S * 4);
    memset(plain, 0, N * S * 4); memset(pinned, 0, N * S * 4);
    printf("hardware threads %u\n", std::thread::hardware_concurrency());
    for (int dstk = 0; dstk < 2; dstk++)
        for (int nt = 0; nt < 2; nt++)
            for (unsigned T : {1u, 2u, 4u, 8u, 12u, 16u, 24u, 32u}) {
                float *dst = dstk ? pinned : plain;
                double best = 1e9;
                for (int rep = 0; rep < 5; rep++) {
                    std::atomic<size_t> next{0};
                    const double t0 = now();
                    std::vector<std::thread> th;
                    for (unsigned t = 0; t < T; t++)
                        th.emplace_back([&] {
                            for (size_t i; (i = next.fetch_add(1)) < N;) { if (nt) copy_nt(dst + i * S, slices[i], S); else memcpy(dst + i * S, slices[i], S * 4); }
                        });
                    for (auto &x : th) x.join();
                    best = std::min(best, now() - t0);
                }
                printf("%-7s %-6s T=%2u  %6.2f ms  %6.1f GB/s\n", dstk ? "pinned" : "malloc", nt ? "nt" : "memcpy", T, best * 1e3, N * S * 4 / best / 1e9);
            }
    // pinned -> device in 32-segment pieces
    hipStream_t s; hipStreamCreate(&s);
    for (int rep = 0; rep < 3; rep++) {
        const double t0 = now();
        for (size_t i = 0; i < N; i += 32) hipMemcpyAsync(dev + i * S, pinned + i * S, 32 * S * 4, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        const double t = now() - t0;
        printf("H2D pinned 32-segment pieces: %6.2f ms %6.1f GB/s\n", t * 1e3, N * S * 4 / t / 1e9);
    }
    return 0;
}

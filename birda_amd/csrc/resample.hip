// Polyphase resampler for gfx950: the reference's per-segment rate conversion
// (reference src/audio/resample.rs:10-91 -> rubato 4.0.0 `Fft<f32>`, FixedSync::Both, chunk 1024;
// rubato is not vendored, Cargo.lock:2300-2303) as ONE dense polyphase operator on the f32 MFMA.
//
// What rubato's synchronous FFT resampler computes per block (fft_in samples in, fft_out out):
// zero-pad to 2 fft_in, rFFT, multiply by the spectrum of a BlackmanHarris^2-windowed sinc of
// length fft_in, keep min(fft_in + 1, fft_out) bins, irFFT at 2 fft_out, overlap-add the halves.
// Every step is linear, and the composite is shift-invariant with period (P, Q) = (from, to)/gcd
// (measured: |g(n + Q, i + P) - g(n, i)| <= 5e-11 upsampling, 4e-7 for 48 k -> 32 k, relative to
// max |g| ~ 0.7), so a segment's output is
//     y[N m + p] = sum_d x[hop m + d] t_p[d],   N = lcm(Q, 160), hop = N P / Q
// i.e. a [N x K] x [K x frames] GEMM per segment over overlapping frames of hop `hop` -- the
// same shape as the STFT x mel front-end.  The taps t_p[d] are read off rubato's exact block
// operator, built here in double precision on unit impulses at plan time (once per rate pair),
// so the start-up transient, the half-block delay and the zero-padded last block all come out as
// in the reference; taps below 1e-9 max|t| are dropped (K ~ fft_in + hop when upsampling).
#include <cmath>
#include <complex>
#include <map>
#include <mutex>
#include <numeric>
#include <tuple>
#include <vector>

#include "kernels.hpp"

namespace bh {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

typedef std::complex<double> cd;

// generic mixed-radix complex FFT (decimation in time on the smallest prime factor), double
void fft_rec(const cd *in, cd *out, int n, int stride, int sign) {
    if (n == 1) { out[0] = in[0]; return; }
    int p = 2;
    while (n % p) p++;
    const int m = n / p;
    std::vector<cd> sub((size_t)n);
    for (int r = 0; r < p; r++) fft_rec(in + (size_t)r * stride, sub.data() + (size_t)r * m, m, stride * p, sign);
    for (int k = 0; k < m; k++)
        for (int q = 0; q < p; q++) {
            const int ko = k + q * m;
            cd acc = 0;
            for (int r = 0; r < p; r++) {
                const double ang = sign * 2.0 * M_PI * (double)((long long)r * ko % n) / (double)n;
                acc += sub[(size_t)r * m + k] * cd(std::cos(ang), std::sin(ang));
            }
            out[ko] = acc;
        }
}
std::vector<cd> fft(const std::vector<cd> &x, int sign) {
    std::vector<cd> y(x.size());
    fft_rec(x.data(), y.data(), (int)x.size(), 1, sign);
    return y;
}

double blackman_harris2(int i, int n) {
    const double a0 = 0.35875, a1 = 0.48829, a2 = 0.14128, a3 = 0.01168;
    const double x = 2.0 * M_PI * (double)i / (double)n;
    const double w = a0 - a1 * std::cos(x) + a2 * std::cos(2 * x) - a3 * std::cos(3 * x);
    return w * w;
}

constexpr size_t kResampleLdsBytes = 150 * 1024;          // what a workgroup's input span may take of the CU's 160 KB
std::mutex g_plan_mu;
std::map<std::tuple<int, uint32_t, uint32_t>, ResamplePlan> g_plans;  // the operator lives in ONE device's memory: keyed by device ordinal too

}  // namespace

void resample_sizes(uint32_t from, uint32_t to, int *fft_in, int *fft_out) {
    const uint32_t g = std::gcd(from, to);
    const uint32_t min_in = from / g;
    const uint32_t chunks = (uint32_t)std::ceil(1024.0 / (double)min_in);
    // (a header may name any 32-bit rate: the products saturate instead of wrapping -- resample_plan refuses such a pair by size)
    const uint64_t fi = (uint64_t)chunks * (from / g), fo = (uint64_t)chunks * (to / g);
    *fft_in = (int)std::min<uint64_t>(fi, 1u << 30);
    *fft_out = (int)std::min<uint64_t>(fo, 1u << 30);
}

size_t resample_output_len(size_t n, uint32_t from, uint32_t to) {
    if (from == to) return n;
    int fi, fo;
    resample_sizes(from, to, &fi, &fo);
    const size_t full = n / (size_t)fi, rem = n - full * (size_t)fi;
    size_t out = full * (size_t)fo;
    if (rem) out += std::min((size_t)fo, (size_t)std::ceil((double)rem * (double)to / (double)from));
    return out;
}

// Builds (or returns the cached) plan; *err receives a message on failure.
const ResamplePlan *resample_plan(uint32_t from, uint32_t to, const char **err) {
    std::lock_guard<std::mutex> lock(g_plan_mu);
    auto key = std::make_tuple(current_device(), from, to);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return &it->second;
    int ni, no;
    resample_sizes(from, to, &ni, &no);
    const uint32_t g = std::gcd(from, to);
    const int P = (int)(from / g), Q = (int)(to / g);
    // Two forms of the same operator.  POLYPHASE (up-sampling and decimation by at most 1.5): the composite is shift-invariant with
    // period (P, Q), a frame is `hop` = q P input samples and N = q Q = lcm(Q, 160) output phases, the taps come from impulses in the
    // first `hop` positions of a block.  BLOCK (round 6; decimation by more than 1.5 -- 88.2 / 96 / 192 / 256 / 384 kHz recordings):
    // rubato's block is NOT shift-invariant there.  It truncates the 2 fft_in-point spectrum to 2 fft_out bins -- a brick wall that
    // is circular in time -- and what leaks past a block's end is overlap-added into the NEXT block's output: two recordings that
    // differ by a shift of `hop` samples come out 1e-4 (96 -> 48 kHz) to 5e-3 (384 -> 48 kHz) apart in the oracle itself, and the
    // polyphase form was that far from it (untested before: the four pairs of tests/test_resampler_gpu.py all decimate by <= 1.5).
    // What IS invariant is a shift by a whole block: a frame is one block (hop = fft_in input samples, N = fft_out outputs), K spans
    // the previous and the current block, and tap() below -- written for the polyphase form -- is then exact.
    const bool block_form = (uint64_t)from * 2 > (uint64_t)to * 3;
    const int N = block_form ? no : std::lcm(Q, 160), q = block_form ? 0 : N / Q, hop = block_form ? ni : q * P;
    if (!block_form && (hop > ni || N > 2 * no)) { *err = "resampler: rate pair outside the built range"; return nullptr; }
    // Refused BEFORE the operator is computed (round 6: found by tools/fuzz_wav_decoder.py -- a header that says 47 999 Hz, or 128 Hz, or
    // 1.5 GHz, asked for `hop` FFTs of length 2 fft_out and a table of hop x 2 fft_out doubles (36 GB for 47 999 -> 48 000) before the
    // LDS check below could say no: the process sat in here until the inference watchdog killed it, one bad file ending a whole
    // directory run).  The kernel's frame tile holds (16 FT - 1) hop + K (>= 128) input samples in LDS, FT = 1 at the least; every pair whose rates share a divisor
    // of a few hundred -- 8 / 11.025 / 16 / 22.05 / 32 / 44.1 / 88.2 / 96 kHz against 32 / 48 kHz -- has hop <= 480; the recorders that
    // sample at 192 / 256 / 300 / 384 kHz have 640-1 920 and take fewer frames a workgroup (launch_resample).
    if (((size_t)15 * hop + 128) * 4 + 32 > kResampleLdsBytes) { *err = "resampler: frame span exceeds LDS (the two sample rates share too small a divisor)"; return nullptr; }
    if (ni > 32768 || no > 32768) { *err = "resampler: rate ratio outside the built range"; return nullptr; }
    // rubato's filter: windowed sinc of length fft_in, unit sum, scaled 1 / (2 fft_in)
    const double cutoff = ni > no ? (double)powf(0.4f, 16.0f / (float)ni) * (double)no / (double)ni
                                  : (double)powf(0.4f, 16.0f / (float)ni);
    std::vector<cd> sinc((size_t)2 * ni, 0.0);
    double sum = 0.0;
    for (int i = 0; i < ni; i++) {
        const double x = (double)i - (double)(ni / 2), arg = M_PI * cutoff * x;
        const double s = std::fabs(arg) < 1e-12 ? 1.0 : std::sin(arg) / arg;
        sinc[i] = s * blackman_harris2(i, ni);
        sum += sinc[i].real();
    }
    for (int i = 0; i < ni; i++) sinc[i] = sinc[i].real() / sum / (2.0 * ni);
    const std::vector<cd> Hf = fft(sinc, -1);
    const int new_len = ni < no ? ni + 1 : no;
    // R[i][n]: the block operator's response at output n (0 .. 2 fft_out) to an impulse at input i
    std::vector<double> R((size_t)hop * 2 * no);
    std::vector<cd> B((size_t)2 * no);
    for (int i = 0; i < hop; i++) {  // only impulse positions [0, hop) are ever looked up
        std::fill(B.begin(), B.end(), cd(0, 0));
        for (int k = 0; k < new_len; k++) {
            const double ang = -2.0 * M_PI * (double)((long long)k * i % (2 * ni)) / (double)(2 * ni);
            B[k] = cd(std::cos(ang), std::sin(ang)) * Hf[k];
        }
        B[0] = cd(B[0].real(), 0.0);
        if (new_len > no) B[no] = cd(B[no].real(), 0.0);
        for (int k = 1; k < no; k++) B[2 * no - k] = std::conj(B[k]);
        const std::vector<cd> b = fft(B, +1);
        for (int n = 0; n < 2 * no; n++) R[(size_t)i * 2 * no + n] = b[n].real();
    }
    // taps t_p[d] = R[hop m + d][N m + p] with m = ceil(-d / hop)
    const int d_lo = -((2 * no + N - 1) / N + 1) * hop, d_hi = hop;  // scan range [d_lo, d_hi)
    auto tap = [&](int p, int d) -> double {
        const int m = d >= 0 ? -(d / hop) : (-d + hop - 1) / hop;
        const int i = hop * m + d, n = N * m + p;
        if (p >= N || i < 0 || i >= hop || n < 0 || n >= 2 * no) return 0.0;   // (p >= N: the padding columns of the last 160-phase block)
        return R[(size_t)i * 2 * no + n];
    };
    double tmax = 0.0;
    for (int p = 0; p < N; p++)
        for (int d = d_lo; d < d_hi; d++) tmax = std::max(tmax, std::fabs(tap(p, d)));
    int dmin = d_hi, dmax = d_lo;
    for (int d = d_lo; d < d_hi; d++) {
        double mx = 0.0;
        for (int p = 0; p < N; p++) mx = std::max(mx, std::fabs(tap(p, d)));
        if (mx > 1e-9 * tmax) { dmin = std::min(dmin, d); dmax = std::max(dmax, d); }
    }
    if (dmax < dmin) { *err = "resampler: empty operator"; return nullptr; }
    ResamplePlan pl{};
    pl.from = from; pl.to = to; pl.hop = hop; pl.N = N; pl.nblk = (N + 159) / 160; pl.dmin = dmin;
    pl.K = (dmax - dmin + 1 + 127) / 128 * 128;   // 4 waves x an even number of 16-deep groups
    const size_t span_floats = (size_t)15 * hop + pl.K;          // one frame row tile a workgroup at the least (launch_resample picks 4 / 2 / 1)
    if (span_floats * 4 + 32 > kResampleLdsBytes) { *err = "resampler: frame span exceeds LDS"; return nullptr; }
    // fragment-major operator per 160-column block: [blk][K/16][10][64 lanes][4]
    std::vector<float> frag((size_t)pl.nblk * pl.K * 160);
    for (int cb = 0; cb < pl.nblk; cb++)
        for (int gq = 0; gq < pl.K / 16; gq++)
            for (int mt = 0; mt < 10; mt++)
                for (int lane = 0; lane < 64; lane++)
                    for (int c = 0; c < 4; c++) {
                        const int k = 16 * gq + 4 * (lane >> 4) + c, p = cb * 160 + 16 * mt + (lane & 15);
                        frag[((((size_t)cb * (pl.K / 16) + gq) * 10 + mt) * 64 + lane) * 4 + c] = (float)tap(p, dmin + k);
                    }
    float *dptr = nullptr;
    if (hipMalloc((void **)&dptr, frag.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(dptr, frag.data(), frag.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        *err = "resampler: device upload failed";
        return nullptr;
    }
    pl.d_op = dptr;
    // the same operator pre-split into f16 hi + lo planes for the split-f16 kernel (32-deep steps)
    std::vector<_Float16> frag16((size_t)pl.nblk * pl.K * 160 * 2);
    float op_max = 0.0f;
    for (float v : frag) op_max = std::max(op_max, std::fabs(v));
    const int op_s = f16_scale_exponent(op_max);   // planes hold G * 2^s; the kernel's store multiplies by 2^-s
    pl.op16_unscale = std::ldexp(1.0f, -op_s);
    for (int cb = 0; cb < 2 * pl.nblk; cb++)   // column blocks of 80 phases (5 tiles) for this kernel
        for (int st = 0; st < pl.K / 32; st++)
            for (int mt = 0; mt < 5; mt++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++) {
                        const int k = 32 * st + 8 * (lane >> 4) + j, p = cb * 80 + 16 * mt + (lane & 15);
                        const float v = std::ldexp((float)tap(p, dmin + k), op_s);
                        const _Float16 hi = (_Float16)v;
                        const size_t base = ((((size_t)cb * (pl.K / 32) + st) * 5 + mt) * 2) * 64 * 8;
                        frag16[base + (size_t)lane * 8 + j] = hi;
                        frag16[base + 64 * 8 + (size_t)lane * 8 + j] = (_Float16)(v - (float)hi);
                    }
    void *dptr16 = nullptr;
    if (hipMalloc(&dptr16, frag16.size() * sizeof(_Float16)) != hipSuccess ||
        hipMemcpy(dptr16, frag16.data(), frag16.size() * sizeof(_Float16), hipMemcpyHostToDevice) != hipSuccess) {
        *err = "resampler: device upload failed";
        return nullptr;
    }
    pl.d_op16 = dptr16;
    return &g_plans.emplace(key, pl).first->second;
}

// ---------------------------------------------------------------------------------------
// grid (frame tiles of 16 FT, column blocks of 160 phases, n_seg), block 256 = 4 waves.
// LDS holds the tile's input span (zero outside [0, src_len)); the K reduction is split across
// the 4 waves, the operator streams from L2 in fragment-major 1-KiB loads, partial sums meet in
// LDS and wave w stores frame tile w (4 consecutive outputs per lane: 16-B stores).
// ---------------------------------------------------------------------------------------
constexpr int RS_MT = 10;

// FT: frame row tiles (of 16) a workgroup owns -- 4 (64 frames) while the tile's input span, (16 FT - 1) hop + K samples, fits
// the LDS; 2 or 1 for the pairs that decimate by a large factor (round 6: 192 / 256 / 384 kHz recorders against 48 / 32 kHz models have
// hop 640-1 920 and were refused).  The K reduction stays split over the four waves; waves 0 .. FT - 1 own the frame tiles.  Every
// output sample is the same sum in the same order whatever FT (the waves' partial sums meet in wave order).
template <int FT>
__global__ __launch_bounds__(256) void resample_kernel(const float *__restrict__ in, size_t in_stride, int src_len,
                                                       float *__restrict__ out, size_t out_stride, int out_len,
                                                       int n_valid, const float *__restrict__ op, int hop, int N,
                                                       int K, int dmin) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int seg = blockIdx.z, cb = blockIdx.y, t0 = blockIdx.x * (16 * FT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const float *xseg = in + (size_t)seg * in_stride;
    const int span = (16 * FT - 1) * hop + K;
    const int g0 = t0 * hop + dmin;
    for (int i0 = tid; i0 < span; i0 += 256 * 8) {   // 8 independent loads in flight per thread
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int gi = g0 + i0 + u * 256;
            v[u] = (i0 + u * 256 < span && gi >= 0 && gi < src_len) ? xseg[gi] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (i0 + u * 256 < span) smem[i0 + u * 256] = v[u];
    }
    __syncthreads();

    f32x4 acc[FT][RS_MT];
#pragma unroll
    for (int f = 0; f < FT; f++)
#pragma unroll
        for (int m = 0; m < RS_MT; m++) acc[f][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int gpw = K / 64, gbeg = wave * gpw;
    const float4 *gA = reinterpret_cast<const float4 *>(op) + (size_t)cb * (K / 16) * RS_MT * 64 + lane;
    // two operator register sets used alternately; scheduling fences keep each load one group ahead
    float4 a0[RS_MT], a1[RS_MT];
#pragma unroll
    for (int m = 0; m < RS_MT; m++) a0[m] = gA[((size_t)gbeg * RS_MT + m) * 64];
    const float *xf = smem + li * hop;
    auto group = [&](int g, const float4 (&a)[RS_MT]) {
        const int jb = g * 16 + 4 * kq;
        float b[4][FT];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int f = 0; f < FT; f++) b[c][f] = xf[f * 16 * hop + jb + c];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int m = 0; m < RS_MT; m++) {
                const float av = c == 0 ? a[m].x : c == 1 ? a[m].y : c == 2 ? a[m].z : a[m].w;
#pragma unroll
                for (int f = 0; f < FT; f++)
                    acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[c][f], acc[f][m], 0, 0, 0);
            }
    };
    for (int gi = 0; gi < gpw; gi += 2) {   // gpw even: K % 128 == 0
        if (gi + 1 < gpw) {   // never issue a prefetch nobody consumes (see mel_kernel)
#pragma unroll
            for (int m = 0; m < RS_MT; m++) a1[m] = gA[((size_t)(gbeg + gi + 1) * RS_MT + m) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        group(gbeg + gi, a0);
        __builtin_amdgcn_sched_barrier(0);
        if (gi + 2 < gpw) {
#pragma unroll
            for (int m = 0; m < RS_MT; m++) a0[m] = gA[((size_t)(gbeg + gi + 2) * RS_MT + m) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (gi + 1 < gpw) group(gbeg + gi + 1, a1);
        __builtin_amdgcn_sched_barrier(0);
    }

    __syncthreads();  // every wave is done reading the span
    float4 *red = reinterpret_cast<float4 *>(smem);
    // red[owner frame tile f][source wave, the owner's own left out][m][lane]
#pragma unroll
    for (int f = 0; f < FT; f++) {
        if (f == wave) continue;
        const int slot = wave - (wave > f ? 1 : 0);
#pragma unroll
        for (int m = 0; m < RS_MT; m++)
            red[((f * 3 + slot) * RS_MT + m) * 64 + lane] = make_float4(acc[f][m][0], acc[f][m][1], acc[f][m][2], acc[f][m][3]);
    }
    __syncthreads();
    if (wave >= FT) return;          // (after the last barrier)
    const int t = t0 + wave * 16 + li;
    float *oseg = out + (size_t)seg * out_stride;
#pragma unroll
    for (int m = 0; m < RS_MT; m++) {
        f32x4 v = acc[0][m];
#pragma unroll
        for (int f = 1; f < FT; f++)
            if (wave == f) v = acc[f][m];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (s == wave) continue;
            const int slot = s - (s > wave ? 1 : 0);
            const float4 q = red[((wave * 3 + slot) * RS_MT + m) * 64 + lane];
            v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
        }
        const int pc = cb * 160 + m * 16 + kq * 4;                // 4 consecutive output phases (the block form's N is any number:
        const long o = (long)t * N + pc;                          //  columns past it are padding, and rows may start unaligned)
        if (pc + 3 < N && (N & 3) == 0 && o + 3 < out_len && o + 3 < n_valid) {
            *reinterpret_cast<float4 *>(oseg + o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (pc + r < N && o + r < out_len) oseg[o + r] = (o + r < n_valid) ? v[r] : 0.0f;  // resize(.., 0.0) pads
        }
    }
}

// The same kernel on the split-f16 MFMA (classifier precision f16x3 / f16): the operator arrives as f16 hi / lo planes,
// the input samples are split in registers (8 consecutive k per lane and 32-deep step), three
// v_mfma_f32_16x16x32_f16 per product (hi*hi + hi*lo + lo*hi, f32 accumulate): the f32 MFMA runs at the vector rate
// (157 TFLOP/s), this one at 2.5 PFLOP/s / 3.  Same staging, K split, reduction and stores as above.
typedef _Float16 rs_f16x8 __attribute__((ext_vector_type(8)));
constexpr int RS16_MT = 5;   // column blocks of 80 phases: 80 accumulator + 80 operator registers, two workgroups per CU
template <int FT>
__global__ __launch_bounds__(256, 2) void resample16_kernel(const float *__restrict__ in, size_t in_stride, int src_len,
                                                         float *__restrict__ out, size_t out_stride, int out_len,
                                                         int n_valid, const rs_f16x8 *__restrict__ op, int hop, int N,
                                                         int K, int dmin, float op_unscale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int seg = blockIdx.z, cb = blockIdx.y, t0 = blockIdx.x * (16 * FT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const float *xseg = in + (size_t)seg * in_stride;
    const int span = (16 * FT - 1) * hop + K;
    const int g0 = t0 * hop + dmin;
    float amax = 0.0f;   // largest |sample| of this workgroup's span
    for (int i0 = tid; i0 < span; i0 += 256 * 8) {   // 8 independent loads in flight per thread
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int gi = g0 + i0 + u * 256;
            v[u] = (i0 + u * 256 < span && gi >= 0 && gi < src_len) ? xseg[gi] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (i0 + u * 256 < span) { smem[i0 + u * 256] = v[u]; amax = fmaxf(amax, fabsf(v[u])); }
    }
    // Block floating point for the f16 operand split: the span is multiplied by the power of two that puts its largest
    // sample in [2^13, 2^14) (quiet recordings -- |x| ~ 1e-4 is ordinary field audio -- would otherwise sit among the f16
    // subnormals and lose their lo halves entirely), and the store multiplies by its inverse.  Exact (powers of two), local
    // to the workgroup, independent of every other segment.  A span of zeros / denormals, or one holding inf, keeps scale 1.
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    float *amax_s = smem + ((span + 3) & ~3);
    if (lane == 0) amax_s[wave] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(amax_s[0], amax_s[1]), fmaxf(amax_s[2], amax_s[3]));
    const int ex = (int)((__float_as_uint(amax) >> 23) & 255u);        // amax = m 2^(ex - 127), m in [1, 2)
    const bool rescale = ex >= 14 && ex <= 253 && ex != 140;           // 2^(140 - ex) and its inverse are normal floats
    const float in_scale = rescale ? __uint_as_float((unsigned)(267 - ex) << 23) : 1.0f;     // 2^(140 - ex)
    const float out_unscale = op_unscale * (rescale ? __uint_as_float((unsigned)(ex - 13) << 23) : 1.0f);   // x 2^(ex - 140)
    if (rescale) {
        for (int i = tid; i < span; i += 256) smem[i] *= in_scale;
    }
    __syncthreads();

    f32x4 acc[FT][RS16_MT];
#pragma unroll
    for (int f = 0; f < FT; f++)
#pragma unroll
        for (int m = 0; m < RS16_MT; m++) acc[f][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int spw = K / 128, sbeg = wave * spw;   // 32-deep steps per wave (K % 128 == 0)
    const rs_f16x8 *gA = op + (size_t)cb * (K / 32) * RS16_MT * 2 * 64 + lane;
    rs_f16x8 a0h[RS16_MT], a0l[RS16_MT], a1h[RS16_MT], a1l[RS16_MT];
    auto load = [&](int st, rs_f16x8 (&ah)[RS16_MT], rs_f16x8 (&al)[RS16_MT]) {
#pragma unroll
        for (int m = 0; m < RS16_MT; m++) {
            ah[m] = gA[(((size_t)st * RS16_MT + m) * 2 + 0) * 64];
            al[m] = gA[(((size_t)st * RS16_MT + m) * 2 + 1) * 64];
        }
    };
    const float *xf = smem + li * hop;
    auto step = [&](int st, const rs_f16x8 (&ah)[RS16_MT], const rs_f16x8 (&al)[RS16_MT]) {
        const int j0 = st * 32 + 8 * kq;
        bh_f16x8 bh[FT], bl[FT];
#pragma unroll
        for (int f = 0; f < FT; f++) {
            float y[8];
#pragma unroll
            for (int jj = 0; jj < 8; jj++) y[jj] = xf[f * 16 * hop + j0 + jj];
            bh_split8(y, bh[f], bl[f]);
        }
        __builtin_amdgcn_sched_barrier(0);   // the split stays out of the MFMA sequence (see mel_kernel)
#pragma unroll
        for (int m = 0; m < RS16_MT; m++)
#pragma unroll
            for (int f = 0; f < FT; f++) {
                acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[f], acc[f][m], 0, 0, 0);
                acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[f], acc[f][m], 0, 0, 0);
                acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[f], acc[f][m], 0, 0, 0);
            }
    };
    // the last one or two steps are peeled: no conditional load inside the loop (see mel_kernel)
    load(sbeg, a0h, a0l);
    {
        int si = 0;
        for (; si + 2 < spw; si += 2) {
            load(sbeg + si + 1, a1h, a1l);
            __builtin_amdgcn_sched_barrier(0);
            step(sbeg + si, a0h, a0l);
            __builtin_amdgcn_sched_barrier(0);
            load(sbeg + si + 2, a0h, a0l);
            __builtin_amdgcn_sched_barrier(0);
            step(sbeg + si + 1, a1h, a1l);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (si + 1 < spw) {
            load(sbeg + si + 1, a1h, a1l);
            __builtin_amdgcn_sched_barrier(0);
            step(sbeg + si, a0h, a0l);
            __builtin_amdgcn_sched_barrier(0);
            step(sbeg + si + 1, a1h, a1l);
        } else {
            step(sbeg + si, a0h, a0l);
        }
    }

    __syncthreads();  // every wave is done reading the span
    float4 *red = reinterpret_cast<float4 *>(smem);
    // red[owner frame tile f][source wave, the owner's own left out][m][lane]
#pragma unroll
    for (int f = 0; f < FT; f++) {
        if (f == wave) continue;
        const int slot = wave - (wave > f ? 1 : 0);
#pragma unroll
        for (int m = 0; m < RS16_MT; m++)
            red[((f * 3 + slot) * RS16_MT + m) * 64 + lane] = make_float4(acc[f][m][0], acc[f][m][1], acc[f][m][2], acc[f][m][3]);
    }
    __syncthreads();
    if (wave >= FT) return;          // (after the last barrier)
    const int t = t0 + wave * 16 + li;
    float *oseg = out + (size_t)seg * out_stride;
#pragma unroll
    for (int m = 0; m < RS16_MT; m++) {
        f32x4 v = acc[0][m];
#pragma unroll
        for (int f = 1; f < FT; f++)
            if (wave == f) v = acc[f][m];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (s == wave) continue;
            const int slot = s - (s > wave ? 1 : 0);
            const float4 q = red[((wave * 3 + slot) * RS16_MT + m) * 64 + lane];
            v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] *= out_unscale;   // the operator planes hold G * 2^s, the span x * 2^(140 - ex)
        const int pc = cb * 80 + m * 16 + kq * 4;                // 4 consecutive output phases (the block form's N is any number:
        const long o = (long)t * N + pc;                          //  columns past it are padding, and rows may start unaligned)
        if (pc + 3 < N && (N & 3) == 0 && o + 3 < out_len && o + 3 < n_valid) {
            *reinterpret_cast<float4 *>(oseg + o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (pc + r < N && o + r < out_len) oseg[o + r] = (o + r < n_valid) ? v[r] : 0.0f;  // resize(.., 0.0) pads
        }
    }
}

void launch_resample(const ResamplePlan &pl, const float *d_in, size_t in_stride, int src_len, float *d_out,
                     size_t out_stride, int out_len, int n_seg, bool split_f16, hipStream_t s) {
    const int n_valid = (int)std::min<size_t>((size_t)out_len, resample_output_len((size_t)src_len, pl.from, pl.to));
    const int frames = (out_len + pl.N - 1) / pl.N;
    // frame row tiles a workgroup: as many of 4 / 2 / 1 as the input span leaves room for (resample_plan refused what not even one fits)
    auto span_bytes_of = [&](int ft) { return ((size_t)(16 * ft - 1) * pl.hop + pl.K) * sizeof(float) + 32; };   // (+ the f16 kernel's 4 wave maxima)
    const int ft = span_bytes_of(4) <= kResampleLdsBytes ? 4 : span_bytes_of(2) <= kResampleLdsBytes ? 2 : 1;
    const size_t span_bytes = span_bytes_of(ft);
    dim3 block(256);
#define BH_RS_LAUNCH(FTV)                                                                                                                  \
    do {                                                                                                                                   \
        if (split_f16) {                                                                                                                   \
            static DeviceOnce attr16_set;                                                                                                  \
            attr16_set.run([] { (void)hipFuncSetAttribute((const void *)resample16_kernel<FTV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); \
            const size_t smem16 = std::max(span_bytes, (size_t)FTV * 3 * RS16_MT * 64 * sizeof(float4));                                   \
            dim3 grid16((frames + 16 * FTV - 1) / (16 * FTV), 2 * pl.nblk, n_seg);                                                          \
            hipLaunchKernelGGL(resample16_kernel<FTV>, grid16, block, smem16, s, d_in, in_stride, src_len, d_out, out_stride, out_len,     \
                               n_valid, (const rs_f16x8 *)pl.d_op16, pl.hop, pl.N, pl.K, pl.dmin, pl.op16_unscale);                        \
        } else {                                                                                                                           \
            static DeviceOnce attr_set;                                                                                                    \
            attr_set.run([] { (void)hipFuncSetAttribute((const void *)resample_kernel<FTV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); \
            const size_t smem = std::max(span_bytes, (size_t)FTV * 3 * RS_MT * 64 * sizeof(float4));                                       \
            dim3 grid((frames + 16 * FTV - 1) / (16 * FTV), pl.nblk, n_seg);                                                                \
            hipLaunchKernelGGL(resample_kernel<FTV>, grid, block, smem, s, d_in, in_stride, src_len, d_out, out_stride, out_len,            \
                               n_valid, pl.d_op, pl.hop, pl.N, pl.K, pl.dmin);                                                             \
        }                                                                                                                                  \
    } while (0)
    if (ft == 4) BH_RS_LAUNCH(4); else if (ft == 2) BH_RS_LAUNCH(2); else BH_RS_LAUNCH(1);
#undef BH_RS_LAUNCH
}

}  // namespace bh

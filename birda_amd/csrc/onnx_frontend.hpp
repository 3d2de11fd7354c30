// onnx_frontend.hpp -- the spectrogram front-end of a classifier's `.onnx` file, READ OFF THE GRAPH inside the library.
//
// birda hands ClassifierBuilder::model_path() the model file and ONNX Runtime executes whatever front-end that graph holds
// (reference src/inference/classifier.rs:269-283; SURVEY.md Appendix B: `mag_scale` is a LEARNED scalar, the band edges, the
// affine after the power law and the mel matrices are the model's, not a family's).  Round 4's native route skipped those nodes and
// asserted a family table; a trained file whose constants differ would have been answered wrongly without a word (VERDICT r4
// missing #2).  This header is the C++ port of birda_amd/frontend_recover.py + birda_amd/onnx_eval.py (which stay as the second
// witness, tests/test_onnx_native.py): nothing matches a spelling.  The nodes between the audio input and the first 2-D
// convolution are RUN by a small float64 evaluator on probe signals and the parameters of the one front-end shape the device
// kernels implement are fitted to the responses:
//
//     x_n = 2 ((x - min x) / (max x - min x + eps) - 0.5)                                   per segment
//     T_b = frames(x_n; L_b, H_b) . G_b,   G_b = diag(hann_L) . cos(2 pi k n / L) . W_b     [L x n_mels]
//     S_b = scale_b (T_b^2)^expo_b + shift_b, mel axis optionally reversed, branches stacked as channels [N, C, n_mels, n_frames]
//
//   1. the spectrogram tensor = the data input of the first Conv with a 2-D kernel; the branch tensors T_b = the inputs of the
//      first squaring nodes (Mul(t, t) / Pow(t, c)) on the way there;
//   2. tail T_b -> S: constants fed AT T_b give scale, shift, exponent (three values fix them, two more check the form); ramps
//      fed at T_b give the axis order and the mel flip;
//   3. eps: the same impulse on a signal of range 2 and of range 0.002;
//   4. H_b: the last frame an impulse reaches bounds it, a shifted probe confirms it; G_b: one impulse per residue class of the
//      frame step, MANY PER PROBE ROW (impulses further apart than the longest possible frame never share a frame, and with the
//      signal's extremes pinned at two samples the normalisation is fixed, so T is affine in everything else): 6 rows instead of
//      the 278 of the Python witness for BirdNET's first branch;
//   5. G_b is factored over the Hann-windowed cosines in closed form (the cosines are orthogonal; the DC row, which the window
//      makes unobservable, is pinned to zero): a residual means the graph's window / transform is not what the kernels fold,
//      and the file is refused with that reason;
//   6. the whole sub-graph is re-run on random audio (loud; quiet with a DC offset) against the closed form above evaluated
//      from the fitted, float32-rounded parameters: <= 2e-5 or the file is refused.
//
// Operator set of the evaluator: element-wise arithmetic, reductions, shape ops, Conv, MatMul / Gemm, STFT,
// BatchNormalization.  Anything else is refused BY OPERATOR NAME (BH_ERR_MODEL) -- never assumed.  Untrusted input: every shape is
// checked, every array is bounded (EVAL_MAX_ELEMS), every product of sizes that sets a loop count is bounded (EVAL_MAX_MACS);
// fuzzed under ASan + UBSan with the other loaders (tests/test_host_sanitizers.py).  Host code only; nothing here runs per
// segment: a BirdNET-sized front-end takes ~1-2 s of the create on the pool's granted cores.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#if defined(__linux__)
#include <sched.h>
#endif

#include "model.hpp"
#include "onnx_graph.hpp"

namespace bh {
namespace onnxf {

using onnxc::Graph;
using onnxc::Node;
using onnxc::Tensor;

struct EvalError : std::runtime_error { using std::runtime_error::runtime_error; };
struct RecoverError : std::runtime_error { using std::runtime_error::runtime_error; };

constexpr size_t EVAL_MAX_ELEMS = 1u << 26;            // 64 M elements (512 MB of float64) per array
constexpr uint64_t EVAL_MAX_MACS = 1ull << 37;         // per operator (the largest legitimate one: 8 rows of a 2048-tap DFT Conv = 1.7e10)
constexpr size_t EVAL_MAX_RANK = 8;
constexpr size_t EVAL_MAX_LIVE = (size_t)3 << 28;      // 768 M elements alive at once (6 GB of float64); a BirdNET front-end peaks at ~40 M
constexpr size_t RECOVER_MAX_NODES = 4096;             // nodes between the audio input and the spectrogram (the published front-ends: a few dozen)

// ---- threads: the cores this process is GRANTED (affinity mask and cgroup quota), at most 16 ---------------------------------
inline unsigned usable_threads_uncached() {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
#if defined(__linux__)
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min<unsigned>(n, (unsigned)c); }
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0}; long long period = 0;
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long long quota = atoll(q);
            if (quota > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
        }
        fclose(f);
    }
#endif
    return std::max(1u, std::min(n, 16u));
}
inline unsigned usable_threads() { static const unsigned n = usable_threads_uncached(); return n; }

template <class F> inline void parallel_for(size_t n, size_t min_per_thread, F &&body) {
    const size_t nt = std::min<size_t>(usable_threads(), std::max<size_t>(1, n / std::max<size_t>(1, min_per_thread)));
    if (nt <= 1) { body(0, n); return; }
    std::vector<std::thread> th;
    std::string err;
    std::atomic<bool> failed{false};
    const size_t step = (n + nt - 1) / nt;
    for (size_t t = 0; t < nt; t++) {
        const size_t lo = t * step, hi = std::min(n, lo + step);
        if (lo >= hi) break;
        th.emplace_back([&, lo, hi]() { try { body(lo, hi); } catch (...) { failed = true; } });
    }
    for (auto &t : th) t.join();
    if (failed) throw EvalError("worker thread failed");
}

// ---- C[M x N] (+)= A[M x K] . B[K x N], row-major float64, blocked for L1 / L2 and register-tiled 4 x 8 -----------------------
// The inner kernel is written so that the compiler vectorises it; the AVX2 + FMA clone is chosen at run time where the host has
// it (the pool's EPYCs do), the baseline clone otherwise.
#define BH_GEMM_BODY                                                                                                   \
    for (size_t j0 = 0; j0 < N; j0 += NB) {                                                                            \
        const size_t jn = std::min(NB, N - j0);                                                                        \
        for (size_t k0 = 0; k0 < K; k0 += KB) {                                                                        \
            const size_t kn = std::min(KB, K - k0);                                                                    \
            size_t i = i_lo;                                                                                           \
            for (; i + 4 <= i_hi; i += 4) {                                                                            \
                double *c0 = C + i * ldc + j0, *c1 = c0 + ldc, *c2 = c1 + ldc, *c3 = c2 + ldc;                         \
                const double *a0 = A + i * lda + k0, *a1 = a0 + lda, *a2 = a1 + lda, *a3 = a2 + lda;                   \
                for (size_t k = 0; k < kn; k++) {                                                                      \
                    const double x0 = a0[k], x1 = a1[k], x2 = a2[k], x3 = a3[k];                                       \
                    const double *b = B + (k0 + k) * ldb + j0;                                                         \
                    for (size_t j = 0; j < jn; j++) { const double bv = b[j]; c0[j] += x0 * bv; c1[j] += x1 * bv; c2[j] += x2 * bv; c3[j] += x3 * bv; } \
                }                                                                                                      \
            }                                                                                                          \
            for (; i < i_hi; i++) {                                                                                    \
                double *c0 = C + i * ldc + j0;                                                                         \
                const double *a0 = A + i * lda + k0;                                                                   \
                for (size_t k = 0; k < kn; k++) { const double x0 = a0[k]; const double *b = B + (k0 + k) * ldb + j0; for (size_t j = 0; j < jn; j++) c0[j] += x0 * b[j]; } \
            }                                                                                                          \
        }                                                                                                              \
    }

inline void gemm_rows_base(size_t i_lo, size_t i_hi, size_t N, size_t K, const double *A, size_t lda, const double *B, size_t ldb, double *C, size_t ldc) {
    constexpr size_t NB = 512, KB = 256;
    BH_GEMM_BODY
}
#if defined(__x86_64__) && (defined(__GNUC__) || defined(__clang__))
__attribute__((target("avx2,fma"))) inline void gemm_rows_avx2(size_t i_lo, size_t i_hi, size_t N, size_t K, const double *A, size_t lda, const double *B, size_t ldb, double *C, size_t ldc) {
    constexpr size_t NB = 512, KB = 256;
    BH_GEMM_BODY
}
inline bool host_has_avx2() { static const bool v = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma"); return v; }
#else
inline void gemm_rows_avx2(size_t i_lo, size_t i_hi, size_t N, size_t K, const double *A, size_t lda, const double *B, size_t ldb, double *C, size_t ldc) { gemm_rows_base(i_lo, i_hi, N, K, A, lda, B, ldb, C, ldc); }
inline bool host_has_avx2() { return false; }
#endif
#undef BH_GEMM_BODY

// C must be initialised by the caller (zeros or a bias)
inline void gemm_acc(size_t M, size_t N, size_t K, const double *A, size_t lda, const double *B, size_t ldb, double *C, size_t ldc) {
    if (!M || !N || !K) return;
    const bool avx = host_has_avx2();
    const uint64_t macs = (uint64_t)M * N * K;
    auto rows = [&](size_t lo, size_t hi) { if (avx) gemm_rows_avx2(lo, hi, N, K, A, lda, B, ldb, C, ldc); else gemm_rows_base(lo, hi, N, K, A, lda, B, ldb, C, ldc); };
    if (macs < (1ull << 22)) { rows(0, M); return; }
    // whole groups of four rows per thread
    const size_t groups = (M + 3) / 4;
    parallel_for(groups, std::max<size_t>(1, (size_t)((1ull << 21) / std::max<uint64_t>(1, (uint64_t)N * K * 4))), [&](size_t lo, size_t hi) { rows(lo * 4, std::min(M, hi * 4)); });
}

// ---- the value type: an n-d array of float64 or int64 --------------------------------------------------------------------
struct Arr {
    std::vector<int64_t> d;
    std::vector<double> v;
    std::vector<int64_t> iv;
    bool is_int = false;
    size_t size() const { return is_int ? iv.size() : v.size(); }
    size_t rank() const { return d.size(); }
    double f(size_t i) const { return is_int ? (double)iv[i] : v[i]; }
};

inline size_t shape_elems(const std::vector<int64_t> &d) {
    if (d.size() > EVAL_MAX_RANK) throw EvalError("tensor of rank " + std::to_string(d.size()));
    uint64_t n = 1;
    for (int64_t x : d) {
        if (x < 0) throw EvalError("negative dimension");
        n *= (uint64_t)x;
        if (n > EVAL_MAX_ELEMS) throw EvalError("tensor larger than the evaluator's bound (" + std::to_string(EVAL_MAX_ELEMS) + " elements)");
    }
    return (size_t)n;
}
inline Arr make_f(std::vector<int64_t> d) { Arr a; a.v.assign(shape_elems(d), 0.0); a.d = std::move(d); return a; }
inline Arr make_i(std::vector<int64_t> d) { Arr a; a.is_int = true; a.iv.assign(shape_elems(d), 0); a.d = std::move(d); return a; }
inline Arr scalar_f(double x) { Arr a; a.v = {x}; return a; }
inline std::string shape_str(const std::vector<int64_t> &d) { std::string s = "["; for (size_t i = 0; i < d.size(); i++) s += (i ? ", " : "") + std::to_string(d[i]); return s + "]"; }
inline std::vector<size_t> strides_of(const std::vector<int64_t> &d) {
    std::vector<size_t> s(d.size(), 1);
    for (size_t i = d.size(); i-- > 1;) s[i - 1] = s[i] * (size_t)d[i];
    return s;
}
inline std::vector<int64_t> ints_of(const Arr &a) {
    std::vector<int64_t> r(a.size());
    for (size_t i = 0; i < r.size(); i++) {
        if (a.is_int) r[i] = a.iv[i];
        else { const double x = std::trunc(a.v[i]); if (!(std::fabs(x) < 9e18)) throw EvalError("index value out of range"); r[i] = (int64_t)x; }
    }
    return r;
}
inline int64_t norm_axis(int64_t ax, size_t rank, const char *what) {
    if (ax < -(int64_t)rank || ax >= (int64_t)rank) throw EvalError(std::string(what) + ": axis " + std::to_string(ax) + " of a rank-" + std::to_string(rank) + " tensor");
    return ax < 0 ? ax + (int64_t)rank : ax;
}
// out[i] = in[map[i]] (map[i] == SIZE_MAX: fill) -- every shape operator is an index map applied to whichever storage the array has
inline Arr take(const Arr &a, std::vector<int64_t> d, const std::vector<size_t> &map, double fill = 0.0) {
    Arr r;
    r.is_int = a.is_int;
    r.d = std::move(d);
    if (a.is_int) { r.iv.resize(map.size()); for (size_t i = 0; i < map.size(); i++) r.iv[i] = map[i] == SIZE_MAX ? (int64_t)fill : a.iv[map[i]]; }
    else { r.v.resize(map.size()); for (size_t i = 0; i < map.size(); i++) r.v[i] = map[i] == SIZE_MAX ? fill : a.v[map[i]]; }
    return r;
}
// index map of a strided view: out dims `d`, per-axis source stride `st` (in elements, may be 0 or negative), source offset `off`
inline std::vector<size_t> view_map(const std::vector<int64_t> &d, const std::vector<int64_t> &st, int64_t off) {
    const size_t n = shape_elems(d);
    std::vector<size_t> map(n);
    if (!n) return map;
    std::vector<int64_t> idx(d.size(), 0);
    int64_t cur = off;
    const size_t r = d.size();
    for (size_t i = 0; i < n; i++) {
        map[i] = (size_t)cur;
        for (size_t ax = r; ax-- > 0;) {
            cur += st[ax];
            if (++idx[ax] < d[ax]) break;
            cur -= st[ax] * d[ax];
            idx[ax] = 0;
        }
    }
    return map;
}
inline std::vector<int64_t> broadcast_shape(const std::vector<int64_t> &a, const std::vector<int64_t> &b, const char *what) {
    const size_t r = std::max(a.size(), b.size());
    std::vector<int64_t> o(r);
    for (size_t i = 0; i < r; i++) {
        const int64_t x = i + a.size() >= r ? a[i + a.size() - r] : 1, y = i + b.size() >= r ? b[i + b.size() - r] : 1;
        if (x != y && x != 1 && y != 1) throw EvalError(std::string(what) + ": shapes " + shape_str(a) + " and " + shape_str(b) + " do not broadcast");
        o[i] = x == 1 ? y : x;
    }
    return o;
}
inline std::vector<int64_t> broadcast_strides(const std::vector<int64_t> &src, const std::vector<int64_t> &out) {
    std::vector<int64_t> st(out.size(), 0);
    const auto ss = strides_of(src);
    for (size_t i = 0; i < src.size(); i++) {
        const size_t o = i + out.size() - src.size();
        st[o] = src[i] == 1 ? 0 : (int64_t)ss[i];
    }
    return st;
}

template <class F> inline Arr binary_f(const Arr &a, const Arr &b, const char *what, F f) {
    Arr r = make_f(broadcast_shape(a.d, b.d, what));
    const size_t n = r.v.size();
    if (!n) return r;
    if (a.d == r.d && b.d == r.d && !a.is_int && !b.is_int) { for (size_t i = 0; i < n; i++) r.v[i] = f(a.v[i], b.v[i]); return r; }
    if (a.d == r.d && b.size() == 1 && !a.is_int) { const double y = b.f(0); for (size_t i = 0; i < n; i++) r.v[i] = f(a.v[i], y); return r; }
    const auto sa = broadcast_strides(a.d, r.d), sb = broadcast_strides(b.d, r.d);
    const size_t rk = r.d.size();
    const int64_t inner = rk ? r.d[rk - 1] : 1, isa = rk ? sa[rk - 1] : 0, isb = rk ? sb[rk - 1] : 0;
    std::vector<int64_t> idx(rk, 0);
    int64_t oa = 0, ob = 0;
    for (size_t i = 0; i < n; i += (size_t)inner) {
        for (int64_t j = 0; j < inner; j++) r.v[i + j] = f(a.f((size_t)(oa + j * isa)), b.f((size_t)(ob + j * isb)));
        for (size_t ax = rk > 0 ? rk - 1 : 0; ax-- > 0;) {
            oa += sa[ax]; ob += sb[ax];
            if (++idx[ax] < r.d[ax]) break;
            oa -= sa[ax] * r.d[ax]; ob -= sb[ax] * r.d[ax];
            idx[ax] = 0;
        }
    }
    return r;
}
template <class F> inline Arr binary_i(const Arr &a, const Arr &b, const char *what, F f) {
    Arr r = make_i(broadcast_shape(a.d, b.d, what));
    const auto ma = view_map(r.d, broadcast_strides(a.d, r.d), 0), mb = view_map(r.d, broadcast_strides(b.d, r.d), 0);
    for (size_t i = 0; i < r.iv.size(); i++) r.iv[i] = f(a.iv[ma[i]], b.iv[mb[i]]);
    return r;
}
template <class F> inline Arr unary_f(const Arr &a, F f) {
    Arr r = make_f(a.d);
    for (size_t i = 0; i < r.v.size(); i++) r.v[i] = f(a.f(i));
    return r;
}
inline Arr to_float(const Arr &a) { if (!a.is_int) return a; Arr r = make_f(a.d); for (size_t i = 0; i < r.v.size(); i++) r.v[i] = (double)a.iv[i]; return r; }

// ---- an initializer as an evaluator value --------------------------------------------------------------------------------
inline Arr arr_of_tensor(const Tensor &t, const std::string &name) {
    Arr a;
    a.d = t.dims;
    const size_t n = shape_elems(a.d);
    if (t.dtype == 1) {
        if (t.has_raw ? t.raw.n != n * 4 : t.fl.size() != n) throw EvalError("constant '" + name + "' without data");     // (never read past the payload)
        a.v.resize(n);
        for (size_t i = 0; i < n; i++) a.v[i] = (double)t.at(i);
    }
    else if (t.dtype == 11) { if (t.dl.size() != n) throw EvalError("constant '" + name + "' without data"); a.v = t.dl; }
    else if (t.dtype == 7 || t.dtype == 6 || t.dtype == 9) { if (t.il.size() != n) throw EvalError("constant '" + name + "' without data"); a.is_int = true; a.iv = t.il; }
    else throw EvalError("constant '" + name + "' of element type " + std::to_string(t.dtype) + " (float32, float64, int64, int32 and bool are read)");
    return a;
}

// in-place iterative radix-2 FFT (n a power of two), separate real / imaginary arrays
inline void fft_pow2(double *re, double *im, size_t n, const std::vector<double> &cs, const std::vector<double> &sn) {
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t half = len / 2, step = n / len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < half; k++) {
                const double wr = cs[k * step], wi = -sn[k * step];
                const double xr = re[i + k + half] * wr - im[i + k + half] * wi, xi = re[i + k + half] * wi + im[i + k + half] * wr;
                re[i + k + half] = re[i + k] - xr; im[i + k + half] = im[i + k] - xi;
                re[i + k] += xr; im[i + k] += xi;
            }
    }
}

// ---- the evaluator ------------------------------------------------------------------------------------------------------------
class Evaluator {
public:
    explicit Evaluator(const Graph &g) : g_(g) {
        for (size_t i = 0; i < g.nodes.size(); i++)
            for (const auto &o : g.nodes[i].out) if (!o.empty()) prod_[o] = i;
    }
    using Env = std::map<std::string, Arr>;

    bool is_const(const std::string &t) const { return g_.init.count(t) != 0; }
    // node indices needed for `targets`, in graph order, not looking behind the tensors in `stop`
    std::vector<size_t> ancestors(const std::vector<std::string> &targets, const std::set<std::string> &stop) const {
        std::set<size_t> need;
        std::vector<std::string> stack(targets);
        while (!stack.empty()) {
            const std::string t = stack.back(); stack.pop_back();
            if (stop.count(t) || is_const(t)) continue;
            auto it = prod_.find(t);
            if (it == prod_.end() || need.count(it->second)) continue;
            need.insert(it->second);
            for (const auto &x : g_.nodes[it->second].in) if (!x.empty()) stack.push_back(x);
        }
        return std::vector<size_t>(need.begin(), need.end());
    }
    bool depends_on(const std::string &tensor, const std::string &source) const {
        if (tensor == source) return true;
        for (size_t i : ancestors({tensor}, {}))
            for (const auto &x : g_.nodes[i].in) if (x == source) return true;
        return false;
    }
    int producer(const std::string &t) const { auto it = prod_.find(t); return it == prod_.end() ? -1 : (int)it->second; }

    // `feeds` may name ANY tensor (graph inputs or intermediate ones); only what lies between them and `targets` is computed
    std::vector<Arr> run(const Env &feeds, const std::vector<std::string> &targets) {
        Env env(feeds);
        std::set<std::string> stop;
        for (const auto &kv : feeds) stop.insert(kv.first);
        const auto order = ancestors(targets, stop);
        // the last reader of every tensor, so that the large intermediates are dropped as soon as they have been consumed
        std::map<std::string, size_t> last;
        for (size_t i : order) for (const auto &x : g_.nodes[i].in) last[x] = i;
        std::set<std::string> keep(targets.begin(), targets.end());
        for (size_t i : order) {
            const Node &n = g_.nodes[i];
            std::vector<const Arr *> in;
            for (const auto &x : n.in) {
                if (x.empty()) { in.push_back(nullptr); continue; }
                auto it = env.find(x);
                if (it == env.end()) {
                    auto ci = g_.init.find(x);
                    if (ci == g_.init.end()) throw EvalError("node '" + (n.name.empty() ? n.op : n.name) + "': input '" + x + "' is neither fed nor computed");
                    it = env.emplace(x, constant(x, ci->second)).first;
                }
                in.push_back(&it->second);
            }
            std::vector<Arr> out = op(n, in);
            for (size_t k = 0; k < n.out.size() && k < out.size(); k++) if (!n.out[k].empty()) env[n.out[k]] = std::move(out[k]);
            for (const auto &x : n.in) if (!x.empty() && !keep.count(x) && !feeds.count(x) && last[x] == i && !is_const(x)) env.erase(x);
            // (a hostile graph may keep many large intermediates alive at once: the evaluation's footprint is bounded as a whole)
            size_t live = 0;
            for (const auto &kv : env) live += kv.second.size();
            if (live > EVAL_MAX_LIVE) throw EvalError("the front-end's intermediates exceed the evaluator's memory bound");
        }
        std::vector<Arr> res;
        for (const auto &t : targets) {
            auto it = env.find(t);
            if (it == env.end()) {
                auto ci = g_.init.find(t);
                if (ci == g_.init.end()) throw EvalError("tensor '" + t + "' is not in the graph");
                res.push_back(constant(t, ci->second));
            } else res.push_back(it->second);
        }
        return res;
    }

private:
    const Graph &g_;
    std::map<std::string, size_t> prod_;
    std::map<std::string, Arr> const_cache_;

    const Arr &constant(const std::string &name, const Tensor &t) {
        auto it = const_cache_.find(name);
        if (it == const_cache_.end()) it = const_cache_.emplace(name, arr_of_tensor(t, name)).first;
        return it->second;
    }
    static const Arr &need(const std::vector<const Arr *> &x, size_t i, const Node &n) {
        if (i >= x.size() || !x[i]) throw EvalError(n.op + " ('" + n.name + "'): input " + std::to_string(i) + " missing");
        return *x[i];
    }
    static const Arr *opt(const std::vector<const Arr *> &x, size_t i) { return i < x.size() ? x[i] : nullptr; }
    static bool both_int(const Arr &a, const Arr &b) { return a.is_int && b.is_int; }

    std::vector<Arr> op(const Node &n, const std::vector<const Arr *> &x) {
        const std::string &o = n.op;
        auto one = [](Arr a) { std::vector<Arr> r; r.push_back(std::move(a)); return r; };
        // element-wise
        if (o == "Identity" || o == "Dropout") return one(need(x, 0, n));
        if (o == "Add") { const Arr &a = need(x, 0, n), &b = need(x, 1, n); return one(both_int(a, b) ? binary_i(a, b, "Add", [](int64_t p, int64_t q) { return (int64_t)((uint64_t)p + (uint64_t)q); }) : binary_f(a, b, "Add", [](double p, double q) { return p + q; })); }
        if (o == "Sub") { const Arr &a = need(x, 0, n), &b = need(x, 1, n); return one(both_int(a, b) ? binary_i(a, b, "Sub", [](int64_t p, int64_t q) { return (int64_t)((uint64_t)p - (uint64_t)q); }) : binary_f(a, b, "Sub", [](double p, double q) { return p - q; })); }
        if (o == "Mul") { const Arr &a = need(x, 0, n), &b = need(x, 1, n); return one(both_int(a, b) ? binary_i(a, b, "Mul", [](int64_t p, int64_t q) { return (int64_t)((uint64_t)p * (uint64_t)q); }) : binary_f(a, b, "Mul", [](double p, double q) { return p * q; })); }
        if (o == "Div") {
            const Arr &a = need(x, 0, n), &b = need(x, 1, n);
            if (both_int(a, b)) return one(binary_i(a, b, "Div", [](int64_t p, int64_t q) { if (q == 0 || (p == INT64_MIN && q == -1)) throw EvalError("Div: integer division by zero"); return p / q; }));
            return one(binary_f(a, b, "Div", [](double p, double q) { return p / q; }));
        }
        if (o == "Pow") return one(binary_f(need(x, 0, n), need(x, 1, n), "Pow", [](double p, double q) { return std::pow(p, q); }));
        if (o == "Min" || o == "Max") {
            Arr r = to_float(need(x, 0, n));
            const bool mn = o == "Min";
            for (size_t k = 1; k < x.size(); k++) r = binary_f(r, need(x, k, n), o.c_str(), [mn](double p, double q) { return mn ? std::fmin(p, q) : std::fmax(p, q); });
            return one(std::move(r));
        }
        if (o == "Neg") return one(unary_f(need(x, 0, n), [](double p) { return -p; }));
        if (o == "Abs") return one(unary_f(need(x, 0, n), [](double p) { return std::fabs(p); }));
        if (o == "Sqrt") return one(unary_f(need(x, 0, n), [](double p) { return std::sqrt(p); }));
        if (o == "Exp") return one(unary_f(need(x, 0, n), [](double p) { return std::exp(p); }));
        if (o == "Log") return one(unary_f(need(x, 0, n), [](double p) { return std::log(p); }));
        if (o == "Cos") return one(unary_f(need(x, 0, n), [](double p) { return std::cos(p); }));
        if (o == "Sin") return one(unary_f(need(x, 0, n), [](double p) { return std::sin(p); }));
        if (o == "Tanh") return one(unary_f(need(x, 0, n), [](double p) { return std::tanh(p); }));
        if (o == "Reciprocal") return one(unary_f(need(x, 0, n), [](double p) { return 1.0 / p; }));
        if (o == "Relu") return one(unary_f(need(x, 0, n), [](double p) { return p > 0.0 ? p : 0.0; }));
        if (o == "Sigmoid") return one(unary_f(need(x, 0, n), [](double p) { return 1.0 / (1.0 + std::exp(-p)); }));
        if (o == "Softplus") return one(unary_f(need(x, 0, n), [](double p) { return p > 0 ? p + std::log1p(std::exp(-p)) : std::log1p(std::exp(p)); }));
        if (o == "Clip") {
            Arr r = to_float(need(x, 0, n));
            const Arr *lo = opt(x, 1), *hi = opt(x, 2);
            const bool has_lo = lo || n.a.count("min"), has_hi = hi || n.a.count("max");
            const double l = lo ? (lo->size() == 1 ? lo->f(0) : throw EvalError("Clip: min must be a scalar")) : (double)n.getf("min", 0.f);
            const double h = hi ? (hi->size() == 1 ? hi->f(0) : throw EvalError("Clip: max must be a scalar")) : (double)n.getf("max", 0.f);
            for (auto &v : r.v) { if (has_lo) v = std::fmax(v, l); if (has_hi) v = std::fmin(v, h); }
            return one(std::move(r));
        }
        if (o == "Cast") {
            const Arr &a = need(x, 0, n);
            const int64_t to = n.geti("to", 1);
            if (to == 1 || to == 11 || to == 10 || to == 16) return one(to_float(a));
            if (to == 7 || to == 6) { Arr r = make_i(a.d); const auto iv = ints_of(a); r.iv = iv; return one(std::move(r)); }
            if (to == 9) { Arr r = make_i(a.d); for (size_t i = 0; i < r.iv.size(); i++) r.iv[i] = a.f(i) != 0.0; return one(std::move(r)); }
            throw EvalError("Cast to data type " + std::to_string(to));
        }
        if (o == "ConstantOfShape") {
            const auto shp = ints_of(need(x, 0, n));
            // (the `value` attribute is a tensor: parse_node keeps only Constant's; the default and by far the usual fill is 0)
            if (n.a.count("value")) throw EvalError("ConstantOfShape with an explicit value tensor");
            return one(make_f(shp));
        }
        // reductions
        if (o == "ReduceMin" || o == "ReduceMax" || o == "ReduceSum" || o == "ReduceMean") return one(reduce(n, x));
        // shape operators
        if (o == "Shape") { const Arr &a = need(x, 0, n); Arr r = make_i({(int64_t)a.d.size()}); r.iv = a.d; return one(std::move(r)); }
        if (o == "Transpose") {
            const Arr &a = need(x, 0, n);
            const size_t rk = a.rank();
            std::vector<int64_t> perm;
            if (const auto *p = n.ints("perm")) perm = *p; else for (size_t i = rk; i-- > 0;) perm.push_back((int64_t)i);
            if (perm.size() != rk) throw EvalError("Transpose: perm of the wrong length");
            const auto ss = strides_of(a.d);
            std::vector<int64_t> d(rk), st(rk);
            std::vector<bool> seen(rk, false);
            for (size_t i = 0; i < rk; i++) {
                const int64_t p = norm_axis(perm[i], rk, "Transpose");
                if (seen[(size_t)p]) throw EvalError("Transpose: perm repeats an axis");
                seen[(size_t)p] = true;
                d[i] = a.d[(size_t)p]; st[i] = (int64_t)ss[(size_t)p];
            }
            return one(take(a, d, view_map(d, st, 0)));
        }
        if (o == "Flatten") {
            const Arr &a = need(x, 0, n);
            const int64_t ax = n.geti("axis", 1) < 0 ? n.geti("axis", 1) + (int64_t)a.rank() : n.geti("axis", 1);
            if (ax < 0 || ax > (int64_t)a.rank()) throw EvalError("Flatten: axis");
            int64_t lead = 1;
            for (int64_t i = 0; i < ax; i++) lead *= a.d[(size_t)i];
            Arr r = a;
            r.d = {lead, lead ? (int64_t)a.size() / lead : 0};
            return one(std::move(r));
        }
        if (o == "Reshape") {
            const Arr &a = need(x, 0, n);
            std::vector<int64_t> shp;
            if (const Arr *s = opt(x, 1)) shp = ints_of(*s); else if (const auto *p = n.ints("shape")) shp = *p; else throw EvalError("Reshape without a shape");
            const bool allowzero = n.geti("allowzero", 0) != 0;
            int64_t known = 1; int infer = -1;
            for (size_t i = 0; i < shp.size(); i++) {
                if (shp[i] == 0 && !allowzero) { if (i >= a.rank()) throw EvalError("Reshape: 0 beyond the input's rank"); shp[i] = a.d[i]; }
                if (shp[i] == -1) { if (infer >= 0) throw EvalError("Reshape: two -1 dimensions"); infer = (int)i; }
                else { if (shp[i] < 0 || (shp[i] > 0 && known > (int64_t)EVAL_MAX_ELEMS / shp[i])) throw EvalError("Reshape: bad shape"); known *= shp[i]; }
            }
            if (infer >= 0) { if (known == 0 || (int64_t)a.size() % known) throw EvalError("Reshape: size mismatch"); shp[(size_t)infer] = (int64_t)a.size() / known; }
            if (shape_elems(shp) != a.size()) throw EvalError("Reshape: " + shape_str(a.d) + " -> " + shape_str(shp));
            Arr r = a; r.d = shp;
            return one(std::move(r));
        }
        if (o == "Squeeze") {
            const Arr &a = need(x, 0, n);
            std::vector<int64_t> axes; bool given = false;
            if (const Arr *s = opt(x, 1)) { axes = ints_of(*s); given = true; } else if (const auto *p = n.ints("axes")) { axes = *p; given = true; }
            std::set<int64_t> drop;
            for (int64_t ax : axes) { const int64_t q = norm_axis(ax, a.rank(), "Squeeze"); if (a.d[(size_t)q] != 1) throw EvalError("Squeeze of a dimension that is not 1"); drop.insert(q); }
            Arr r = a; r.d.clear();
            for (size_t i = 0; i < a.rank(); i++) if (given ? !drop.count((int64_t)i) : a.d[i] != 1) r.d.push_back(a.d[i]);
            return one(std::move(r));
        }
        if (o == "Unsqueeze") {
            const Arr &a = need(x, 0, n);
            std::vector<int64_t> axes;
            if (const Arr *s = opt(x, 1)) axes = ints_of(*s); else if (const auto *p = n.ints("axes")) axes = *p; else throw EvalError("Unsqueeze without axes");
            const size_t rk = a.rank() + axes.size();
            if (rk > EVAL_MAX_RANK) throw EvalError("Unsqueeze: rank");
            std::set<int64_t> at;
            for (int64_t ax : axes) if (!at.insert(norm_axis(ax, rk, "Unsqueeze")).second) throw EvalError("Unsqueeze: repeated axis");
            Arr r = a; r.d.clear();
            size_t src = 0;
            for (size_t i = 0; i < rk; i++) r.d.push_back(at.count((int64_t)i) ? 1 : a.d[src++]);
            return one(std::move(r));
        }
        if (o == "Concat") {
            if (x.empty()) throw EvalError("Concat without inputs");
            const Arr &a0 = need(x, 0, n);
            const int64_t ax = norm_axis(n.geti("axis", 0), a0.rank(), "Concat");
            std::vector<int64_t> d = a0.d;
            d[(size_t)ax] = 0;
            bool is_int = true;
            for (size_t k = 0; k < x.size(); k++) {
                const Arr &a = need(x, k, n);
                if (a.rank() != a0.rank()) throw EvalError("Concat: ranks differ");
                for (size_t i = 0; i < a.rank(); i++) if ((int64_t)i != ax && a.d[i] != a0.d[i]) throw EvalError("Concat: shapes differ off the axis");
                d[(size_t)ax] += a.d[(size_t)ax];
                is_int = is_int && a.is_int;
            }
            Arr r = is_int ? make_i(d) : make_f(d);
            size_t outer = 1, inner = 1;
            for (int64_t i = 0; i < ax; i++) outer *= (size_t)d[(size_t)i];
            for (size_t i = (size_t)ax + 1; i < d.size(); i++) inner *= (size_t)d[i];
            size_t off = 0;
            for (size_t k = 0; k < x.size(); k++) {
                const Arr &a = *x[k];
                const size_t w = (size_t)a.d[(size_t)ax] * inner, W = (size_t)d[(size_t)ax] * inner;
                for (size_t oi = 0; oi < outer; oi++)
                    for (size_t j = 0; j < w; j++) { if (is_int) r.iv[oi * W + off + j] = a.iv[oi * w + j]; else r.v[oi * W + off + j] = a.f(oi * w + j); }
                off += w;
            }
            return one(std::move(r));
        }
        if (o == "Expand") {
            const Arr &a = need(x, 0, n);
            const auto d = broadcast_shape(a.d, ints_of(need(x, 1, n)), "Expand");
            return one(take(a, d, view_map(d, broadcast_strides(a.d, d), 0)));
        }
        if (o == "Tile") {
            const Arr &a = need(x, 0, n);
            const auto reps = ints_of(need(x, 1, n));
            if (reps.size() != a.rank()) throw EvalError("Tile: repeats of the wrong length");
            std::vector<int64_t> d(a.rank());
            for (size_t i = 0; i < a.rank(); i++) { if (reps[i] < 0 || (a.d[i] > 0 && reps[i] > (int64_t)EVAL_MAX_ELEMS / a.d[i])) throw EvalError("Tile: repeats"); d[i] = a.d[i] * reps[i]; }
            const size_t nel = shape_elems(d);
            const auto ss = strides_of(a.d);
            std::vector<size_t> map(nel);
            std::vector<int64_t> idx(a.rank(), 0);
            for (size_t i = 0; i < nel; i++) {
                size_t src = 0;
                for (size_t q = 0; q < a.rank(); q++) src += (size_t)(idx[q] % a.d[q]) * ss[q];
                map[i] = src;
                for (size_t q = a.rank(); q-- > 0;) { if (++idx[q] < d[q]) break; idx[q] = 0; }
            }
            return one(take(a, d, map));
        }
        if (o == "Range") {
            const Arr &s = need(x, 0, n), &l = need(x, 1, n), &dl = need(x, 2, n);
            if (s.size() != 1 || l.size() != 1 || dl.size() != 1) throw EvalError("Range: scalars expected");
            const double st = s.f(0), lim = l.f(0), de = dl.f(0);
            if (de == 0.0 || !std::isfinite(st) || !std::isfinite(lim) || !std::isfinite(de)) throw EvalError("Range: bad arguments");
            const double cnt = std::ceil((lim - st) / de);
            if (cnt > (double)EVAL_MAX_ELEMS) throw EvalError("Range: too long");
            const int64_t m = cnt > 0 ? (int64_t)cnt : 0;
            const bool ii = s.is_int && l.is_int && dl.is_int;
            Arr r = ii ? make_i({m}) : make_f({m});
            for (int64_t i = 0; i < m; i++) { if (ii) r.iv[(size_t)i] = s.iv[0] + i * dl.iv[0]; else r.v[(size_t)i] = st + (double)i * de; }
            return one(std::move(r));
        }
        if (o == "Gather") {
            const Arr &a = need(x, 0, n), &ind = need(x, 1, n);
            const int64_t ax = norm_axis(n.geti("axis", 0), a.rank(), "Gather");
            const auto idx = ints_of(ind);
            std::vector<int64_t> d(a.d.begin(), a.d.begin() + ax);
            d.insert(d.end(), ind.d.begin(), ind.d.end());
            d.insert(d.end(), a.d.begin() + ax + 1, a.d.end());
            const size_t nel = shape_elems(d);
            size_t outer = 1, inner = 1;
            for (int64_t i = 0; i < ax; i++) outer *= (size_t)a.d[(size_t)i];
            for (size_t i = (size_t)ax + 1; i < a.rank(); i++) inner *= (size_t)a.d[i];
            const int64_t dim = a.d[(size_t)ax];
            std::vector<size_t> map(nel);
            size_t w = 0;
            for (size_t oi = 0; oi < outer; oi++)
                for (size_t k = 0; k < idx.size(); k++) {
                    int64_t q = idx[k];
                    if (q < -dim || q >= dim) throw EvalError("Gather: index out of range");
                    if (q < 0) q += dim;
                    for (size_t j = 0; j < inner; j++) map[w++] = (oi * (size_t)dim + (size_t)q) * inner + j;
                }
            return one(take(a, d, map));
        }
        if (o == "Slice") {
            const Arr &a = need(x, 0, n);
            std::vector<int64_t> starts, ends, axes, steps;
            if (x.size() > 1) {
                starts = ints_of(need(x, 1, n)); ends = ints_of(need(x, 2, n));
                if (const Arr *s = opt(x, 3)) axes = ints_of(*s);
                if (const Arr *s = opt(x, 4)) steps = ints_of(*s);
            } else {
                const auto *s = n.ints("starts"), *e = n.ints("ends");
                if (!s || !e) throw EvalError("Slice without starts / ends");
                starts = *s; ends = *e;
                if (const auto *p = n.ints("axes")) axes = *p;
            }
            if (axes.empty()) for (size_t i = 0; i < starts.size(); i++) axes.push_back((int64_t)i);
            if (steps.empty()) steps.assign(starts.size(), 1);
            if (ends.size() != starts.size() || axes.size() != starts.size() || steps.size() != starts.size()) throw EvalError("Slice: argument lengths differ");
            std::vector<int64_t> d = a.d, st(a.rank());
            const auto ss = strides_of(a.d);
            for (size_t i = 0; i < a.rank(); i++) st[i] = (int64_t)ss[i];
            int64_t off = 0;
            std::vector<char> seen_ax(a.rank(), 0);
            for (size_t k = 0; k < starts.size(); k++) {
                const size_t ax = (size_t)norm_axis(axes[k], a.rank(), "Slice");
                // (ONNX forbids a repeated axis; the clamps below are against the ORIGINAL extent, so a second slice of one axis would
                //  place its offset past the first slice's end -- ADVICE r5: x[10], starts [5, 9], axes [0, 0] read x[14])
                if (seen_ax[ax]) throw EvalError("Slice: axis " + std::to_string(ax) + " given twice");
                seen_ax[ax] = 1;
                const int64_t dim = a.d[ax], step = steps[k];
                if (step == 0) throw EvalError("Slice: step 0");
                int64_t s = starts[k], e = ends[k];
                // ONNX: negative values count from the end, then clamp to [0, dim] (positive step) or [-1, dim - 1] (negative step)
                if (s < 0) s = s < -dim ? -dim - 1 : s + dim;
                if (e < 0) e = e < -dim ? -dim - 1 : e + dim;
                int64_t cnt;
                if (step > 0) { s = std::min(std::max<int64_t>(s, 0), dim); e = std::min(std::max<int64_t>(e, 0), dim); cnt = e > s ? (e - s + step - 1) / step : 0; }
                else { s = std::min(std::max<int64_t>(s, -1), dim - 1); e = std::min(std::max<int64_t>(e, -1), dim - 1); cnt = s > e ? (s - e + (-step) - 1) / (-step) : 0; }
                off += s * st[ax] * (cnt > 0 ? 1 : 0);
                st[ax] *= step;
                d[ax] = cnt;
            }
            {   // the view's first and last element lie inside the array, whatever the arguments were (every strided view is held to this)
                int64_t lo = off, hi = off;
                bool empty = false;
                for (size_t i = 0; i < a.rank(); i++) {
                    if (d[i] == 0) empty = true;
                    else if (st[i] > 0) hi += (d[i] - 1) * st[i];
                    else lo += (d[i] - 1) * st[i];
                }
                if (!empty && (lo < 0 || hi >= (int64_t)a.size())) throw EvalError("Slice: view outside its array");
            }
            return one(take(a, d, view_map(d, st, off)));
        }
        if (o == "Pad") {
            const Arr &a = need(x, 0, n);
            if (n.gets("mode", "constant") != "constant") throw EvalError("Pad mode " + n.gets("mode", "constant"));
            std::vector<int64_t> pads;
            if (const Arr *p = opt(x, 1)) pads = ints_of(*p); else if (const auto *q = n.ints("pads")) pads = *q; else throw EvalError("Pad without pads");
            if (opt(x, 3)) throw EvalError("Pad with an axes input");
            const double val = opt(x, 2) ? (opt(x, 2)->size() == 1 ? opt(x, 2)->f(0) : throw EvalError("Pad: scalar value expected")) : (double)n.getf("value", 0.f);
            const size_t rk = a.rank();
            if (pads.size() != 2 * rk) throw EvalError("Pad: pads of the wrong length");
            std::vector<int64_t> d(rk);
            for (size_t i = 0; i < rk; i++) {
                if (pads[i] < 0 || pads[rk + i] < 0 || pads[i] > (int64_t)EVAL_MAX_ELEMS || pads[rk + i] > (int64_t)EVAL_MAX_ELEMS) throw EvalError("Pad: negative or oversized pads");
                d[i] = a.d[i] + pads[i] + pads[rk + i];
            }
            const size_t nel = shape_elems(d);
            const auto ss = strides_of(a.d);
            std::vector<size_t> map(nel);
            std::vector<int64_t> idx(rk, 0);
            for (size_t i = 0; i < nel; i++) {
                size_t src = 0; bool inside = true;
                for (size_t q = 0; q < rk; q++) { const int64_t p = idx[q] - pads[q]; if (p < 0 || p >= a.d[q]) { inside = false; break; } src += (size_t)p * ss[q]; }
                map[i] = inside ? src : SIZE_MAX;
                for (size_t q = rk; q-- > 0;) { if (++idx[q] < d[q]) break; idx[q] = 0; }
            }
            return one(take(a, d, map, val));
        }
        // linear algebra
        if (o == "MatMul") return one(matmul(to_float(need(x, 0, n)), to_float(need(x, 1, n))));
        if (o == "Gemm") {
            Arr A = to_float(need(x, 0, n)), B = to_float(need(x, 1, n));
            if (A.rank() != 2 || B.rank() != 2) throw EvalError("Gemm: 2-d operands expected");
            auto tr = [](const Arr &m) { Arr t = make_f({m.d[1], m.d[0]}); for (int64_t i = 0; i < m.d[0]; i++) for (int64_t j = 0; j < m.d[1]; j++) t.v[(size_t)(j * m.d[0] + i)] = m.v[(size_t)(i * m.d[1] + j)]; return t; };
            if (n.geti("transA", 0)) A = tr(A);
            if (n.geti("transB", 0)) B = tr(B);
            Arr r = matmul(A, B);
            const double alpha = (double)n.getf("alpha", 1.0f), beta = (double)n.getf("beta", 1.0f);
            if (alpha != 1.0) for (auto &v : r.v) v *= alpha;
            if (const Arr *c = opt(x, 2)) r = binary_f(r, *c, "Gemm", [beta](double p, double q) { return p + beta * q; });
            return one(std::move(r));
        }
        if (o == "Conv") return one(conv(n, to_float(need(x, 0, n)), to_float(need(x, 1, n)), opt(x, 2)));
        if (o == "STFT") return one(stft(n, x));
        if (o == "BatchNormalization") {
            const Arr &a = need(x, 0, n);
            if (a.rank() < 2 || x.size() < 5) throw EvalError("BatchNormalization: operands");
            const int64_t c = a.d[1];
            for (size_t k = 1; k < 5; k++) if ((int64_t)need(x, k, n).size() != c) throw EvalError("BatchNormalization: parameters must have one value per channel");
            const double eps = (double)n.getf("epsilon", 1e-5f);
            Arr r = to_float(a);
            size_t inner = 1;
            for (size_t i = 2; i < a.rank(); i++) inner *= (size_t)a.d[i];
            for (size_t i = 0; i < r.v.size(); i++) {
                const size_t ch = (i / inner) % (size_t)c;
                r.v[i] = (r.v[i] - x[3]->f(ch)) / std::sqrt(x[4]->f(ch) + eps) * x[1]->f(ch) + x[2]->f(ch);
            }
            return one(std::move(r));
        }
        throw EvalError("operator " + o + " (node '" + (n.name.empty() ? (n.out.empty() ? std::string() : n.out[0]) : n.name) + "') is outside the front-end operator set");
    }

    Arr reduce(const Node &n, const std::vector<const Arr *> &x) {
        const Arr a = to_float(need(x, 0, n));
        std::vector<int64_t> axes; bool given = false;
        if (const Arr *s = opt(x, 1)) { axes = ints_of(*s); given = true; } else if (const auto *p = n.ints("axes")) { axes = *p; given = true; }
        const bool keep = n.geti("keepdims", 1) != 0;
        if (axes.empty()) {
            if (given && n.geti("noop_with_empty_axes", 0)) return a;
            for (size_t i = 0; i < a.rank(); i++) axes.push_back((int64_t)i);
        }
        std::vector<bool> red(a.rank(), false);
        for (int64_t ax : axes) red[(size_t)norm_axis(ax, a.rank(), n.op.c_str())] = true;
        std::vector<int64_t> dk(a.rank()), dout;
        for (size_t i = 0; i < a.rank(); i++) { dk[i] = red[i] ? 1 : a.d[i]; if (!red[i] || keep) dout.push_back(dk[i]); }
        Arr r = make_f(dk);
        const int kind = n.op == "ReduceMin" ? 0 : n.op == "ReduceMax" ? 1 : 2;
        if (a.size() == 0) { if (kind < 2 && r.v.size()) throw EvalError(n.op + " of an empty tensor"); r.d = dout; return r; }
        const double init = kind == 0 ? INFINITY : kind == 1 ? -INFINITY : 0.0;
        std::fill(r.v.begin(), r.v.end(), init);
        const auto so = strides_of(dk);
        std::vector<int64_t> idx(a.rank(), 0);
        size_t cnt = 1;
        for (size_t i = 0; i < a.rank(); i++) if (red[i]) cnt *= (size_t)a.d[i];
        for (size_t i = 0; i < a.v.size(); i++) {
            size_t oi = 0;
            for (size_t q = 0; q < a.rank(); q++) if (!red[q]) oi += (size_t)idx[q] * so[q];
            double &t = r.v[oi];
            const double v = a.v[i];
            if (kind == 0) t = v < t || std::isnan(v) ? v : t; else if (kind == 1) t = v > t || std::isnan(v) ? v : t; else t += v;
            for (size_t q = a.rank(); q-- > 0;) { if (++idx[q] < a.d[q]) break; idx[q] = 0; }
        }
        if (n.op == "ReduceMean") for (auto &v : r.v) v /= (double)cnt;
        r.d = dout;
        return r;
    }

    static Arr matmul(const Arr &a, const Arr &b) {
        if (a.rank() < 1 || b.rank() < 1) throw EvalError("MatMul of a scalar");
        // numpy.matmul semantics: 1-d operands get a unit axis that is dropped again; batch axes broadcast
        std::vector<int64_t> ad = a.d, bd = b.d;
        const bool a1 = ad.size() == 1, b1 = bd.size() == 1;
        if (a1) ad.insert(ad.begin(), 1);
        if (b1) bd.push_back(1);
        const int64_t M = ad[ad.size() - 2], K = ad.back(), N = bd.back();
        if (bd[bd.size() - 2] != K) throw EvalError("MatMul: " + shape_str(a.d) + " x " + shape_str(b.d));
        const std::vector<int64_t> ab(ad.begin(), ad.end() - 2), bb(bd.begin(), bd.end() - 2);
        const auto batch = broadcast_shape(ab, bb, "MatMul");
        std::vector<int64_t> od = batch;
        od.push_back(M); od.push_back(N);
        Arr r = make_f(od);
        const size_t nb = shape_elems(batch);
        if ((uint64_t)nb * (uint64_t)M * (uint64_t)N * (uint64_t)std::max<int64_t>(K, 1) > EVAL_MAX_MACS) throw EvalError("MatMul beyond the evaluator's work bound");
        if (bb.empty() || shape_elems(bb) == 1) {
            // one weight matrix for every batch entry: a single [batch . M, K] x [K, N] product
            gemm_acc(nb * (size_t)M, (size_t)N, (size_t)K, a.v.data(), (size_t)K, b.v.data(), (size_t)N, r.v.data(), (size_t)N);
        } else {
            const auto ma = view_map(batch, broadcast_strides(ab, batch), 0), mb = view_map(batch, broadcast_strides(bb, batch), 0);
            for (size_t i = 0; i < nb; i++)
                gemm_acc((size_t)M, (size_t)N, (size_t)K, a.v.data() + ma[i] * (size_t)(M * K), (size_t)K, b.v.data() + mb[i] * (size_t)(K * N), (size_t)N, r.v.data() + i * (size_t)(M * N), (size_t)N);
        }
        if (a1) r.d.erase(r.d.end() - 2);
        if (b1) r.d.pop_back();
        return r;
    }

    // 1-d and 2-d convolution, group 1 or depthwise, dilation 1: im2col + GEMM per batch entry
    static Arr conv(const Node &n, const Arr &xin, const Arr &w, const Arr *bias) {
        const size_t nd = xin.rank() >= 2 ? xin.rank() - 2 : 0;
        if ((nd != 1 && nd != 2) || w.rank() != nd + 2) throw EvalError("Conv: 1-d and 2-d convolutions are evaluated, input " + shape_str(xin.d) + ", weights " + shape_str(w.d));
        if (const auto *dl = n.ints("dilations")) for (int64_t d : *dl) if (d != 1) throw EvalError("Conv: dilation");
        const int64_t group = n.geti("group", 1);
        const int64_t N = xin.d[0], C = xin.d[1], M = w.d[0], Cg = w.d[1];
        const int64_t ih = nd == 2 ? xin.d[2] : 1, iw = xin.d[nd + 1], kh = nd == 2 ? w.d[2] : 1, kw = w.d[nd + 1];
        int64_t sh = 1, sw = 1, pt = 0, pl = 0, pb = 0, pr = 0;
        if (const auto *st = n.ints("strides")) { if (st->size() != nd) throw EvalError("Conv: strides"); sh = nd == 2 ? (*st)[0] : 1; sw = (*st)[nd - 1]; }
        if (sh < 1 || sw < 1 || kh < 1 || kw < 1 || C < 1 || M < 1) throw EvalError("Conv: bad strides or shapes");
        const std::string autop = n.gets("auto_pad", "NOTSET");
        if (autop == "SAME_UPPER" || autop == "SAME_LOWER") {
            const int64_t oh0 = (ih + sh - 1) / sh, ow0 = (iw + sw - 1) / sw;
            const int64_t th = std::max<int64_t>((oh0 - 1) * sh + kh - ih, 0), tw = std::max<int64_t>((ow0 - 1) * sw + kw - iw, 0);
            pt = autop == "SAME_UPPER" ? th / 2 : th - th / 2; pb = th - pt;
            pl = autop == "SAME_UPPER" ? tw / 2 : tw - tw / 2; pr = tw - pl;
        } else if (autop == "NOTSET" || autop == "VALID") {
            if (const auto *pd = n.ints("pads")) {
                if (pd->size() != 2 * nd) throw EvalError("Conv: pads");
                if (nd == 2) { pt = (*pd)[0]; pl = (*pd)[1]; pb = (*pd)[2]; pr = (*pd)[3]; } else { pl = (*pd)[0]; pr = (*pd)[1]; }
            }
        } else throw EvalError("Conv: auto_pad " + autop);
        if (pt < 0 || pl < 0 || pb < 0 || pr < 0 || pt > (1 << 20) || pl > (1 << 20) || pb > (1 << 20) || pr > (1 << 20)) throw EvalError("Conv: pads");
        if (ih + pt + pb < kh || iw + pl + pr < kw) throw EvalError("Conv: kernel larger than the padded input");
        const int64_t oh = (ih + pt + pb - kh) / sh + 1, ow = (iw + pl + pr - kw) / sw + 1;
        const bool depthwise = group == C && Cg == 1 && group > 1;
        if (!depthwise && (group != 1 || Cg != C)) throw EvalError("Conv: grouped (non-depthwise) convolution or channel mismatch");
        if (depthwise && M % group) throw EvalError("Conv: depthwise multiplier");
        if (bias && (int64_t)bias->size() != M) throw EvalError("Conv: bias width");
        std::vector<int64_t> od{N, M};
        if (nd == 2) od.push_back(oh);
        od.push_back(ow);
        Arr r = make_f(od);
        const size_t P = (size_t)(oh * ow), KK = (size_t)(kh * kw);
        const uint64_t macs = (uint64_t)N * (uint64_t)M * P * KK * (uint64_t)(depthwise ? 1 : C);
        if (macs > EVAL_MAX_MACS) throw EvalError("Conv beyond the evaluator's work bound");
        if (bias) for (int64_t b = 0; b < N; b++) for (int64_t m = 0; m < M; m++) std::fill_n(r.v.begin() + (size_t)((b * M + m)) * P, P, bias->f((size_t)m));
        auto sample = [&](int64_t b, int64_t c, int64_t y, int64_t xx) -> double {
            return (y < 0 || y >= ih || xx < 0 || xx >= iw) ? 0.0 : xin.v[(size_t)(((b * C + c) * ih + y) * iw + xx)];
        };
        if (depthwise) {
            const int64_t mult = M / group;
            parallel_for((size_t)(N * M), 4, [&](size_t lo, size_t hi) {
                for (size_t q = lo; q < hi; q++) {
                    const int64_t b = (int64_t)q / M, m = (int64_t)q % M, c = m / mult;
                    double *out = r.v.data() + q * P;
                    for (int64_t oy = 0; oy < oh; oy++)
                        for (int64_t ox = 0; ox < ow; ox++) {
                            double acc = 0.0;
                            for (int64_t ky = 0; ky < kh; ky++)
                                for (int64_t kx = 0; kx < kw; kx++) acc += sample(b, c, oy * sh - pt + ky, ox * sw - pl + kx) * w.v[(size_t)((m * kh + ky) * kw + kx)];
                            out[oy * ow + ox] += acc;
                        }
                }
            });
            return r;
        }
        const size_t Kc = (size_t)C * KK;
        if (Kc * P > EVAL_MAX_ELEMS) throw EvalError("Conv: im2col buffer beyond the evaluator's bound");
        std::vector<double> cols(Kc * P);
        for (int64_t b = 0; b < N; b++) {
            parallel_for(Kc, 64, [&](size_t lo, size_t hi) {
                for (size_t kq = lo; kq < hi; kq++) {
                    const int64_t c = (int64_t)(kq / KK), ky = (int64_t)(kq % KK) / kw, kx = (int64_t)(kq % KK) % kw;
                    double *row = cols.data() + kq * P;
                    for (int64_t oy = 0; oy < oh; oy++)
                        for (int64_t ox = 0; ox < ow; ox++) row[oy * ow + ox] = sample(b, c, oy * sh - pt + ky, ox * sw - pl + kx);
                }
            });
            gemm_acc((size_t)M, P, Kc, w.v.data(), Kc, cols.data(), P, r.v.data() + (size_t)b * (size_t)M * P, P);
        }
        return r;
    }

    // ONNX STFT (opset 17): signal [N, S] or [N, S, 1] (real), frame_step, window (optional), frame_length (optional) -> [N, frames, bins, 2]
    static Arr stft(const Node &n, const std::vector<const Arr *> &x) {
        const Arr sig = to_float(need(x, 0, n));
        if (sig.rank() == 3 ? sig.d[2] != 1 : sig.rank() != 2) throw EvalError("STFT: a real signal [N, S] or [N, S, 1] is expected");
        const Arr &stp = need(x, 1, n);
        if (stp.size() != 1) throw EvalError("STFT: frame_step");
        const int64_t step = ints_of(stp)[0];
        const Arr *win = opt(x, 2), *len = opt(x, 3);
        if (!win && !len) throw EvalError("STFT without window and frame_length");
        const int64_t L = len ? (len->size() == 1 ? ints_of(*len)[0] : throw EvalError("STFT: frame_length")) : (int64_t)win->size();
        if (win && (int64_t)win->size() != L) throw EvalError("STFT: window length differs from frame_length");
        const int64_t N = sig.d[0], S = sig.d[1];
        if (step < 1 || L < 1 || L > S || L > (1 << 20)) throw EvalError("STFT: frame geometry");
        const int64_t frames = (S - L) / step + 1, bins = n.geti("onesided", 1) ? L / 2 + 1 : L;
        Arr r = make_f({N, frames, bins, 2});
        const bool pow2 = (L & (L - 1)) == 0 && L >= 2;
        if (!pow2 && (uint64_t)N * frames * L * bins > EVAL_MAX_MACS) throw EvalError("STFT beyond the evaluator's work bound");
        std::vector<double> cs((size_t)L), sn((size_t)L);
        for (int64_t k = 0; k < L; k++) { cs[(size_t)k] = std::cos(2.0 * M_PI * (double)k / (double)L); sn[(size_t)k] = std::sin(2.0 * M_PI * (double)k / (double)L); }
        parallel_for((size_t)(N * frames), 8, [&](size_t lo, size_t hi) {
            std::vector<double> re((size_t)L), im((size_t)L);
            for (size_t q = lo; q < hi; q++) {
                const int64_t b = (int64_t)q / frames, t = (int64_t)q % frames;
                const double *s = sig.v.data() + (size_t)(b * S + t * step);
                for (int64_t i = 0; i < L; i++) { re[(size_t)i] = s[i] * (win ? win->f((size_t)i) : 1.0); im[(size_t)i] = 0.0; }
                double *out = r.v.data() + q * (size_t)bins * 2;
                if (pow2) {
                    fft_pow2(re.data(), im.data(), (size_t)L, cs, sn);
                    for (int64_t k = 0; k < bins; k++) { out[2 * k] = re[(size_t)k]; out[2 * k + 1] = im[(size_t)k]; }
                } else {
                    for (int64_t k = 0; k < bins; k++) {
                        double ar = 0.0, ai = 0.0;
                        for (int64_t i = 0; i < L; i++) { const size_t ph = (size_t)((i * k) % L); ar += re[(size_t)i] * cs[ph]; ai -= re[(size_t)i] * sn[ph]; }
                        out[2 * k] = ar; out[2 * k + 1] = ai;
                    }
                }
            }
        });
        return r;
    }
};

// ---- what the recovery returns -------------------------------------------------------------------------------------------
struct RecoveredBranch {
    uint32_t L = 0, H = 0, n_mels = 0, n_frames = 0, flip = 0;
    double expo = 0, scale = 0, shift = 0, residual = 0;
    float fmin = 0, fmax = 0;
    std::vector<float> mel_w;      // [L / 2 + 1][n_mels], DC row zero
};
struct Recovered {
    std::string spectrogram;       // the tensor the conv stack starts from
    uint32_t sample_count = 0, spec_h = 0, spec_w = 0;
    double eps = 0, verify_err = 0;
    std::vector<RecoveredBranch> branches;    // in channel order
};

// A[n][k] = hann_periodic[n] cos(2 pi k n / L) applied to W [L/2+1][n_mels] -> G [L][n_mels]
inline void hann_cos_apply(uint32_t L, uint32_t n_mels, const std::vector<double> &W, std::vector<double> &G) {
    const uint32_t bins = L / 2 + 1;
    std::vector<double> A((size_t)L * bins);
    for (uint32_t nn = 0; nn < L; nn++) {
        const double w = 0.5 - 0.5 * std::cos(2.0 * M_PI * (double)nn / (double)L);
        for (uint32_t k = 0; k < bins; k++) A[(size_t)nn * bins + k] = w * std::cos(2.0 * M_PI * (double)(((uint64_t)nn * k) % L) / (double)L);
    }
    G.assign((size_t)L * n_mels, 0.0);
    gemm_acc(L, n_mels, bins, A.data(), bins, W.data(), n_mels, G.data(), n_mels);
}

// [N][S] audio -> [N][C][n_mels][n_frames]: the front-end as the device kernels compute it, in float64, from float32-rounded parameters
inline std::vector<double> closed_form_spectrogram(const std::vector<double> &x, size_t N, size_t S, double eps, const std::vector<RecoveredBranch> &br) {
    const size_t C = br.size();
    if (!C) return {};
    const size_t Hs = br[0].n_mels, Ws = br[0].n_frames;
    std::vector<double> out(N * C * Hs * Ws, 0.0), xn(N * S);
    for (size_t i = 0; i < N; i++) {
        double mn = INFINITY, mx = -INFINITY;
        for (size_t s = 0; s < S; s++) { mn = std::fmin(mn, x[i * S + s]); mx = std::fmax(mx, x[i * S + s]); }
        const double sc = 2.0 / ((mx - mn) + eps);
        for (size_t s = 0; s < S; s++) xn[i * S + s] = (x[i * S + s] - mn) * sc - 1.0;
    }
    for (size_t c = 0; c < C; c++) {
        const RecoveredBranch &b = br[c];
        std::vector<double> W(b.mel_w.begin(), b.mel_w.end()), G;
        hann_cos_apply(b.L, b.n_mels, W, G);
        const double expo = 1.0 / (1.0 + std::exp((double)(float)std::log(1.0 / b.expo - 1.0)));
        const double scale = (double)(float)b.scale, shift = (double)(float)b.shift;
        std::vector<double> fr((size_t)b.n_frames * b.L), T((size_t)b.n_frames * b.n_mels);
        for (size_t i = 0; i < N; i++) {
            for (uint32_t t = 0; t < b.n_frames; t++) memcpy(&fr[(size_t)t * b.L], &xn[i * S + (size_t)t * b.H], sizeof(double) * b.L);
            std::fill(T.begin(), T.end(), 0.0);
            gemm_acc(b.n_frames, b.n_mels, b.L, fr.data(), b.L, G.data(), b.n_mels, T.data(), b.n_mels);
            for (uint32_t t = 0; t < b.n_frames; t++)
                for (uint32_t m = 0; m < b.n_mels; m++) {
                    const double v = T[(size_t)t * b.n_mels + m];
                    const double o = (v == 0.0 ? 0.0 : std::exp(expo * std::log(v * v))) * scale + shift;
                    const uint32_t row = b.flip ? b.n_mels - 1 - m : m;
                    out[((i * C + c) * Hs + row) * Ws + t] = o;
                }
        }
    }
    return out;
}

inline double median_of(std::vector<double> v) {
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end());
    return v.size() % 2 ? v[v.size() / 2] : 0.5 * (v[v.size() / 2 - 1] + v[v.size() / 2]);
}

// The recovery (frontend_recover.py recover_frontend, same steps and tolerances; the frame operator's rows come from
// multi-impulse probes and its factorisation is the closed form instead of numpy's least squares).
inline Recovered recover_frontend(const Graph &g, const onnxc::ValueInfo &audio) {
    Evaluator ev(g);
    if (audio.dims.size() < 1 || audio.dims.size() > 3) throw RecoverError("audio input of rank " + std::to_string(audio.dims.size()));
    std::vector<int64_t> tail_dims(audio.dims.begin() + (audio.dims.size() > 1 ? 1 : 0), audio.dims.end());
    int64_t S = 0;
    for (int64_t d : tail_dims) {
        if (d <= 0) throw RecoverError("input '" + audio.name + "': the sample count must be static");
        if (d > 1) { if (S) throw RecoverError("input '" + audio.name + "': expected one sample axis"); S = d; }
    }
    if (S < 16 || S > (1 << 24)) throw RecoverError("input '" + audio.name + "': " + std::to_string(S) + " samples");
    const bool batched = audio.dims.size() > 1;
    auto feed = [&](const std::vector<double> &rows, size_t nrows) {
        Arr a;
        a.d = tail_dims;
        if (batched) a.d.insert(a.d.begin(), (int64_t)nrows); else if (nrows != 1) throw RecoverError("the audio input has no batch axis");
        a.v = rows;
        Evaluator::Env e;
        e[audio.name] = std::move(a);
        return e;
    };
    auto run = [&](const Evaluator::Env &feeds, const std::vector<std::string> &targets) {
        try { return ev.run(feeds, targets); }
        catch (const EvalError &e) { throw RecoverError(std::string("the front-end cannot be evaluated: ") + e.what()); }
    };

    // 1. the spectrogram tensor and the branch tensors
    std::string spec;
    int candidates = 0;
    for (const auto &n : g.nodes) {
        if (n.op != "Conv" || n.in.size() < 2) continue;
        auto it = g.init.find(n.in[1]);
        if (it == g.init.end() || it->second.dims.size() != 4 || it->second.dims[2] <= 1 || it->second.dims[3] <= 1) continue;
        if (++candidates > 64) break;     // (each test walks the graph: a hostile file must not make this quadratic)
        if (!ev.depends_on(n.in[0], audio.name)) continue;
        spec = n.in[0];
        break;
    }
    if (spec.empty()) throw RecoverError("no 2-D convolution downstream of the audio input: nowhere to enter the conv stack");
    const auto front = ev.ancestors({spec}, {audio.name});
    if (front.size() > RECOVER_MAX_NODES) throw RecoverError(std::to_string(front.size()) + " nodes in front of the first 2-D convolution: not a spectrogram front-end");
    std::vector<size_t> squarers;
    for (size_t i : front) {
        const Node &n = g.nodes[i];
        if (n.op == "Mul" && n.in.size() == 2 && n.in[0] == n.in[1]) squarers.push_back(i);
        else if (n.op == "Pow" && n.in.size() == 2) {
            bool scalar_const = false;
            try { const auto r = ev.run({}, {n.in[1]}); scalar_const = r[0].size() == 1; } catch (const EvalError &) {}
            if (scalar_const) squarers.push_back(i);
        }
    }
    std::vector<std::string> first;
    for (size_t i : squarers) {
        const std::string &t = g.nodes[i].in[0];
        const auto behind_v = ev.ancestors({t}, {audio.name});
        const std::set<size_t> behind(behind_v.begin(), behind_v.end());
        bool later = false;
        for (size_t j : squarers) later |= behind.count(j) != 0;
        if (!later && std::find(first.begin(), first.end(), t) == first.end() && ev.depends_on(t, audio.name)) first.push_back(t);
    }
    if (first.empty()) throw RecoverError("no squaring node (Mul(t, t) / Pow(t, const)) between the audio input and the spectrogram");
    if (first.size() > 8) throw RecoverError(std::to_string(first.size()) + " squared tensors: not a spectrogram front-end");

    // seeded probe signal (splitmix64: the recovery is deterministic)
    uint64_t rs = 0xB1DAull;
    auto rnd = [&]() { rs += 0x9e3779b97f4a7c15ull; uint64_t z = rs; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; z ^= z >> 31; return (double)(z >> 11) * (1.0 / 9007199254740992.0); };
    auto gauss = [&]() { const double u1 = std::max(rnd(), 1e-300), u2 = rnd(); return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2); };
    std::vector<double> x0((size_t)S);
    for (auto &v : x0) v = -0.7 + 1.4 * rnd();
    std::vector<std::string> targets{spec};
    targets.insert(targets.end(), first.begin(), first.end());
    const auto ref = run(feed(x0, 1), targets);
    if (ref[0].rank() != 4) throw RecoverError("spectrogram tensor '" + spec + "' has rank " + std::to_string(ref[0].rank()));
    const size_t C = (size_t)ref[0].d[1], Hs = (size_t)ref[0].d[2], Ws = (size_t)ref[0].d[3];
    if (first.size() != C) throw RecoverError(std::to_string(first.size()) + " squared tensors feed a " + std::to_string(C) + "-channel spectrogram: one branch per channel expected");
    if (Hs < 1 || Ws < 2 || Hs > 4096 || Ws > (1 << 20)) throw RecoverError("spectrogram of " + std::to_string(Hs) + " x " + std::to_string(Ws));
    std::vector<std::vector<int64_t>> t_shapes;
    for (size_t b = 0; b < C; b++) {
        if (ref[1 + b].is_int || ref[1 + b].rank() < 2 || ref[1 + b].d[0] != 1) throw RecoverError("branch tensor '" + first[b] + "' has shape " + shape_str(ref[1 + b].d));
        t_shapes.push_back(ref[1 + b].d);
    }

    // the tail: spectrogram [C][H][W] from values fed AT the branch tensors
    auto tail = [&](const std::vector<Arr> &vals) {
        Evaluator::Env e;
        for (size_t b = 0; b < C; b++) e[first[b]] = vals[b];
        auto r = run(e, {spec});
        if (r[0].rank() != 4 || (size_t)r[0].d[0] != 1 || (size_t)r[0].d[1] != C || (size_t)r[0].d[2] != Hs || (size_t)r[0].d[3] != Ws || r[0].is_int)
            throw RecoverError("the tail after the squaring changes the spectrogram's shape");
        return std::move(r[0].v);
    };
    auto consts = [&](double v, int only, double other) {
        std::vector<Arr> r;
        for (size_t b = 0; b < C; b++) { Arr a = make_f(t_shapes[b]); std::fill(a.v.begin(), a.v.end(), (only < 0 || only == (int)b) ? v : other); r.push_back(std::move(a)); }
        return r;
    };
    const size_t HW = Hs * Ws;

    // 2. the element-wise tail
    const auto f1 = tail(consts(1.0, -1, 1.0));
    std::vector<size_t> chan_of;
    for (size_t b = 0; b < C; b++) {
        const auto f2 = tail(consts(2.0, (int)b, 1.0));
        std::vector<size_t> hit;
        for (size_t c = 0; c < C; c++) { double d = 0; for (size_t i = 0; i < HW; i++) d = std::fmax(d, std::fabs(f2[c * HW + i] - f1[c * HW + i])); if (d > 0 || std::isnan(d)) hit.push_back(c); }
        if (hit.size() != 1) throw RecoverError("branch tensor '" + first[b] + "' reaches " + std::to_string(hit.size()) + " channels: not one channel per branch");
        chan_of.push_back(hit[0]);
    }
    { std::vector<size_t> s = chan_of; std::sort(s.begin(), s.end()); for (size_t c = 0; c < C; c++) if (s[c] != c) throw RecoverError("branches do not map onto the channels one to one"); }
    const double probe_v[5] = {0.5, 1.0, 2.0, 3.0, -1.0};
    std::vector<std::vector<double>> fv;
    for (double v : probe_v) fv.push_back(tail(consts(v, -1, v)));
    struct Tail { double expo, scale, shift; };
    std::vector<Tail> tails;
    for (size_t b = 0; b < C; b++) {
        const size_t c = chan_of[b];
        double vals[5];
        for (int q = 0; q < 5; q++) {
            const double *a = fv[(size_t)q].data() + c * HW;
            double lo = INFINITY, hi = -INFINITY, mx = 0; bool fin = true;
            for (size_t i = 0; i < HW; i++) { fin = fin && std::isfinite(a[i]); lo = std::fmin(lo, a[i]); hi = std::fmax(hi, a[i]); mx = std::fmax(mx, std::fabs(a[i])); }
            if (!fin || hi - lo > 1e-9 * std::fmax(1.0, mx))
                throw RecoverError("channel " + std::to_string(c) + ": the tail after the squaring is not one scalar function for the whole branch (per-mel affine / normalisation layers are not representable in the container)");
            vals[q] = a[0];
        }
        const double den = vals[1] - vals[0];
        if (den == 0.0 || (vals[2] - vals[1]) / den <= 0.0) throw RecoverError("channel " + std::to_string(c) + ": the tail does not depend on the squared value");
        const double expo = std::log((vals[2] - vals[1]) / den) / std::log(4.0);
        const double scale = (vals[2] - vals[1]) / (std::pow(4.0, expo) - 1.0), shift = vals[1] - scale;
        const double tol = 1e-9 * std::fmax(1.0, std::fmax(std::fabs(scale), std::fabs(shift)));
        if (!(std::fabs(scale * std::pow(9.0, expo) + shift - vals[3]) <= tol * 10) || !(std::fabs(vals[4] - vals[1]) <= tol))
            throw RecoverError("channel " + std::to_string(c) + ": the tail is not scale * (t^2)^p + shift (an even power law)");
        if (!(expo > 0.0 && expo < 1.0)) throw RecoverError("channel " + std::to_string(c) + ": exponent " + std::to_string(expo) + " is outside (0, 1), not 1 / (1 + exp(mag_scale))");
        tails.push_back({expo, scale, shift});
    }
    // axis order and mel flip: a ramp along one axis of T_b must come out along H (mel) or W (time) of its channel
    struct Axes { size_t mel, time; bool flip; };
    std::vector<Axes> axes;
    for (size_t b = 0; b < C; b++) {
        const auto &shp = t_shapes[b];
        std::vector<size_t> var;
        for (size_t a = 0; a < shp.size(); a++) if (shp[a] > 1) var.push_back(a);
        if (var.size() != 2) throw RecoverError("branch tensor '" + first[b] + "' has shape " + shape_str(shp) + ": expected a (mel, time) matrix per segment");
        int mel_ax = -1, time_ax = -1; bool flip = false;
        const auto ss = strides_of(shp);
        for (size_t a : var) {
            const size_t len = (size_t)shp[a];
            std::vector<double> ramp(len), want(len);
            double wmax = 0;
            for (size_t i = 0; i < len; i++) { ramp[i] = 1.0 + (double)i / (double)len; want[i] = tails[b].scale * std::pow(ramp[i] * ramp[i], tails[b].expo) + tails[b].shift; wmax = std::fmax(wmax, std::fabs(want[i])); }
            auto vals = consts(1.0, -1, 1.0);
            for (size_t i = 0; i < vals[b].v.size(); i++) vals[b].v[i] = ramp[(i / ss[a]) % len];
            const auto o = tail(vals);
            const double *out = o.data() + chan_of[b] * HW;           // [Hs][Ws]
            const double tol = 1e-9 * std::fmax(1.0, wmax);
            auto rows_const = [&]() { for (size_t y = 0; y < Hs; y++) for (size_t z = 1; z < Ws; z++) if (!(std::fabs(out[y * Ws + z] - out[y * Ws]) <= tol)) return false; return true; };
            auto cols_const = [&]() { for (size_t z = 0; z < Ws; z++) for (size_t y = 1; y < Hs; y++) if (!(std::fabs(out[y * Ws + z] - out[z]) <= tol)) return false; return true; };
            auto col0_is = [&](bool rev) { for (size_t y = 0; y < Hs; y++) if (!(std::fabs(out[(rev ? Hs - 1 - y : y) * Ws] - want[y]) <= tol)) return false; return true; };
            auto row0_is = [&]() { for (size_t z = 0; z < Ws; z++) if (!(std::fabs(out[z] - want[z]) <= tol)) return false; return true; };
            if (len == Hs && rows_const() && col0_is(false)) mel_ax = (int)a;
            else if (len == Hs && rows_const() && col0_is(true)) { mel_ax = (int)a; flip = true; }
            else if (len == Ws && cols_const() && row0_is()) time_ax = (int)a;
            else throw RecoverError("branch " + std::to_string(b) + ": axis " + std::to_string(a) + " of '" + first[b] + "' does not map onto the mel or the time axis of the spectrogram");
        }
        if (mel_ax < 0 || time_ax < 0) throw RecoverError("branch " + std::to_string(b) + ": could not tell the mel axis from the time axis");
        axes.push_back({(size_t)mel_ax, (size_t)time_ax, flip});
    }

    // T_b of a batch -> [rows][frames][mels]
    auto branch_matrix = [&](const Arr &t, size_t b, size_t rows) {
        if (t.is_int || t.rank() != t_shapes[b].size() || (size_t)t.d[0] != rows) throw RecoverError("branch tensor '" + first[b] + "' changes its shape with the batch size");
        for (size_t a = 1; a < t.rank(); a++) if (t.d[a] != t_shapes[b][a]) throw RecoverError("branch tensor '" + first[b] + "' changes its shape with the batch size");
        const auto ss = strides_of(t.d);
        std::vector<double> m(rows * Ws * Hs);
        for (size_t r = 0; r < rows; r++)
            for (size_t f = 0; f < Ws; f++)
                for (size_t q = 0; q < Hs; q++) m[(r * Ws + f) * Hs + q] = t.v[r * ss[0] + f * ss[axes[b].time] + q * ss[axes[b].mel]];
        return m;
    };

    // 3 + 4. the linear part, through the normalisation.  Probe signals: zero, extremes pinned at the first two samples (min = -1,
    // max = +1 whatever else the probe holds), impulses of 0.5: T is affine in the signal while min / max do not move, so
    // differences against the base response are exact.  A probe row may hold many impulses.
    const double U = 0.5;
    std::vector<double> base((size_t)S, 0.0);
    base[0] = -1.0; base[1] = 1.0;
    const size_t probe_batch = batched ? 8 : 1;
    auto responses = [&](size_t b, const std::vector<std::vector<int64_t>> &rows_pos, double scale_sig) {
        std::vector<double> out;
        std::vector<double> bs(base);
        for (auto &v : bs) v *= scale_sig;
        const auto t0 = branch_matrix(run(feed(bs, 1), {first[b]})[0], b, 1);
        for (size_t i = 0; i < rows_pos.size(); i += probe_batch) {
            const size_t nr = std::min(probe_batch, rows_pos.size() - i);
            std::vector<double> xr(nr * (size_t)S);
            for (size_t r = 0; r < nr; r++) {
                std::copy(base.begin(), base.end(), xr.begin() + r * (size_t)S);
                for (int64_t p : rows_pos[i + r]) { if (p < 2 || p >= S) throw RecoverError("internal: probe position"); xr[r * (size_t)S + (size_t)p] += U; }
                for (size_t s = 0; s < (size_t)S; s++) xr[r * (size_t)S + s] *= scale_sig;
            }
            const auto m = branch_matrix(run(feed(xr, nr), {first[b]})[0], b, nr);
            for (size_t r = 0; r < nr; r++) for (size_t q = 0; q < Ws * Hs; q++) out.push_back(m[r * Ws * Hs + q] - t0[q]);
        }
        return out;      // [rows][frames][mels]
    };
    auto absmax = [](const double *p, size_t n) { double m = 0; for (size_t i = 0; i < n; i++) m = std::fmax(m, std::fabs(p[i])); return m; };

    Recovered rec;
    rec.spectrogram = spec; rec.sample_count = (uint32_t)S; rec.spec_h = (uint32_t)Hs; rec.spec_w = (uint32_t)Ws;
    rec.branches.resize(C);
    std::vector<double> eps_est;
    for (size_t b = 0; b < C; b++) {
        const size_t n_frames = Ws, n_mels = Hs, FM = Ws * Hs;
        // the last frame an impulse reaches: t_hi = floor(p / H) (three neighbouring positions: a Hann window's first row is zero)
        const int64_t p0 = (S * 3) / 4;
        const auto d3 = responses(b, {{p0}, {p0 + 1}, {p0 + 2}}, 1.0);
        std::vector<double> mag(3 * n_frames);
        double magmax = 0;
        for (size_t r = 0; r < 3; r++) for (size_t t = 0; t < n_frames; t++) { mag[r * n_frames + t] = absmax(&d3[(r * n_frames + t) * n_mels], n_mels); magmax = std::fmax(magmax, mag[r * n_frames + t]); }
        int64_t t_hi = -1;
        for (size_t r = 0; r < 3; r++) for (size_t t = 0; t < n_frames; t++) if (mag[r * n_frames + t] > 1e-13 * std::fmax(magmax, 1e-300)) t_hi = std::max<int64_t>(t_hi, (int64_t)t);
        if (t_hi < 0 || !(magmax > 0)) throw RecoverError("branch " + std::to_string(b) + ": an impulse at sample " + std::to_string(p0) + " does not reach '" + first[b] + "'");
        if (t_hi < 1) throw RecoverError("branch " + std::to_string(b) + ": fewer than two frames");
        const double lo_h = (double)p0 / ((double)t_hi + 1.0), hi_h = ((double)p0 + 2.0) / (double)t_hi;
        std::vector<int64_t> cands;
        for (int64_t h = std::max<int64_t>(1, (int64_t)std::floor(lo_h)); h <= (int64_t)std::ceil(hi_h) && cands.size() < 16; h++) cands.push_back(h);
        // (TWO impulse positions, 37 samples apart: a Hann-windowed cosine operator is even about L / 2, so ONE impulse that lands on
        //  row L / 2 + j passes a wrong step H - 2 j as well -- row L / 2 - j of the next frame holds the same numbers; seeded plan
        //  158 of the soak run, L 512 / H 261, read as 257.  No step but the true one maps both positions onto equal rows.)
        const int64_t pm = S / 2, pm2 = pm + 37;
        std::vector<std::vector<int64_t>> rows{{pm}, {pm2}};
        std::vector<int64_t> tried;
        for (int64_t h : cands) if (pm2 + h < S) { rows.push_back({pm + h}); rows.push_back({pm2 + h}); tried.push_back(h); }
        const auto dmh = responses(b, rows, 1.0);
        const double dm_max = std::fmax(absmax(dmh.data(), FM), absmax(dmh.data() + FM, FM));
        int64_t H = 0;
        for (size_t k = 0; k < tried.size() && !H; k++) {
            double diff = 0;
            for (size_t which = 0; which < 2; which++) {
                const double *dm = dmh.data() + which * FM, *dh = dmh.data() + (2 + 2 * k + which) * FM;
                for (size_t t = 1; t < n_frames; t++) for (size_t m = 0; m < n_mels; m++) diff = std::fmax(diff, std::fabs(dh[t * n_mels + m] - dm[(t - 1) * n_mels + m]));
            }
            if (diff <= 1e-11 * std::fmax(dm_max, 1e-300) && dm_max > 0) H = tried[k];
        }
        if (!H) throw RecoverError("branch " + std::to_string(b) + ": no frame step near " + std::to_string(lo_h) + " makes the response shift-invariant");
        if ((int64_t)n_frames * H - H >= S) throw RecoverError("branch " + std::to_string(b) + ": " + std::to_string(n_frames) + " frames of step " + std::to_string(H) + " do not fit " + std::to_string(S) + " samples");
        const int64_t l_max = S - ((int64_t)n_frames - 1) * H, l_min = std::max<int64_t>(S - (int64_t)n_frames * H + 1, 1);
        // eps from the same impulse on a signal 1000 x smaller: delta T = 2 s u / (2 s + eps) . G[row]
        const double s_small = 1e-3;
        const auto ds = responses(b, {{pm}}, s_small);
        const double *dm = dmh.data();          // (the response to the impulse at pm)
        const double dm_top = absmax(dm, FM);
        std::vector<double> ratios;
        for (size_t i = 0; i < FM; i++) if (std::fabs(dm[i]) > 0.1 * dm_top) ratios.push_back(dm[i] / ds[i]);
        const double rr = median_of(ratios);
        if (!std::isfinite(rr) || std::fabs(rr * s_small - 1.0) < 1e-9) throw RecoverError("the front-end does not normalise by the segment's range (min / max): not the container's front-end");
        double eps = 2.0 * s_small * (1.0 - rr) / (rr * s_small - 1.0);
        if (std::fabs(eps) < 1e-12) eps = 0.0;
        if (!(eps >= 0 && eps <= 1e-2)) throw RecoverError("normalisation epsilon " + std::to_string(eps) + " is not plausible");
        eps_est.push_back(eps);
        const double kappa = 2.0 / (2.0 + eps);
        // every row of G: one impulse per residue class r of the frame step; frame t sees row p - t H.  Impulses of one probe row
        // are D >= l_max + 2 H apart (a multiple of H), so no frame holds two of them and the frames that can see impulse p --
        // those with 0 <= p - t H < l_max + H -- are its own.
        const int64_t span = l_max + H;                                        // rows of G that are looked at
        const int64_t D = H * ((l_max + 2 * H + H - 1) / H);
        const int64_t first_p = H * ((span + H - 1) / H + 1);                  // frames 0 .. first_p / H below the first impulse cover its `span` rows
        // (an impulse must also lie under a frame that exists: p < n_frames H, or its first rows are never seen)
        const int64_t p_end = std::min<int64_t>(S, (int64_t)n_frames * H);
        if (first_p + H > p_end) throw RecoverError("branch " + std::to_string(b) + ": segment too short for its frame geometry");
        const int64_t per_row = std::max<int64_t>(1, (p_end - first_p - H) / D + 1);
        std::vector<std::vector<int64_t>> grow;
        for (int64_t q = 0; q < H; q++) {
            const int64_t slot = q % per_row;
            if (slot == 0) grow.emplace_back();
            grow.back().push_back(first_p + slot * D + q);
        }
        const auto dg = responses(b, grow, 1.0);
        std::vector<double> G((size_t)span * n_mels, 0.0);
        std::vector<char> seen((size_t)span, 0);
        for (size_t r = 0; r < grow.size(); r++)
            for (int64_t p : grow[r])
                for (int64_t t = 0; t < (int64_t)n_frames; t++) {
                    const int64_t nrow = p - t * H;
                    if (nrow < 0 || nrow >= span) continue;
                    for (size_t m = 0; m < n_mels; m++) G[(size_t)nrow * n_mels + m] = dg[(r * n_frames + (size_t)t) * n_mels + m] / (kappa * U);
                    seen[(size_t)nrow] = 1;
                }
        double gmax = 0;
        std::vector<double> rowmag((size_t)span);
        for (int64_t nn = 0; nn < span; nn++) { rowmag[(size_t)nn] = absmax(&G[(size_t)nn * n_mels], n_mels); gmax = std::fmax(gmax, rowmag[(size_t)nn]); }
        int64_t L = 0;
        for (int64_t nn = 0; nn < span; nn++) if (rowmag[(size_t)nn] > 1e-13 * gmax) L = nn + 1;
        bool all_seen = true;
        for (int64_t nn = 0; nn < L; nn++) all_seen = all_seen && seen[(size_t)nn];
        if (!all_seen || L < l_min || L > l_max || !(gmax > 0))
            throw RecoverError("branch " + std::to_string(b) + ": support of the frame operator ends at " + std::to_string(L) + ", outside [" + std::to_string(l_min) + ", " + std::to_string(l_max) + "] implied by " + std::to_string(n_frames) + " frames");
        if (L % 2 || L < 4) throw RecoverError("branch " + std::to_string(b) + ": odd frame length " + std::to_string(L));
        // 5. G = diag(hann) . cos . W with W[0] = 0, in closed form: g[n] = G[n] / hann[n] is even about L / 2 for such an
        //    operator and its cosine series is W; g[0] (unobservable: hann[0] = 0) is the value that makes W[0] vanish
        const uint32_t bins = (uint32_t)(L / 2 + 1);
        std::vector<double> gq((size_t)L * n_mels, 0.0), W((size_t)bins * n_mels, 0.0);
        for (int64_t nn = 1; nn < L; nn++) {
            const double w = 0.5 - 0.5 * std::cos(2.0 * M_PI * (double)nn / (double)L);
            for (size_t m = 0; m < n_mels; m++) { gq[(size_t)nn * n_mels + m] = G[(size_t)nn * n_mels + m] / w; gq[m] -= gq[(size_t)nn * n_mels + m]; }
        }
        {   // W[k][m] = c_k / L . sum_n cos(2 pi k n / L) g[n][m]   (c_0 = c_{L/2} = 1, else 2): one [bins x L] x [L x mels] product
            std::vector<double> Ck((size_t)bins * (size_t)L);
            for (uint32_t k = 0; k < bins; k++) {
                const double ck = (k == 0 || k == bins - 1 ? 1.0 : 2.0) / (double)L;
                for (int64_t nn = 0; nn < L; nn++) Ck[(size_t)k * (size_t)L + (size_t)nn] = ck * std::cos(2.0 * M_PI * (double)(((uint64_t)k * (uint64_t)nn) % (uint64_t)L) / (double)L);
            }
            gemm_acc(bins, n_mels, (size_t)L, Ck.data(), (size_t)L, gq.data(), n_mels, W.data(), n_mels);
            for (size_t m = 0; m < n_mels; m++) W[m] = 0.0;
        }
        std::vector<double> Gfit;
        hann_cos_apply((uint32_t)L, (uint32_t)n_mels, W, Gfit);
        double resid = 0;
        for (size_t i = 0; i < (size_t)L * n_mels; i++) resid = std::fmax(resid, std::fabs(Gfit[i] - G[i]));
        resid /= std::fmax(gmax, 1e-300);
        if (!(resid <= 1e-6))     // (float32 operator weights in the graph leave ~1e-8)
            throw RecoverError("branch " + std::to_string(b) + ": the frame operator is not a Hann-windowed real DFT followed by a mel matrix (relative residual " + std::to_string(resid) + "): window or transform differ from what the kernels fold");
        RecoveredBranch &br = rec.branches[chan_of[b]];
        br.L = (uint32_t)L; br.H = (uint32_t)H; br.n_mels = (uint32_t)n_mels; br.n_frames = (uint32_t)n_frames; br.flip = axes[b].flip ? 1u : 0u;
        br.expo = tails[b].expo; br.scale = tails[b].scale; br.shift = tails[b].shift; br.residual = resid;
        br.mel_w.resize(W.size());
        double wmax = 0;
        for (size_t i = 0; i < W.size(); i++) { br.mel_w[i] = (float)W[i]; wmax = std::fmax(wmax, std::fabs(W[i])); }
        // (fmin / fmax are informational in the container: the band the matrix covers, from its non-zero rows, in BINS here --
        //  the caller, who knows the sample rate, scales them)
        int64_t r_lo = -1, r_hi = -1;
        for (uint32_t k = 0; k < bins; k++) if (absmax(&W[(size_t)k * n_mels], n_mels) > 1e-5 * wmax) { if (r_lo < 0) r_lo = k; r_hi = k; }
        br.fmin = r_lo < 0 ? 0.f : (float)std::max<int64_t>(r_lo - 1, 0);
        br.fmax = r_hi < 0 ? 0.f : (float)std::min<int64_t>(r_hi + 1, L / 2);
    }
    double e_lo = INFINITY, e_hi = -INFINITY, e_sum = 0;
    for (double e : eps_est) { e_lo = std::fmin(e_lo, e); e_hi = std::fmax(e_hi, e); e_sum += e; }
    if (e_hi - e_lo > 1e-9) throw RecoverError("branches disagree on the normalisation epsilon");
    rec.eps = (double)(float)(e_sum / (double)eps_est.size());

    // 6. the whole sub-graph against the closed form, on signals it has not seen
    const size_t NV = batched ? 2 : 1;
    std::vector<double> xs(NV * (size_t)S);
    for (size_t s = 0; s < (size_t)S; s++) xs[s] = -0.9 + 1.8 * rnd();
    if (NV > 1) for (size_t s = 0; s < (size_t)S; s++) xs[(size_t)S + s] = 0.31 + 0.004 * gauss();
    const auto got = run(feed(xs, NV), {spec})[0];
    const auto want = closed_form_spectrogram(xs, NV, (size_t)S, rec.eps, rec.branches);
    if (got.is_int || got.v.size() != want.size()) throw RecoverError("the spectrogram changes its shape with the batch size");
    double err = 0, wmax = 0;
    for (size_t i = 0; i < want.size(); i++) { err = std::fmax(err, std::fabs(got.v[i] - want[i])); wmax = std::fmax(wmax, std::fabs(want[i])); if (!std::isfinite(got.v[i])) err = INFINITY; }
    rec.verify_err = err / std::fmax(wmax, 1e-300);
    if (!(rec.verify_err <= 2e-5)) {
        // The spectrogram ends in |v|^(2 expo), which is not Lipschitz at v = 0: where a mel projection all but vanishes, the float32
        // rounding of the fitted mel matrix (1e-7 of the largest projection) moves the pixel by |dv|^(2 expo) -- 5e-5 of the largest
        // pixel at an exponent of 0.19 (a learned mag_scale of 1.45), found by the random plans of round 6.  Such a front-end IS the
        // graph's: it is held to the graph BEFORE the power law instead -- both spectrograms taken back through the fitted affine and
        // exponent to |v|, which a wrong exponent, affine, window or matrix moves by far more than 1e-5 of the largest projection.
        const size_t C = rec.branches.size(), Hs = rec.branches[0].n_mels, Ws = rec.branches[0].n_frames;
        double lin = 0;
        for (size_t c = 0; c < C && std::isfinite(lin); c++) {
            const RecoveredBranch &b = rec.branches[c];
            const double expo = 1.0 / (1.0 + std::exp((double)(float)std::log(1.0 / b.expo - 1.0)));
            const double scale = (double)(float)b.scale, shift = (double)(float)b.shift, inv = 1.0 / (2.0 * expo);
            double umax = 0, uerr = 0;
            for (size_t i = 0; i < NV; i++)
                for (size_t q = 0; q < Hs * Ws; q++) {
                    const size_t at = (i * C + c) * Hs * Ws + q;
                    const double ua = std::pow(std::fmax((got.v[at] - shift) / scale, 0.0), inv), ub = std::pow(std::fmax((want[at] - shift) / scale, 0.0), inv);
                    umax = std::fmax(umax, ub);
                    uerr = std::fmax(uerr, std::fabs(ua - ub));
                    if (!std::isfinite(ua)) uerr = INFINITY;
                }
            lin = std::fmax(lin, uerr / std::fmax(umax, 1e-300));
        }
        if (!(lin <= 1e-5))
            throw RecoverError("recovered front-end differs from the graph on random audio (relative error " + std::to_string(rec.verify_err) +
                               ", " + std::to_string(lin) + " in front of the power law)");
    }
    return rec;
}

}  // namespace onnxf
}  // namespace bh

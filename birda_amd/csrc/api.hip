// libbirda_hip.so -- C ABI (include/birda_hip.h) and the host executor for the gfx950 hot path.
//
// Plays the role of birdnet_onnx::Classifier + ONNX Runtime in the reference
// (src/inference/classifier.rs:191-646): owns the model, the device weights, the batch
// contexts and the per-layer kernel schedule.  No CPU compute path exists here: every
// numeric result comes from the kernels in kernels_frontend.hip / kernels_conv.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <thread>
#include <map>
#include <vector>

#include "../../include/birda_hip.h"
#include "kernels.hpp"
#include "trace.hpp"
#include "model.hpp"
#include "onnx_dense.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// Called from the catch (...) of every extern "C" entry point: no C++ exception (bad_alloc from a vector sized by the
// caller, system_error from std::thread) may cross the C ABI.
int on_exception() noexcept {
    try { throw; }
    catch (const std::bad_alloc &) { return fail(BH_ERR_INTERNAL, "out of host memory"); }
    catch (const std::exception &e) { return fail(BH_ERR_INTERNAL, "internal error: %s", e.what()); }
    catch (...) { return fail(BH_ERR_INTERNAL, "internal error (unknown exception)"); }
}

#define HIPCHK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(BH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

enum Stage { ST_MINMAX = 0, ST_MEL, ST_STEM, ST_DW, ST_PW, ST_GAP, ST_DENSE, ST_TOPK, ST_MBCONV };

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct bh_classifier {
    bh::Model model;
    int device = 0;
    uint32_t top_k = 5;
    float min_conf = 0.1f;
    std::vector<std::string> labels;
    float *d_blob = nullptr;                 // raw model blob (dw / conv weights, biases)
    std::vector<float *> d_w;                // per layer: weights as the kernels want them
    std::vector<int> ldw;                    // per layer: padded row length of d_w (pw / dense)
    std::vector<void *> d_w16;               // per layer: f16 hi / lo fragment planes (pw / dense outside fused blocks), or null
    std::vector<float> w16_unscale;          // per layer: 2^-s of those planes (they hold W * 2^s, kernels.hpp f16_scale_exponent)
    std::vector<char> head_gap;              // per layer: 1 = this 1x1 conv + GELU and the global average pool after it run as one launch
    std::vector<float *> d_owned;            // re-laid buffers to free
    bh::FrontendParams fe{};
    bh::FrontendParams *d_fe = nullptr;      // device copy read by the mel kernel
    std::vector<int> fused_at;               // per layer: index into mb (expand layer of a fused block) or -1
    std::vector<bh::MbDesc> mb;              // fused MBConv blocks (kernels_mbconv.hip)
    int twin_max_segments = 256;             // launches up to this size take the twins (one workgroup per CU at most either way)
    std::vector<bh::MbDesc> mb_small;        // per block: its small-launch twin (cfg < 0: none), same weights (mb_plan_twin)
    int precision = 0;                       // GEMM operands of the fused blocks: 0 f32, 3 f16 hi/lo split, 1 f16
    // BH_FLAG_AUTO (the default): split-f16 compute, and a row whose logits come out inf / NaN from finite samples (an activation
    // left the f16 range) is computed again on the library's own f32 kernels -- by `fb`, a second classifier of the same model
    // file built with BH_FLAG_F32 the first time that happens.  The reference's dispatch never fails a batch on operand range
    // (processor.rs:269-277) and its provider selection degrades with a recorded reason (classifier.rs:742-754).
    bool auto_fallback = false;
    std::string model_path;
    bh_classifier *fb = nullptr;
    std::mutex fb_mu;
    std::atomic<unsigned long long> fallback_segments{0};
    unsigned long long *d_stamps = nullptr;  // BIRDA_HIP_MB_STAMPS=1: [mb.size()][8] phase counters
    uint64_t mel_flops = 0;
    bh::TopkFilter filter;                   // range filter / species list applied to the kept top-k (device tables below)
    float *d_class_score = nullptr;
    unsigned char *d_species_keep = nullptr;
    float *d_bsg = nullptr;                  // intercept | slope | prior, n_classes each
    std::mutex warm_mu;
    std::set<size_t> warmed;                 // WarmupRegistry, classifier.rs:221-246
    bh_batch_context *internal_ctx = nullptr;
    std::mutex internal_mu;
    // Up to three destroyed batch contexts are parked here and handed to the next bh_batch_context_create of the same size: the
    // per-file pipeline creates and destroys a context per file (reference processor.rs:582-603), bhh_process_files keeps three in
    // flight, and a context is ~1 GB of hipMalloc plus pinned staging memory -- milliseconds per file at GPU throughput.
    static constexpr int N_PARKED = 3;
    bh_batch_context *parked_ctx[N_PARKED] = {nullptr, nullptr, nullptr};
    std::mutex parked_mu;
};

struct bh_batch_context {
    bh_classifier *c = nullptr;
    size_t max_batch = 0;        // what the buffers hold
    size_t asked_batch = 0;      // what bh_batch_context_create was asked for (a parked context of up to twice that may serve it): the
                                 // capacity the entry points enforce
    bool keep_tensors = false;
    bool keep_fused = false;   // BIRDA_HIP_KEEP_FUSED=1: a debug context still runs the fused blocks (their outputs are readable)
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;       // H2D of host batches, ahead of the compute stream
    std::vector<hipEvent_t> copy_ev;         // one per sub-slice in flight
    std::vector<hipEvent_t> done_ev;         // bh_predict_pcm*: a sub-slice's rows are in the pinned result buffers
    // Two compute lanes for the sub-slices of a host-fed slice (lanes_begin below): sub-slice k runs on stream (k & 1 ? stream2 :
    // stream) in its own part of the arena, so the launch chain of one sub-slice (21 dependent launches: ~0.85 ms however few
    // segments it holds) runs under the other's kernels instead of after them.
    static constexpr int MAX_LANES = 4;
    hipStream_t lane_stream[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};   // [0] = stream
    hipEvent_t fork_ev = nullptr, join_ev[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    int n_lanes = 3;
    int lanes_in_use = 1;        // of the slice being enqueued
    struct ArenaPlan { std::vector<size_t> t_off; size_t total = 0; };
    std::map<size_t, ArenaPlan> plans;       // arena plan of an n-segment forward (n < max_batch), built on first use
    size_t arena_cap = 0;                    // floats allocated (arena_floats + slack for the lanes' alignment losses)
    uint32_t forced_sub_slices = 0;          // bh_batch_context_set_sub_slices: 0 automatic, 1 whole slices, n equal sub-slices
    float *d_input = nullptr;    // [max_batch][sample_count]
    float *d_minmax = nullptr;   // [max_batch][8][2]
    unsigned *d_inbad = nullptr; // [max_batch][8]: the slice of the segment holds an inf / NaN sample
    float *d_arena = nullptr;
    size_t arena_floats = 0;
    std::vector<size_t> t_off;   // per tensor offset (floats) into the arena
    float *d_logits = nullptr;   // [max_batch][n_classes]
    int32_t *d_topk_idx = nullptr;
    float *d_topk_conf = nullptr;
    float *h_input = nullptr;    // pinned staging
    int16_t *d_pcm = nullptr;    // bh_predict_pcm16: the slice's span of the decoded stream (grow-only)
    size_t pcm_cap = 0;          // bytes
    unsigned long long *d_starts = nullptr;
    size_t starts_cap = 0;       // entries
    float *d_raw = nullptr;      // source-rate segments awaiting the resampler [max_batch][raw_len]
    float *h_raw = nullptr;
    size_t raw_len = 0;
    int32_t *h_topk_idx = nullptr;
    float *h_topk_conf = nullptr;
    unsigned *d_nonfinite = nullptr;   // segments whose logits came out inf / NaN from finite samples (top-k kernel), since the last check
    unsigned *h_nonfinite = nullptr;   // pinned
    // bh_forward_device calls (with top-k buffers) since the last bh_batch_context_synchronize: what BH_FLAG_AUTO re-runs from
    struct Pending { const float *d_seg; size_t n; float *d_logits; int32_t *d_idx; float *d_conf; };
    std::vector<Pending> pending;
    bool pending_overflow = false;
    size_t device_bytes = 0;
    size_t last_n = 0;
    const float *last_logits = nullptr;
    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> ev;
    std::vector<int> ev_stage;
    std::vector<int> ev_layer;   // layer index of the launch an event closes (-1: front-end / top-k)
    float stage_ms[BH_N_STAGES] = {0};
    uint32_t stage_launches[BH_N_STAGES] = {0};
};

namespace {

uint16_t f32_to_f16(float f);
float f16_to_f32(uint16_t h);

// Gf for one branch (see kernels_frontend.hip): double precision on the host, once.
// prec 0: f32 fragment-major; prec 3: f16 hi / lo planes for the split MFMA (same byte count).
std::vector<float> build_gf(const bh::BranchRec &b, const float *W, int nm_pad, int prec, int *scale_exp) {
    const int L = (int)b.frame_length, K = L / 2, nb = (int)b.n_bins, nm = (int)b.n_mels;
    std::vector<double> ct(L);
    for (int i = 0; i < L; i++) ct[i] = std::cos(2.0 * M_PI * (double)i / (double)L);
    std::vector<int> rows;
    for (int k = 0; k < nb; k++) {
        bool nz = false;
        for (int m = 0; m < nm && !nz; m++) nz = W[(size_t)k * nm + m] != 0.0f;
        if (nz) rows.push_back(k);
    }
    std::vector<float> gf((size_t)K * nm_pad, 0.0f);
    std::vector<double> acc(nm);
    for (int j = 0; j < K; j++) {
        const int n = j + 1;
        const double wn = 0.5 - 0.5 * ct[n % L];
        std::fill(acc.begin(), acc.end(), 0.0);
        for (int k : rows) {
            const double cv = ct[(size_t)((long long)k * n % L)];
            const float *wr = W + (size_t)k * nm;
            for (int m = 0; m < nm; m++) acc[m] += cv * (double)wr[m];
        }
        const double scale = (j == K - 1) ? 0.5 * wn : wn;  // the centre sample is added to itself
        for (int m = 0; m < nm; m++) gf[(size_t)j * nm_pad + m] = (float)(scale * acc[m]);
    }
    // f16 planes hold Gf * 2^s (kernels.hpp f16_scale_exponent); the kernel's power law undoes it (BranchParams::log2_bias)
    *scale_exp = 0;
    if (prec != 0) {
        float mx = 0.0f;
        for (float v : gf) mx = std::max(mx, std::fabs(v));
        *scale_exp = bh::f16_scale_exponent(mx);
        for (float &v : gf) v = std::ldexp(v, *scale_exp);
    }
    // MFMA-fragment-major relayout (kernels.hpp BranchParams::gf)
    const int mt_n = nm_pad / 16;
    std::vector<float> frag((size_t)K * nm_pad);
    if (prec == 32) {  // mel32_kernel: [step of 16 k][mel tile of 32][plane hi, lo][64 lanes][8 halves]; within a chunk of 64 k the
        // staged Y rows pair k with k + 32 in one dword, so element jj of step s holds k = 64 (s / 4) + 8 (s % 4) + 4 (lane >> 5) + jj / 2 + 32 (jj % 2)
        uint16_t *h = reinterpret_cast<uint16_t *>(frag.data());
        const int mt32 = nm_pad / 32;
        for (int st = 0; st < K / 16; st++)
            for (int mt = 0; mt < mt32; mt++)
                for (int lane = 0; lane < 64; lane++)
                    for (int jj = 0; jj < 8; jj++) {
                        const float v = gf[(size_t)(64 * (st >> 2) + 8 * (st & 3) + 4 * (lane >> 5) + (jj >> 1) + 32 * (jj & 1)) * nm_pad + 32 * mt + (lane & 31)];
                        const uint16_t hi = f32_to_f16(v);
                        const size_t base = (((size_t)st * mt32 + mt) * 2) * 64 * 8;
                        h[base + (size_t)lane * 8 + jj] = hi;
                        h[base + 64 * 8 + (size_t)lane * 8 + jj] = f32_to_f16(v - f16_to_f32(hi));
                    }
        return frag;
    }
    if (prec != 0) {  // [step of 32 k][mel tile][plane hi, lo][64 lanes][8 halves]: k = 32 s + 4 jj + (lane >> 4)
        // (the k of a step are dealt to the four lane groups round-robin, not in runs of 8: the kernel's frame-strided LDS
        //  reads of the matching samples then fall on distinct banks -- kernels_frontend.hip, mel_kernel)
        uint16_t *h = reinterpret_cast<uint16_t *>(frag.data());
        for (int st = 0; st < K / 32; st++)
            for (int mt = 0; mt < mt_n; mt++)
                for (int lane = 0; lane < 64; lane++)
                    for (int jj = 0; jj < 8; jj++) {
                        const float v = gf[(size_t)(32 * st + 4 * jj + (lane >> 4)) * nm_pad + 16 * mt + (lane & 15)];
                        const uint16_t hi = f32_to_f16(v);
                        const size_t base = (((size_t)st * mt_n + mt) * 2) * 64 * 8;
                        h[base + (size_t)lane * 8 + jj] = hi;
                        h[base + 64 * 8 + (size_t)lane * 8 + jj] = f32_to_f16(v - f16_to_f32(hi));
                    }
        return frag;
    }
    for (int g = 0; g < K / 16; g++)
        for (int mt = 0; mt < mt_n; mt++)
            for (int lane = 0; lane < 64; lane++)
                for (int c = 0; c < 4; c++) {
                    const int k = 16 * g + 4 * (lane >> 4) + c, mel = 16 * mt + (lane & 15);
                    frag[(((size_t)g * mt_n + mt) * 64 + lane) * 4 + c] = gf[(size_t)k * nm_pad + mel];
                }
    return frag;
}

int upload(const void *src, size_t bytes, float **dst) {
    HIPCHK(hipMalloc((void **)dst, bytes ? bytes : 4));
    if (bytes) HIPCHK(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return BH_OK;
}

int read_labels(const char *path, std::vector<std::string> &out) {
    FILE *f = fopen(path, "rb");
    if (!f) return fail(BH_ERR_IO, "cannot open labels file %s", path);
    std::string cur;
    int ch;
    bool any = false;
    while ((ch = fgetc(f)) != EOF) {
        any = true;
        if (ch == '\n') {
            if (!cur.empty() && cur.back() == '\r') cur.pop_back();
            out.push_back(cur);
            cur.clear();
        } else cur.push_back((char)ch);
    }
    if (any && !cur.empty()) {
        if (cur.back() == '\r') cur.pop_back();
        out.push_back(cur);
    }
    fclose(f);
    if (!out.empty() && out[0].size() >= 3 && (unsigned char)out[0][0] == 0xEF && (unsigned char)out[0][1] == 0xBB &&
        (unsigned char)out[0][2] == 0xBF)
        out[0] = out[0].substr(3);
    return BH_OK;
}

void ctx_mark(bh_batch_context *ctx, int stage, int layer = -1) {
    if (!ctx->profiling) return;
    hipEvent_t e;
    // timing-only events: without the system-scope fence (an L2 write-back + invalidate between every two kernels, which the
    // un-profiled pipeline never sees; BIRDA_HIP_EVENT_FENCE=1 restores the default events: A/B aid)
    static const bool fence = getenv("BIRDA_HIP_EVENT_FENCE") && getenv("BIRDA_HIP_EVENT_FENCE")[0] == '1';
    if (hipEventCreateWithFlags(&e, fence ? hipEventDefault : hipEventDisableSystemFence) != hipSuccess) return;
    (void)hipEventRecord(e, ctx->stream);
    ctx->ev.push_back(e);
    ctx->ev_stage.push_back(stage);
    ctx->ev_layer.push_back(layer);
}

// liveness-based arena plan: tensor t is born at step t (tensor 0 = front-end) and dies after
// the last layer that reads it; the embedding tensor and the logits live to the end.
void plan_arena(const bh::Model &m, const std::vector<int> &fused_at, const std::vector<char> &head_gap, size_t max_batch,
                bool keep, std::vector<size_t> &off, size_t &total) {
    const size_t nt = m.layers.size() + 1;
    std::vector<size_t> last(nt, 0), sz(nt);
    for (size_t t = 0; t < nt; t++) { last[t] = t; sz[t] = align_up(m.tensor_floats[t] * max_batch, 64); }
    for (size_t i = 0; i < m.layers.size(); i++) {
        const auto &L = m.layers[i];
        last[L.in_tensor] = std::max(last[L.in_tensor], i + 1);
        if (L.res_tensor != bh::NO_TENSOR) last[L.res_tensor] = std::max(last[L.res_tensor], i + 1);
    }
    if (!keep)
        for (size_t i = 0; i < fused_at.size(); i++)
            if (fused_at[i] >= 0) {
                // one launch reads the block input while it writes the block's last tensor (i + 3; i + 2 for a block without an
                // expand convolution: depthwise -> project); the tensors in between stay in LDS and take no arena space
                const size_t len = (i + 2 < m.layers.size() && m.layers[i].op != bh::OP_DWCONV) ? 3 : 2;
                last[m.layers[i].in_tensor] = std::max(last[m.layers[i].in_tensor], i + len);
                for (size_t k = 1; k < len; k++) sz[i + k] = 0;
            }
    if (!keep)
        for (size_t i = 0; i + 1 < head_gap.size(); i++)
            if (head_gap[i]) {
                // head conv + pool in one launch: workgroups still read the conv's input while finished ones store
                // pooled rows (tensor i+2), so the input lives through step i+2; the conv's output never exists
                last[m.layers[i].in_tensor] = std::max(last[m.layers[i].in_tensor], i + 2);
                sz[i + 1] = 0;
            }
    last[m.h.embedding_tensor] = nt;
    last[nt - 1] = nt;
    off.assign(nt, 0);
    total = 0;
    if (keep) {
        for (size_t t = 0; t < nt; t++) { off[t] = total; total += sz[t]; }
        return;
    }
    struct Live { size_t off, size, last; };
    std::vector<Live> live;
    for (size_t t = 0; t < nt; t++) {
        // tensors whose last reader ran before step t are dead (step t writes tensor t while
        // reading tensors with last >= t)
        live.erase(std::remove_if(live.begin(), live.end(), [&](const Live &l) { return l.last < t; }), live.end());
        std::sort(live.begin(), live.end(), [](const Live &a, const Live &b) { return a.off < b.off; });
        size_t pos = 0;
        for (const auto &l : live) {
            if (pos + sz[t] <= l.off) break;
            pos = std::max(pos, l.off + l.size);
        }
        off[t] = pos;
        live.push_back({pos, sz[t], last[t]});
        total = std::max(total, pos + sz[t]);
    }
}

int ctx_create(bh_classifier *c, size_t max_batch, bool keep, bh_batch_context **out) {
    if (!c || !out || max_batch == 0) return fail(BH_ERR_INVALID, "batch context: bad arguments");
    HIPCHK(hipSetDevice(c->device));
    auto ctx = std::make_unique<bh_batch_context>();
    ctx->c = c;
    ctx->max_batch = max_batch;
    ctx->asked_batch = max_batch;
    ctx->keep_tensors = keep;
    if (const char *kf = getenv("BIRDA_HIP_KEEP_FUSED")) ctx->keep_fused = kf[0] == '1';
    const auto &m = c->model;
    HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    const size_t in_bytes = max_batch * (size_t)m.h.sample_count * sizeof(float);
    HIPCHK(hipMalloc((void **)&ctx->d_input, in_bytes));
    HIPCHK(hipMalloc((void **)&ctx->d_minmax, max_batch * 16 * sizeof(float)));
    HIPCHK(hipMalloc((void **)&ctx->d_inbad, max_batch * 8 * sizeof(unsigned)));
    plan_arena(m, c->fused_at, c->head_gap, max_batch, keep, ctx->t_off, ctx->arena_floats);
    ctx->arena_cap = ctx->arena_floats + 8 * 64 * (m.layers.size() + 1);
    HIPCHK(hipMalloc((void **)&ctx->d_arena, ctx->arena_cap * sizeof(float)));
    if (const char *e = getenv("BIRDA_HIP_NLANES")) ctx->n_lanes = std::max(1, std::min((int)bh_batch_context::MAX_LANES, atoi(e)));
    ctx->lane_stream[0] = ctx->stream;   // (the other lanes' streams are created when a slice first needs them: lanes_begin)
    HIPCHK(hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming));
    HIPCHK(hipMalloc((void **)&ctx->d_logits, max_batch * (size_t)m.h.n_classes * sizeof(float)));
    HIPCHK(hipMalloc((void **)&ctx->d_topk_idx, max_batch * c->top_k * sizeof(int32_t)));
    HIPCHK(hipMalloc((void **)&ctx->d_topk_conf, max_batch * c->top_k * sizeof(float)));
    HIPCHK(hipHostMalloc((void **)&ctx->h_input, in_bytes, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&ctx->h_topk_idx, max_batch * c->top_k * sizeof(int32_t), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&ctx->h_topk_conf, max_batch * c->top_k * sizeof(float), hipHostMallocDefault));
    HIPCHK(hipMalloc((void **)&ctx->d_nonfinite, sizeof(unsigned)));
    HIPCHK(hipMemset(ctx->d_nonfinite, 0, sizeof(unsigned)));
    HIPCHK(hipHostMalloc((void **)&ctx->h_nonfinite, sizeof(unsigned), hipHostMallocDefault));
    *ctx->h_nonfinite = 0;
    ctx->device_bytes = in_bytes + max_batch * 16 * sizeof(float) + ctx->arena_floats * sizeof(float) +
                        max_batch * (size_t)m.h.n_classes * sizeof(float) + max_batch * c->top_k * 8;
    *out = ctx.release();
    return BH_OK;
}

void ctx_destroy(bh_batch_context *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->c->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto e : ctx->ev) (void)hipEventDestroy(e);
    for (auto e : ctx->copy_ev) (void)hipEventDestroy(e);
    for (auto e : ctx->done_ev) (void)hipEventDestroy(e);
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    for (int l = 1; l < bh_batch_context::MAX_LANES; l++) {
        if (ctx->lane_stream[l]) { (void)hipStreamSynchronize(ctx->lane_stream[l]); (void)hipStreamDestroy(ctx->lane_stream[l]); }
        if (ctx->join_ev[l]) (void)hipEventDestroy(ctx->join_ev[l]);
    }
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    (void)hipFree(ctx->d_input); (void)hipFree(ctx->d_minmax); (void)hipFree(ctx->d_inbad); (void)hipFree(ctx->d_arena);
    (void)hipFree(ctx->d_logits); (void)hipFree(ctx->d_topk_idx); (void)hipFree(ctx->d_topk_conf);
    (void)hipHostFree(ctx->h_input); (void)hipHostFree(ctx->h_topk_idx); (void)hipHostFree(ctx->h_topk_conf);
    (void)hipFree(ctx->d_raw); (void)hipHostFree(ctx->h_raw);
    (void)hipFree(ctx->d_pcm); (void)hipFree(ctx->d_starts);
    (void)hipFree(ctx->d_nonfinite); (void)hipHostFree(ctx->h_nonfinite);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// one slice (n <= max_batch) of the forward pass, enqueued on ctx->stream
// (lane: where a sub-slice of a host-fed slice runs -- its stream, its own part of the arena with the plan of an n-segment forward,
//  and its first segment's index within the slice, which places its rows of the per-segment scratch; nullptr: the context's stream
//  and whole arena)
struct SliceLane { hipStream_t s; float *arena; const size_t *t_off; size_t seg0; };
int forward_slice(bh_classifier *c, bh_batch_context *ctx, const float *d_seg, size_t n, float *d_logits,
                  int32_t *d_idx, float *d_conf, const SliceLane *lane = nullptr) {
    const auto &m = c->model;
    hipStream_t s = lane ? lane->s : ctx->stream;
    float *const arena = lane ? lane->arena : ctx->d_arena;
    const size_t *const t_off = lane ? lane->t_off : ctx->t_off.data();
    float *const d_minmax = ctx->d_minmax + (lane ? lane->seg0 * 16 : 0);
    unsigned *const d_inbad = ctx->d_inbad + (lane ? lane->seg0 * 8 : 0);
    auto T = [&](uint32_t t) { return arena + t_off[t]; };
    const uint32_t nl = (uint32_t)m.layers.size();
    bh::TraceRange tr_slice("bh_forward_slice");   // ROCTx ranges (BIRDA_HIP_ROCTX=1): the slice, its front end, every layer group
    ctx_mark(ctx, -1);
    {
        bh::TraceRange tr("front_end: minmax + mel");
        bh::launch_minmax(d_seg, d_minmax, d_inbad, (int)n, (int)m.h.sample_count, s);
        ctx_mark(ctx, ST_MINMAX);
        bh::launch_mel(d_seg, d_minmax, T(0), c->fe, c->d_fe, (int)n, s);
        ctx_mark(ctx, ST_MEL);
    }
    for (uint32_t i = 0; i < nl; i++) {
        const auto &L = m.layers[i];
        static const char *const kOpNames[] = {"layer", "conv", "depthwise", "pointwise", "pool", "dense", "scale"};
        bh::TraceRange tr((!ctx->keep_tensors || ctx->keep_fused) && c->fused_at[i] >= 0 ? "fused_mbconv_block"
                          : L.op < sizeof(kOpNames) / sizeof(kOpNames[0]) ? kOpNames[L.op] : kOpNames[0]);
        const float *in = T(L.in_tensor);
        float *out = (i == nl - 1) ? d_logits : T(i + 1);
        const float *res = L.res_tensor != bh::NO_TENSOR ? T(L.res_tensor) : nullptr;
        const float *bias = c->d_blob + L.b_off;
        bh::ConvParams p{(int)L.in_h, (int)L.in_w, (int)L.out_h, (int)L.out_w, (int)L.cin, (int)L.cout,
                         (int)L.kh, (int)L.kw, (int)L.sh, (int)L.sw, (int)L.pad_t, (int)L.pad_l,
                         (int)L.in_layout, (int)L.act};
        if ((!ctx->keep_tensors || ctx->keep_fused) && c->fused_at[i] >= 0) {
            // expand (i) -> depthwise (i+1) -> project (i+2) in one launch
            // (a launch of at most 256 segments: the one-segment-per-workgroup twin where the block has one -- the two-segment
            //  tiles would leave half of the CUs, or more, without a workgroup)
            const bh::MbDesc &twin = c->mb_small[c->fused_at[i]];
            bh::MbDesc d = (twin.cfg >= 0 && n <= (size_t)c->twin_max_segments) ? twin : c->mb[c->fused_at[i]];
            const size_t ip = d.noexp ? i + 1 : i + 2;   // the project layer
            const auto &LP = m.layers[ip];
            d.X = in;
            d.Y = (ip == nl - 1) ? d_logits : T(ip + 1);
            d.R = LP.res_tensor != bh::NO_TENSOR ? T(LP.res_tensor) : nullptr;
            bh::launch_mbconv(d, (int)n, s);
            ctx_mark(ctx, ST_MBCONV, (int)i);
            i = ip;
            continue;
        }
        switch (L.op) {
        case bh::OP_CONV:
            bh::launch_conv_direct(in, c->d_w[i], bias, out, p, (int)n, s);
            ctx_mark(ctx, ST_STEM, (int)i);
            break;
        case bh::OP_DWCONV:
            bh::launch_dwconv(in, c->d_w[i], bias, out, p, (int)n, s);
            ctx_mark(ctx, ST_DW, (int)i);
            break;
        case bh::OP_PWCONV:
            if (!ctx->keep_tensors && c->head_gap[i]) {
                float *pooled = (i + 1 == nl - 1) ? d_logits : T(i + 2);
                bh::launch_head_gap16(in, c->d_w16[i], bias, pooled, (int)n, (int)(L.out_h * L.out_w), (int)L.cin, (int)L.cout,
                                      (int)L.act, c->precision == 3 ? 3 : 1, c->w16_unscale[i], s);
                ctx_mark(ctx, ST_PW, (int)i);
                i += 1;   // the pool layer is done
                break;
            }
            if (!ctx->keep_tensors && c->d_w16[i])
                bh::launch_pw_gemm16(in, c->d_w16[i], bias, res, out, (int)(n * L.out_h * L.out_w), (int)L.cin, (int)L.cout,
                                     (int)L.act, c->precision == 3 ? 3 : 1, c->w16_unscale[i], s);
            else
            bh::launch_pw_gemm(in, c->d_w[i], bias, res, out, (int)(n * L.out_h * L.out_w), (int)L.cin,
                               (int)L.cout, c->ldw[i], (int)L.act, s);
            ctx_mark(ctx, ST_PW, (int)i);
            break;
        case bh::OP_DENSE:
            if (!ctx->keep_tensors && c->d_w16[i])
                bh::launch_pw_gemm16(in, c->d_w16[i], bias, res, out, (int)n, (int)L.cin, (int)L.cout, (int)L.act,
                                     c->precision == 3 ? 3 : 1, c->w16_unscale[i], s);
            else
            bh::launch_pw_gemm(in, c->d_w[i], bias, res, out, (int)n, (int)L.cin, (int)L.cout, c->ldw[i],
                               (int)L.act, s);
            ctx_mark(ctx, ST_DENSE, (int)i);
            break;
        case bh::OP_GAP:
            bh::launch_gap(in, out, (int)n, (int)(L.in_h * L.in_w), (int)L.cout, s);
            ctx_mark(ctx, ST_GAP, (int)i);
            break;
        case bh::OP_SCALE:   // squeeze-excite: the feature map times its [n][C] gate (res = the gate tensor)
            bh::launch_scale(in, res, out, (int)n, (int)(L.out_h * L.out_w), (int)L.cout, s);
            ctx_mark(ctx, ST_DW, (int)i);
            break;
        default: return fail(BH_ERR_UNSUPPORTED, "layer %u: unsupported op %u", i, L.op);
        }
    }
    if (d_idx && d_conf) {
        bh::TraceRange tr("topk");
        bh::launch_topk(d_logits, (int)n, (int)m.h.n_classes, (int)m.h.output_activation, (int)c->top_k,
                        c->min_conf, c->filter, d_idx, d_conf, d_inbad, ctx->d_nonfinite, s);
        ctx_mark(ctx, ST_TOPK);
    }
    HIPCHK(hipGetLastError());
    ctx->last_n = n;
    ctx->last_logits = d_logits;
    return BH_OK;
}

// The non-finite counter of the forwards enqueued so far: fetch_nonfinite() goes on the stream BEFORE the synchronise that
// ends a call, nonfinite_status() after it.  A hit is an error, not a silent NaN row: in the f16 operand modes an activation
// at or beyond 65 504 turns into inf inside bh_split2 / the f16 conversions (weights cannot: they are pre-scaled at create).
hipError_t fetch_nonfinite(bh_batch_context *ctx) {
    return hipMemcpyAsync(ctx->h_nonfinite, ctx->d_nonfinite, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream);
}
int nonfinite_status(bh_classifier *c, bh_batch_context *ctx) {
    const unsigned bad = *ctx->h_nonfinite;
    if (!bad) return BH_OK;
    *ctx->h_nonfinite = 0;
    (void)hipMemsetAsync(ctx->d_nonfinite, 0, sizeof(unsigned), ctx->stream);
    return fail(BH_ERR_NONFINITE, "%u segment(s) with finite samples produced inf / NaN logits%s", bad,
                c->precision != 0 ? ": an activation left the f16 operand range (|x| >= 65504); build the classifier with BH_FLAG_AUTO (the default: such rows are re-run in f32) or BH_FLAG_F32"
                                  : " (f32 overflow inside the network)");
}

void clear_nonfinite(bh_batch_context *ctx) {
    *ctx->h_nonfinite = 0;
    (void)hipMemsetAsync(ctx->d_nonfinite, 0, sizeof(unsigned), ctx->stream);
}

// ---- BH_FLAG_AUTO: the rows the split-f16 forward could not represent, again on the f32 kernels -----------------------------
constexpr uint32_t FLAG_INTERNAL_NO_ENV = 0x80000000u;   // (the fallback classifier: BIRDA_HIP_PRECISION must not turn it into f16 again)
int internal_ctx(bh_classifier *c, size_t n, bh_batch_context **out);

int fallback_classifier(bh_classifier *c, bh_classifier **out) {
    std::lock_guard<std::mutex> g(c->fb_mu);
    if (!c->fb) {
        bh_config cfg{};
        cfg.model_path = c->model_path.c_str();
        cfg.labels_path = nullptr;
        cfg.top_k = c->top_k; cfg.min_confidence = c->min_conf; cfg.device = c->device;
        cfg.flags = BH_FLAG_F32 | FLAG_INTERNAL_NO_ENV;
        const int rc = bh_classifier_create(&cfg, &c->fb);
        if (rc != BH_OK) { c->fb = nullptr; return rc; }
    }
    *out = c->fb;
    return BH_OK;
}

// Rows `bad` of one forward (inputs d_seg [.][sample_count], outputs as given; d_idx / d_conf / d_emb nullable) computed again by
// the f32 classifier and written over the split-f16 ones.  The caller has synchronised whatever produced those rows.  Top-k
// goes through the PRIMARY classifier's filters (range filter / species list / BSG: device tables of the same device).
int redo_rows_f32(bh_classifier *c, const float *d_seg, const std::vector<size_t> &bad, float *d_logits, int32_t *d_idx, float *d_conf,
                  float *d_emb) {
    if (bad.empty()) return BH_OK;
    bh_classifier *fb = nullptr;
    int rc = fallback_classifier(c, &fb);
    if (rc != BH_OK) return rc;
    std::lock_guard<std::mutex> g(fb->internal_mu);
    const auto &h = c->model.h;
    const size_t S = h.sample_count, NC = h.n_classes, TK = c->top_k, ED = h.embedding_dim;
    fb->filter = c->filter;
    fb->min_conf = c->min_conf;
    constexpr size_t CH = 64;
    bh_batch_context *fx = nullptr;
    rc = internal_ctx(fb, std::min(bad.size(), CH), &fx);
    if (rc != BH_OK) return rc;
    for (size_t q0 = 0; q0 < bad.size(); q0 += CH) {
        const size_t nq = std::min(CH, bad.size() - q0);
        for (size_t j = 0; j < nq; j++)
            HIPCHK(hipMemcpyAsync(fx->d_input + j * S, d_seg + bad[q0 + j] * S, S * sizeof(float), hipMemcpyDeviceToDevice, fx->stream));
        rc = forward_slice(fb, fx, fx->d_input, nq, fx->d_logits, fx->d_topk_idx, fx->d_topk_conf);
        if (rc != BH_OK) return rc;
        const float *f_emb = (d_emb && ED) ? fx->d_arena + fx->t_off[h.embedding_tensor] : nullptr;
        for (size_t j = 0; j < nq; j++) {
            const size_t i = bad[q0 + j];
            if (d_logits) HIPCHK(hipMemcpyAsync(d_logits + i * NC, fx->d_logits + j * NC, NC * sizeof(float), hipMemcpyDeviceToDevice, fx->stream));
            if (d_idx) HIPCHK(hipMemcpyAsync(d_idx + i * TK, fx->d_topk_idx + j * TK, TK * sizeof(int32_t), hipMemcpyDeviceToDevice, fx->stream));
            if (d_conf) HIPCHK(hipMemcpyAsync(d_conf + i * TK, fx->d_topk_conf + j * TK, TK * sizeof(float), hipMemcpyDeviceToDevice, fx->stream));
            if (f_emb) HIPCHK(hipMemcpyAsync(d_emb + i * ED, f_emb + j * ED, ED * sizeof(float), hipMemcpyDeviceToDevice, fx->stream));
        }
        HIPCHK(fetch_nonfinite(fx));
        HIPCHK(hipStreamSynchronize(fx->stream));
        if (*fx->h_nonfinite) {
            const unsigned n_bad = *fx->h_nonfinite;
            clear_nonfinite(fx);
            return fail(BH_ERR_NONFINITE, "%u segment(s) with finite samples produced inf / NaN logits on the f32 kernels too (f32 overflow inside the network)", n_bad);
        }
    }
    c->fallback_segments += bad.size();
    return BH_OK;
}

// Rows [r0, r0 + nr) of a host-fed slice whose results are in the context's buffers (inputs in ctx->d_input, host copies of the
// top-k rows in h_topk_*), after the synchronise that ended the slice: under BH_FLAG_AUTO the rows the top-k stage marked
// (BH_TOPK_NONFINITE in their first slot) are re-run and their host copies refreshed; *redone says whether any were.
int settle_rows(bh_classifier *c, bh_batch_context *ctx, size_t r0, size_t nr, float *d_emb, bool *redone) {
    *redone = false;
    if (!c->auto_fallback) return BH_OK;
    const size_t TK = c->top_k;
    std::vector<size_t> bad;
    for (size_t i = r0; i < r0 + nr; i++)
        if (ctx->h_topk_idx[i * TK] == BH_TOPK_NONFINITE) bad.push_back(i);
    if (bad.empty()) return BH_OK;
    const int rc = redo_rows_f32(c, ctx->d_input, bad, ctx->d_logits, ctx->d_topk_idx, ctx->d_topk_conf, d_emb);
    if (rc != BH_OK) return rc;
    for (size_t i : bad) {
        HIPCHK(hipMemcpy(ctx->h_topk_idx + i * TK, ctx->d_topk_idx + i * TK, TK * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ctx->h_topk_conf + i * TK, ctx->d_topk_conf + i * TK, TK * sizeof(float), hipMemcpyDeviceToHost));
    }
    *redone = true;
    return BH_OK;
}

int check_ctx(bh_classifier *c, bh_batch_context *ctx) {
    if (!c || !ctx) return fail(BH_ERR_INVALID, "null classifier or batch context");
    if (ctx->c != c) return fail(BH_ERR_INVALID, "batch context belongs to another classifier");
    return BH_OK;
}

// Host threads that gather the caller's segment slices into the pinned staging buffer
// (BIRDA_HIP_COPY_THREADS; default min(8, hardware threads / 2)).  One thread moves ~10 GB/s, a fifth of
// what the PCIe link takes.
unsigned copy_threads() {
    static const unsigned n = [] {
        if (const char *e = getenv("BIRDA_HIP_COPY_THREADS")) return (unsigned)std::max(1, atoi(e));
        const unsigned hw = std::thread::hardware_concurrency();
        return std::max(1u, std::min(8u, hw / 2));
    }();
    return n;
}

// Sub-slices of a host-fed slice: each is computed as soon as its own samples are on the copy stream.  A forward has a floor of
// about a millisecond however few segments it holds (21 launches whose workgroups walk their chunk loops serially) and the late
// blocks need >= 128 segments to fill the GPU.
//   * `lanes` (the normal case, lanes_begin below: sub-slices alternate between the context's compute streams, so one's launch
//     chain runs under another's kernels): EQUAL sub-slices of an eighth of the slice, at least 128 segments -- the first one
//     starts the device after 1/8 of the upload, and the floors overlap (measured, 1 000 PCM16 segments from pinned memory:
//     8 sub-slices on 3 lanes 8.8 ms, 6 on 2 lanes 9.1, 4 on 2 lanes 9.5, round 3's growing split in sequence 10.4;
//     12 sub-slices of 84 segments 10.4 again).
//   * in sequence on one stream (profiling / debug contexts, the resampling path, BIRDA_HIP_LANES=0) the split depends on which side
//     is the longer one (a small pipeline model, upload 55 GB/s, forward 0.85 ms + t per segment):
//       - the upload is SHORTER than the compute (PCM16 mono into the v2.4-shaped model: 5.2 against 6.3 us per segment; anything
//         into the Perch-sized one): few sub-slices of GROWING size -- a small first one starts the compute stream early, the last,
//         more than half of the slice, runs at the full-batch rate while nothing is left to upload: (1/8, 1/3, rest) = 9.5 ms per
//         1 000 segments against 11.0 for four quarters;
//       - the upload is LONGER (f32 segments: 10.5 us): the forward of the last sub-slice is all that is left after the last byte
//         has arrived, so it must be small: equal quarters (12.9 ms against 14.8 for the growing split).
// Boundaries are multiples of `align` segments.
bool lanes_possible(const bh_batch_context *ctx) {
    static const bool off = getenv("BIRDA_HIP_LANES") && getenv("BIRDA_HIP_LANES")[0] == '0';
    return !off && !ctx->profiling && !ctx->keep_tensors && ctx->n_lanes >= 2;
}
std::vector<size_t> sub_slice_cuts(const bh_classifier *c, size_t nb, size_t align, bool single, size_t bytes_per_segment, bool lanes = false,
                                   uint32_t ctx_forced = 0) {
    std::vector<size_t> cuts;
    static const int env_forced = getenv("BIRDA_HIP_SUBSLICES") ? atoi(getenv("BIRDA_HIP_SUBSLICES")) : 0;   // (A/B aid: n equal sub-slices)
    const int forced = ctx_forced ? (int)ctx_forced : env_forced;
    auto up = [&](size_t v) { return std::min(nb, (v + align - 1) / align * align); };
    if (forced == 1) {
    } else if (!single && forced > 1 && nb >= 256) {
        const size_t sub = up((nb + forced - 1) / forced);
        for (size_t v = sub; v < nb; v += sub) cuts.push_back(v);
    } else if (!single && lanes && nb >= 256) {
        const size_t sub = up(std::max<size_t>(128, (nb + 7) / 8));
        // a first sub-slice of 64 segments where the upload is the shorter side (PCM16: 5.2 against 6.3 us per segment): the device
        // starts after 0.33 ms instead of 0.65 and the short forward's floor runs under the next sub-slices (pinned PCM16
        // 111-114 k -> 115-117 k segments/s; 32 or 96 segments: no gain; pinned f32 segments, upload-bound, lose 2 % to it)
        static const int first_env = getenv("BIRDA_HIP_FIRST_SUBSLICE") ? atoi(getenv("BIRDA_HIP_FIRST_SUBSLICE")) : -1;   // (A/B aid)
        const double upload_us = (double)bytes_per_segment / 55e3;
        const double compute_us = (2.0 * (double)c->model.macs_per_segment() + (double)c->mel_flops) / 130e6;
        const size_t first = first_env >= 0 ? (size_t)first_env : (upload_us < compute_us ? 64 : 0);
        size_t v0 = sub;
        if (first && first < sub && nb >= 512) { cuts.push_back(up(first)); v0 = up(first) + sub; }
        for (size_t v = v0; v + 64 <= nb; v += sub) cuts.push_back(v);   // (a tail under 64 segments joins the last sub-slice)
    } else
    if (!single && nb >= 512) {
        const double upload_us = (double)bytes_per_segment / 55e3;
        const double compute_us = (2.0 * (double)c->model.macs_per_segment() + (double)c->mel_flops) / 130e6;   // ~130 TFLOP/s over the whole forward
        if (upload_us < compute_us) {
            const size_t c1 = up(std::max<size_t>(128, nb / 8)), c2 = up(std::max<size_t>(c1 + 128, nb * 9 / 20));
            if (c1 < nb) cuts.push_back(c1);
            if (c2 < nb && c2 > c1) cuts.push_back(c2);
        } else {
            const size_t sub = up(std::max<size_t>(128, (nb + 3) / 4));
            for (size_t v = sub; v < nb; v += sub) cuts.push_back(v);
        }
    }
    cuts.push_back(nb);
    return cuts;
}

// The sub-slices of one host-fed slice as concurrent lanes: the plan of each sub-slice's own size, side by side in the arena,
// alternating between the context's two compute streams.  A forward is a chain of 21 dependent launches (~0.85 ms however few
// segments it holds, and the late blocks need >= 128 segments to fill the chip); one after the other on one stream the chains of
// three sub-slices were 2.5 ms of a 10.4 ms slice.  False -- run them in sequence on the context's stream, as before -- for
// single-slice calls, profiling / debug contexts, BIRDA_HIP_LANES=0, or if the plans do not fit the arena.
bool lanes_begin(bh_classifier *c, bh_batch_context *ctx, const std::vector<size_t> &cuts, std::vector<SliceLane> &lanes) {
    lanes.clear();
    if (cuts.size() < 2 || !lanes_possible(ctx)) return false;
    if (ctx->plans.size() > 64) ctx->plans.clear();
    // streams for as many lanes as this slice has sub-slices, created on first use: the runtime maps a process's streams onto a few
    // hardware queues (four by default) in creation order, and every stream that exists -- used or not -- shifts that mapping
    const int want = (int)std::min<size_t>((size_t)ctx->n_lanes, cuts.size());
    for (int l = 1; l < want; l++)
        if (!ctx->lane_stream[l]) {
            if (hipStreamCreateWithFlags(&ctx->lane_stream[l], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&ctx->join_ev[l], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
        }
    ctx->lanes_in_use = want;
    size_t base = 0;
    for (size_t si = 0; si < cuts.size(); si++) {
        const size_t s0 = si ? cuts[si - 1] : 0, ns = cuts[si] - s0;
        auto it = ctx->plans.find(ns);
        if (it == ctx->plans.end()) {
            bh_batch_context::ArenaPlan p;
            plan_arena(c->model, c->fused_at, c->head_gap, ns, false, p.t_off, p.total);
            it = ctx->plans.emplace(ns, std::move(p)).first;
        }
        if (base + it->second.total > ctx->arena_cap) { lanes.clear(); return false; }
        lanes.push_back({ctx->lane_stream[si % (size_t)want], ctx->d_arena + base, it->second.t_off.data(), s0});
        base += align_up(it->second.total, 64);
    }
    // the other lanes' streams start behind everything enqueued on the context's stream so far
    bool ok = hipEventRecord(ctx->fork_ev, ctx->stream) == hipSuccess;
    for (int l = 1; ok && l < want; l++) ok = hipStreamWaitEvent(ctx->lane_stream[l], ctx->fork_ev, 0) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); lanes.clear(); }
    return ok;
}
// ... and the context's stream continues behind them (result downloads, the next slice)
hipError_t lanes_end(bh_batch_context *ctx, const std::vector<SliceLane> &lanes) {
    if (lanes.empty()) return hipSuccess;
    for (int l = 1; l < ctx->lanes_in_use; l++) {
        hipError_t e = hipEventRecord(ctx->join_ev[l], ctx->lane_stream[l]);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->join_ev[l], 0);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
void lanes_sync(bh_batch_context *ctx) {
    for (int l = 1; l < bh_batch_context::MAX_LANES; l++)
        if (ctx->lane_stream[l]) (void)hipStreamSynchronize(ctx->lane_stream[l]);
}

// host slices -> results through ctx.
// A slice of up to max_batch segments is pipelined three ways: worker threads gather 32-segment chunks into
// pinned memory; each chunk's H2D copy is enqueued on the copy stream as soon as it is complete; and the
// slice is computed in up to four sub-slices on the compute stream, each waiting only for its own copies
// (reference: the decode thread filling the channel while the main thread runs batches,
// src/pipeline/processor.rs:647-671).
int predict_slices(bh_classifier *c, bh_batch_context *ctx, const float *const *segments, const float *contig,
                   size_t n, bh_result *out, float *logits_out, float *emb_out, bool whole_slice = false) {
    const auto &m = c->model;
    const size_t S = m.h.sample_count, NC = m.h.n_classes, TK = c->top_k;
    HIPCHK(hipSetDevice(c->device));
    if (segments)
        for (size_t i = 0; i < n; i++)
            if (!segments[i]) return fail(BH_ERR_INVALID, "segment %zu is null", i);
    // A contiguous list in PINNED (hipHostMalloc / hipHostRegister: bh_host_alloc, bh_host_register) memory goes to the device
    // straight from the caller's buffer: the gather into the context's own pinned staging -- a host memcpy, the bound of this
    // entry point for pageable input -- is skipped.
    bool pinned_src = false;
    if (contig && n > 0) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, contig) == hipSuccess) pinned_src = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError();   // pageable memory: "invalid value", not an error of this call
    }
    for (size_t b0 = 0; b0 < n; b0 += ctx->max_batch) {
        const size_t nb = std::min(ctx->max_batch, n - b0);
        constexpr size_t CH = 32;
        const size_t nchunks = (nb + CH - 1) / CH;
        // sub-slices (sub_slice_cuts above), in whole chunks; debug contexts keep one (bh_debug_read_tensor reads the last)
        // (whole_slice: the caller reads an arena tensor of the slice afterwards -- the embeddings of the two-stage path)
        const bool one = ctx->keep_tensors || emb_out || whole_slice;
        std::vector<size_t> cuts = sub_slice_cuts(c, nb, CH, one, S * sizeof(float), lanes_possible(ctx), ctx->forced_sub_slices);
        std::vector<SliceLane> lanes;
        if (!lanes_begin(c, ctx, cuts, lanes) && cuts.size() > 1) cuts = sub_slice_cuts(c, nb, CH, one, S * sizeof(float), false, ctx->forced_sub_slices);
        const size_t nsub = cuts.size();
        while (ctx->copy_ev.size() < nsub) {
            hipEvent_t e;
            HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->copy_ev.push_back(e);
        }
        auto gather = [&](size_t j) {
            const size_t i1 = std::min(nb, (j + 1) * CH);
            for (size_t i = j * CH; i < i1; i++) {
                const float *src = segments ? segments[b0 + i] : contig + (b0 + i) * S;
                memcpy(ctx->h_input + i * S, src, S * sizeof(float));
            }
        };
        const unsigned nthreads = pinned_src ? 1u : (unsigned)std::min<size_t>(copy_threads(), nchunks);
        std::vector<std::atomic<int>> done(nchunks);
        for (auto &d : done) d.store(0, std::memory_order_relaxed);
        std::atomic<size_t> next{0};
        std::vector<std::thread> workers;
        if (nthreads > 1)
            for (unsigned t = 0; t < nthreads; t++)
                workers.emplace_back([&] {
                    for (size_t j; (j = next.fetch_add(1)) < nchunks;) { gather(j); done[j].store(1, std::memory_order_release); }
                });
        int rc = BH_OK;
        size_t si_next = 0;
        for (size_t j = 0; j < nchunks && rc == BH_OK; j++) {
            if (nthreads > 1) while (!done[j].load(std::memory_order_acquire)) std::this_thread::yield();
            else if (!pinned_src) gather(j);
            const size_t i0 = j * CH, i1 = std::min(nb, (j + 1) * CH);
            const float *h_src = pinned_src ? contig + (b0 + i0) * S : ctx->h_input + i0 * S;
            if (hipMemcpyAsync(ctx->d_input + i0 * S, h_src, (i1 - i0) * S * sizeof(float), hipMemcpyHostToDevice,
                               ctx->copy_stream) != hipSuccess) { rc = fail(BH_ERR_HIP, "H2D copy failed"); break; }
            if (i1 == cuts[si_next]) {   // a sub-slice is complete on the copy stream: compute it
                const size_t si = si_next++, s0 = si ? cuts[si - 1] : 0, ns = i1 - s0;
                const SliceLane *lane = lanes.empty() ? nullptr : &lanes[si];
                if (hipEventRecord(ctx->copy_ev[si], ctx->copy_stream) != hipSuccess ||
                    hipStreamWaitEvent(lane ? lane->s : ctx->stream, ctx->copy_ev[si], 0) != hipSuccess) { rc = fail(BH_ERR_HIP, "stream event failed"); break; }
                rc = forward_slice(c, ctx, ctx->d_input + s0 * S, ns, ctx->d_logits + s0 * NC, ctx->d_topk_idx + s0 * TK,
                                   ctx->d_topk_conf + s0 * TK, lane);
            }
        }
        for (auto &w : workers) w.join();
        if (rc == BH_OK && lanes_end(ctx, lanes) != hipSuccess) rc = fail(BH_ERR_HIP, "stream event failed");
        if (rc != BH_OK) {
            (void)hipStreamSynchronize(ctx->copy_stream); lanes_sync(ctx); (void)hipStreamSynchronize(ctx->stream);
            return rc;
        }
        if (out) {
            HIPCHK(hipMemcpyAsync(ctx->h_topk_idx, ctx->d_topk_idx, nb * TK * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipMemcpyAsync(ctx->h_topk_conf, ctx->d_topk_conf, nb * TK * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        }
        if (logits_out)
            HIPCHK(hipMemcpyAsync(logits_out + b0 * NC, ctx->d_logits, nb * NC * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        if (emb_out)
            HIPCHK(hipMemcpyAsync(emb_out + b0 * m.h.embedding_dim, ctx->d_arena + ctx->t_off[m.h.embedding_tensor],
                                  nb * (size_t)m.h.embedding_dim * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        if (!out && c->auto_fallback)   // (logits-only calls: the marks of the top-k rows are how the bad rows are found)
            HIPCHK(hipMemcpyAsync(ctx->h_topk_idx, ctx->d_topk_idx, nb * TK * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(fetch_nonfinite(ctx));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (*ctx->h_nonfinite && c->auto_fallback) {   // BH_FLAG_AUTO: those rows again, on the f32 kernels
            bool redone = false;
            float *d_emb = (emb_out || whole_slice) ? ctx->d_arena + ctx->t_off[m.h.embedding_tensor] : nullptr;
            const int rr = settle_rows(c, ctx, 0, nb, d_emb, &redone);
            if (rr != BH_OK) return rr;
            clear_nonfinite(ctx);
            if (redone && logits_out) HIPCHK(hipMemcpy(logits_out + b0 * NC, ctx->d_logits, nb * NC * sizeof(float), hipMemcpyDeviceToHost));
            if (redone && emb_out)
                HIPCHK(hipMemcpy(emb_out + b0 * m.h.embedding_dim, ctx->d_arena + ctx->t_off[m.h.embedding_tensor],
                                 nb * (size_t)m.h.embedding_dim * sizeof(float), hipMemcpyDeviceToHost));
        }
        if (out)
            for (size_t i = 0; i < nb; i++) {
                bh_result &r = out[b0 + i];
                r.n_pred = 0;
                for (uint32_t k = 0; k < c->top_k; k++) {
                    const int32_t id = ctx->h_topk_idx[i * c->top_k + k];
                    if (id < 0) break;
                    r.index[r.n_pred] = id;
                    r.confidence[r.n_pred] = ctx->h_topk_conf[i * c->top_k + k];
                    r.n_pred++;
                }
            }
        const int nf = nonfinite_status(c, ctx);   // the rows are filled (marked rows come out empty); the call still fails
        if (nf != BH_OK) return nf;
    }
    return BH_OK;
}

// IEEE binary16 round-to-nearest-even of a float, and back (host side of the hi/lo operand split)
uint16_t f32_to_f16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));   // overflow / nan
    if (x < 0x38800000u) {                                                                     // subnormal half
        if (x < 0x33000000u) return (uint16_t)sign;
        const int shift = 113 - (int)(x >> 23);
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const uint32_t half = m >> (shift + 13), rem = m & ((1u << (shift + 13)) - 1), mid = 1u << (shift + 12);
        return (uint16_t)(sign | (half + ((rem > mid || (rem == mid && (half & 1))) ? 1 : 0)));
    }
    const uint32_t e = ((x >> 23) - 112) << 10, m = (x >> 13) & 0x3ffu, rem = x & 0x1fffu;
    uint32_t h = e | m;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
    return (uint16_t)(sign | h);
}
float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { int k = 0; uint32_t mm = m; while (!(mm & 0x400u)) { mm <<= 1; k++; } x = sign | ((uint32_t)(113 - k) << 23) | ((mm & 0x3ffu) << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &x, 4);
    return f;
}

// The block that starts at layer i as a fused launch: expand(1x1, or the stem conv) -> depthwise -> project(1x1) whose
// intermediates have no other reader, described and handed to the tile planner.  Host logic only (no device): plan_fusion uses
// it per block at create, bh_plan_fused_blocks walks a model file with it.
bool describe_fused_block(const bh::Model &m, const std::vector<int> &readers, size_t i, int precision, int force_cfg, bh::MbDesc &d) {
    const size_t nl = m.layers.size();
    if (i + 1 < nl && m.layers[i].op == bh::OP_DWCONV && m.layers[i + 1].op == bh::OP_PWCONV) {
        // depthwise -> project (+ residual) WITHOUT an expand convolution (the expand-ratio-1 blocks of EfficientNet after the
        // first): fused with the block input standing in for the expanded tensor (MbDesc::noexp)
        const auto &D = m.layers[i], &P = m.layers[i + 1];
        if (P.in_tensor != i + 1 || readers[i + 1] != 1 || D.res_tensor != bh::NO_TENSOR) return false;
        if (D.kh != D.kw || D.sh != D.sw || D.cout != P.cin || D.in_layout != 0) return false;
        d = bh::MbDesc{};
        d.noexp = 1;
        d.H = (int)D.in_h; d.W = (int)D.in_w; d.Cin = (int)D.cout; d.Cexp = (int)D.cout; d.Cout = (int)P.cout;
        d.Ho = (int)D.out_h; d.Wo = (int)D.out_w; d.pad_t = (int)D.pad_t; d.pad_l = (int)D.pad_l;
        d.KS = (int)D.kh; d.ST = (int)D.sh;
        d.act_e = (int)D.act; d.act_d = (int)D.act; d.act_p = (int)P.act;   // (no expand activation: the kernel template is keyed on one)
        if (const char *dbg = getenv("BIRDA_HIP_MB_DBG")) d.dbg = atoi(dbg);
        d.prec = precision;
        if (!bh::mb_plan(d, force_cfg)) {
            if (d.prec == 0) return false;
            d.prec = 0;
            if (!bh::mb_plan(d, force_cfg)) return false;
        }
        return true;
    }
    if (i + 2 >= nl) return false;
    const auto &E = m.layers[i], &D = m.layers[i + 1], &P = m.layers[i + 2];
    const bool stem = E.op == bh::OP_CONV && E.in_layout == 1 && E.kh == E.kw && E.sh == E.sw && E.in_tensor == 0;
    if ((E.op != bh::OP_PWCONV && !stem) || D.op != bh::OP_DWCONV || P.op != bh::OP_PWCONV) return false;
    if (D.in_tensor != i + 1 || P.in_tensor != i + 2 || readers[i + 1] != 1 || readers[i + 2] != 1) return false;
    if (E.res_tensor != bh::NO_TENSOR || D.res_tensor != bh::NO_TENSOR) return false;
    if (D.kh != D.kw || D.sh != D.sw || E.cout != D.cout || D.cout != P.cin) return false;
    d = bh::MbDesc{};
    d.H = (int)E.in_h; d.W = (int)E.in_w; d.Cin = (int)E.cin; d.Cexp = (int)E.cout; d.Cout = (int)P.cout;
    if (stem) {  // the depthwise conv sees the stem's output image; the stem itself is gathered
        d.stem = 1; d.stem_c = (int)E.cin; d.stem_h = (int)E.in_h; d.stem_w = (int)E.in_w; d.stem_k = (int)E.kh;
        d.stem_s = (int)E.sh; d.stem_pt = (int)E.pad_t; d.stem_pl = (int)E.pad_l;
        d.H = (int)E.out_h; d.W = (int)E.out_w; d.Cin = (int)(E.kh * E.kw * E.cin);
    }
    d.Ho = (int)D.out_h; d.Wo = (int)D.out_w; d.pad_t = (int)D.pad_t; d.pad_l = (int)D.pad_l;
    d.KS = (int)D.kh; d.ST = (int)D.sh;
    d.act_e = (int)E.act; d.act_d = (int)D.act; d.act_p = (int)P.act;
    if (const char *dbg = getenv("BIRDA_HIP_MB_DBG")) d.dbg = atoi(dbg);
    d.prec = precision;
    // The split-f16 MFMA and the f32 MFMA agree to ~1e-7 of sum|a b|; measured (profiles/), f16x3 is
    // the faster one on every block, the stem's 18-column im2col GEMM included.
    if (d.stem && precision == 3 && getenv("BIRDA_HIP_STEM_F32")) d.prec = 0;   // A/B aid
    if (!bh::mb_plan(d, force_cfg)) {
        if (d.prec == 0) return false;
        d.prec = 0;                       // no f16 instantiation for this shape: f32 one, if any
        if (!bh::mb_plan(d, force_cfg)) return false;
    }
    return true;
}

std::vector<int> tensor_readers(const bh::Model &m) {
    std::vector<int> readers(m.layers.size() + 1, 0);
    for (const auto &L : m.layers) {
        readers[L.in_tensor]++;
        if (L.res_tensor != bh::NO_TENSOR) readers[L.res_tensor]++;
    }
    readers[m.h.embedding_tensor]++;
    return readers;
}

// Prepares a fused launch for every block describe_fused_block accepts (weights re-laid fragment-major for the picked tile config).
int plan_fusion(bh_classifier *c) {
    const auto &m = c->model;
    const size_t nl = m.layers.size();
    c->fused_at.assign(nl, -1);
    const char *fuse_env = getenv("BIRDA_HIP_FUSE");
    if (fuse_env && fuse_env[0] == '0') return BH_OK;
    const char *cfg_env = getenv("BIRDA_HIP_MB_CFG");
    const int force_cfg = cfg_env ? atoi(cfg_env) : -1;
    const std::vector<int> readers = tensor_readers(m);
    for (size_t i = 0; i + 2 < nl; i++) {
        bh::MbDesc d{};
        if (!describe_fused_block(m, readers, i, c->precision, force_cfg, d)) continue;
        // (a no-expand block is layers i = depthwise, i + 1 = project; E then only lends the code below a valid layer to name)
        const auto &E = m.layers[i], &D = m.layers[d.noexp ? i : i + 1], &P = m.layers[d.noexp ? i + 1 : i + 2];
        // per-chunk weight blocks (kernels.hpp MbDesc)
        const int CE = d.CE, NTE = CE / 16, KG = d.KG, NTOP = d.NTOP, nch = d.nchunks, KK = d.KS * d.KS;
        const float *We = m.blob.data() + E.w_off, *Wp = m.blob.data() + P.w_off, *Wd = m.blob.data() + D.w_off;
        const float *be = m.blob.data() + E.b_off, *bd = m.blob.data() + D.b_off;
        const bool h16 = d.prec != 0;
        // f16 operand planes hold We * 2^se and Wp * 2^sp (kernels.hpp f16_scale_exponent); be / bp are multiplied alike
        int se = 0, sp = 0;
        if (h16) {
            float me = 0.0f, mp = 0.0f;
            for (size_t q = 0; !d.noexp && q < (size_t)d.Cin * d.Cexp; q++) me = std::max(me, std::fabs(We[q]));
            for (size_t q = 0; q < (size_t)d.Cexp * d.Cout; q++) mp = std::max(mp, std::fabs(Wp[q]));
            se = bh::f16_scale_exponent(me);
            sp = bh::f16_scale_exponent(mp);
        }
        // GELU blocks of the f16 modes: the expand GELU runs on the SCALED accumulator with coefficients c_k 2^(-k se) and the 2^-se
        // moves into the depthwise taps (kernels.hpp gelu_erf_fast4_scaled).  c_5 2^(-5 se) must stay a normal f32 on both sides:
        // |se| <= 21 (weights 2^7 away from the usual He-normal sizes still land within 2^-8 of the top of the f16 range).
        d.e_fold = 0;
        d.gelu = bh::GeluScaled{0.f, 0.f, 0.f, 0.f, 0.f};
#if BH_GELU_DEGREE == 5
        if (d.noexp) se = 0;
        if (h16 && d.act_e == bh::ACT_GELU_ERF && !d.noexp) {
            se = std::max(-21, std::min(21, se));
            d.e_fold = 1;
            float gc[5];
            for (int k = 1; k <= 5; k++) gc[k - 1] = std::ldexp(bh::kGeluCoef[k - 1], -k * se);
            d.gelu = bh::GeluScaled{gc[0], gc[1], gc[2], gc[3], gc[4]};
        }
#endif
        // ... and both GELUs of such a block leave TWICE their value (gelu2x_fast4, kernels.hpp): the expand one's factor joins the
        // 2^-se in the depthwise taps (x2e), the depthwise one's raises the exponent the project accumulators live at (x2d).
        int x2e = 0, x2d = 0;
#if BH_GELU_DEGREE == 5
        if (h16 && d.act_d == bh::ACT_GELU_ERF) x2d = 1;
        x2e = d.e_fold;
#endif
        const int spa = sp + x2d;   // the project accumulators hold 2^spa times the output
        d.e_unscale = std::ldexp(1.0f, -se); d.p_scale = std::ldexp(1.0f, spa); d.p_unscale = std::ldexp(1.0f, -spa);
        const size_t frag = h16 ? 512 : 256, psteps = h16 ? (CE + 31) / 32 : NTE;
        const bool p16 = h16 && CE == 16;   // project GEMM as one 16-deep step: [column tile]{hi, lo}[64 lanes][4 halves]
        const size_t we_fl = (size_t)KG * NTE * frag + CE, wp_fl = p16 ? (size_t)NTOP * 256 : psteps * NTOP * frag, wd_fl = (size_t)KK * CE + CE;
        std::vector<float> wef(nch * we_fl, 0.0f), wpf(nch * wp_fl, 0.0f), wdf(nch * wd_fl, 0.0f);
        // (stem block in the f16 modes: the kernel gathers its im2col columns by memory runs, kernels_mbconv.hip -- position
        //  8 kq + 3 q + dx of the one 32-deep step is tap (dy, dx) of channel ch with run 2 kq + q = 3 ch + dy)
        auto we_at = [&](int k, int n) {
            if (d.stem && h16) {
                const int kq = k >> 3, jj = k & 7, r = 2 * kq + jj / 3, dx = jj % 3;
                if (k >= 32 || jj >= 6 || r >= 3 * d.stem_c) return 0.0f;
                const int ch = r / 3, dy = r - 3 * ch;
                k = (dy * 3 + dx) * d.stem_c + ch;
            }
            return (!d.noexp && k < d.Cin && n < d.Cexp) ? std::ldexp(We[(size_t)k * d.Cexp + n], se) : 0.0f;
        };
        auto wp_at = [&](int k, int n) { return (k < d.Cexp && n < d.Cout) ? std::ldexp(Wp[(size_t)k * d.Cout + n], sp) : 0.0f; };
        // f16: element jj of lane's 8-half fragment = k = 32 g + 8 (lane >> 4) + jj; hi plane then lo plane
        auto put16 = [&](std::vector<float> &dst, size_t base_fl, int plane, int lane, int jj, float v) {
            uint16_t *h = reinterpret_cast<uint16_t *>(dst.data() + base_fl) + ((size_t)plane * 64 + lane) * 8 + jj;
            const uint16_t hi = f32_to_f16(v);
            *h = plane == 0 ? hi : f32_to_f16(v - f16_to_f32(hi));
        };
        for (int ch = 0; ch < nch; ch++) {
            for (int g = 0; g < KG; g++)
                for (int j = 0; j < NTE; j++)
                    for (int lane = 0; lane < 64; lane++) {
                        const int n = ch * CE + 16 * j + (lane & 15);
                        if (h16) {
                            for (int jj = 0; jj < 8; jj++) {
                                const float v = we_at(32 * g + 8 * (lane >> 4) + jj, n);
                                const size_t base = ch * we_fl + ((size_t)g * NTE + j) * 512;
                                put16(wef, base, 0, lane, jj, v);
                                put16(wef, base, 1, lane, jj, v);
                            }
                        } else {
                            for (int cc = 0; cc < 4; cc++)
                                wef[ch * we_fl + (((size_t)g * NTE + j) * 64 + lane) * 4 + cc] = we_at(16 * g + 4 * (lane >> 4) + cc, n);
                        }
                    }
            for (int n = 0; n < CE; n++) wef[ch * we_fl + (size_t)KG * NTE * frag + n] = (!d.noexp && ch * CE + n < d.Cexp) ? std::ldexp(be[ch * CE + n], se) : 0.0f;
            if (p16) {
                for (int j = 0; j < NTOP; j++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int jj = 0; jj < 4; jj++) {
                            const float v = wp_at(ch * CE + 4 * (lane >> 4) + jj, 16 * j + (lane & 15));
                            uint16_t *h = reinterpret_cast<uint16_t *>(wpf.data() + ch * wp_fl + (size_t)j * 256);
                            const uint16_t hi = f32_to_f16(v);
                            h[(size_t)lane * 4 + jj] = hi;
                            h[(size_t)(64 + lane) * 4 + jj] = f32_to_f16(v - f16_to_f32(hi));
                        }
            }
            for (int g = 0; g < (int)psteps && !p16; g++)
                for (int j = 0; j < NTOP; j++)
                    for (int lane = 0; lane < 64; lane++) {
                        const int n = 16 * j + (lane & 15);
                        if (h16) {
                            for (int jj = 0; jj < 8; jj++) {
                                const int kk = 32 * g + 8 * (lane >> 4) + jj;          // k inside the chunk (zero padding past CE)
                                const float v = kk < CE ? wp_at(ch * CE + kk, n) : 0.0f;
                                const size_t base = ch * wp_fl + ((size_t)g * NTOP + j) * 512;
                                put16(wpf, base, 0, lane, jj, v);
                                put16(wpf, base, 1, lane, jj, v);
                            }
                        } else {
                            for (int cc = 0; cc < 4; cc++)
                                wpf[ch * wp_fl + (((size_t)g * NTOP + j) * 64 + lane) * 4 + cc] = wp_at(ch * CE + 16 * g + 4 * (lane >> 4) + cc, n);
                        }
                    }
            for (int tap = 0; tap < KK; tap++)
                for (int n = 0; n < CE; n++)
                    wdf[ch * wd_fl + (size_t)tap * CE + n] = ch * CE + n < d.Cexp ? std::ldexp(Wd[(size_t)tap * d.Cexp + ch * CE + n], d.e_fold ? -se - x2e : 0) : 0.0f;
            for (int n = 0; n < CE; n++) wdf[ch * wd_fl + (size_t)KK * CE + n] = ch * CE + n < d.Cexp ? bd[ch * CE + n] : 0.0f;
        }
        float *dwe = nullptr, *dwp = nullptr, *dwd = nullptr;
        int rc = upload(wef.data(), wef.size() * sizeof(float), &dwe);
        if (rc != BH_OK) return rc;
        c->d_owned.push_back(dwe);
        rc = upload(wpf.data(), wpf.size() * sizeof(float), &dwp);
        if (rc != BH_OK) return rc;
        c->d_owned.push_back(dwp);
        rc = upload(wdf.data(), wdf.size() * sizeof(float), &dwd);
        if (rc != BH_OK) return rc;
        c->d_owned.push_back(dwd);
        d.We = dwe; d.Wp = dwp; d.Wd = dwd;
        d.bp = c->d_blob + P.b_off;
        if (spa != 0) {   // bp * 2^spa: the project accumulators start there
            std::vector<float> bps(d.Cout);
            for (int n = 0; n < d.Cout; n++) bps[n] = std::ldexp(m.blob[P.b_off + n], spa);
            float *dbp = nullptr;
            rc = upload(bps.data(), bps.size() * sizeof(float), &dbp);
            if (rc != BH_OK) return rc;
            c->d_owned.push_back(dbp);
            d.bp = dbp;
        }
        c->fused_at[i] = (int)c->mb.size();
        c->mb.push_back(d);
        {
            bh::MbDesc tw{};
            tw.cfg = -1;
            static const bool no_twin = getenv("BIRDA_HIP_MB_TWIN") && getenv("BIRDA_HIP_MB_TWIN")[0] == '0';   // (A/B aid)
            if (force_cfg < 0 && !no_twin && bh::mb_plan_twin(d, tw)) c->mb_small.push_back(tw);
            else { tw.cfg = -1; c->mb_small.push_back(tw); }
        }
        i += d.noexp ? 1 : 2;
    }
    return BH_OK;
}

int internal_ctx(bh_classifier *c, size_t n, bh_batch_context **out) {
    if (c->internal_ctx && c->internal_ctx->max_batch >= n) { *out = c->internal_ctx; return BH_OK; }
    if (c->internal_ctx) { ctx_destroy(c->internal_ctx); c->internal_ctx = nullptr; }
    int rc = ctx_create(c, n, false, &c->internal_ctx);
    *out = c->internal_ctx;
    return rc;
}

}  // namespace

extern "C" {

int bh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *bh_backend_name(void) { return "HIP (gfx950)"; }

// ---- provider arm + default batch size (reference classifier.rs:662-1089, lib.rs:256-288) ---------------
static void fill_device_facts(int32_t ordinal, bh_provider_status *o) {
    hipDeviceProp_t prop;
    if (ordinal < 0 || hipGetDeviceProperties(&prop, ordinal) != hipSuccess) { (void)hipGetLastError(); return; }
    snprintf(o->device_name, sizeof o->device_name, "%s", prop.name);
    snprintf(o->arch, sizeof o->arch, "%s", prop.gcnArchName);
    o->compute_units = (uint32_t)prop.multiProcessorCount;
    o->hbm_bytes = (uint64_t)prop.totalGlobalMem;
}

int bh_select_provider(const char *requested, int32_t device_ordinal, bh_provider_status *out) try {
    if (!requested || !out) return fail(BH_ERR_INVALID, "select_provider: null argument");
    memset(out, 0, sizeof *out);
    std::string req(requested);
    for (char &ch : req) if (ch >= 'A' && ch <= 'Z') ch = (char)(ch + 32);
    snprintf(out->requested, sizeof out->requested, "%s", req.c_str());
    const int ndev = bh_device_count();
    out->device_count = (uint32_t)std::max(ndev, 0);
    out->device = -1;
    auto use_hip = [&]() {
        out->device = device_ordinal < 0 ? 0 : device_ordinal;
        snprintf(out->actual, sizeof out->actual, "HIP");
        fill_device_facts(out->device, out);
    };
    if (req == "cpu") { snprintf(out->actual, sizeof out->actual, "CPU"); return BH_OK; }
    if (req == "auto" || req == "gpu") {
        if (ndev > 0 && device_ordinal < ndev) { use_hip(); return BH_OK; }
        snprintf(out->actual, sizeof out->actual, "CPU");
        snprintf(out->fallback_reason, sizeof out->fallback_reason, "No GPU providers available");
        return BH_OK;
    }
    if (req == "hip" || req == "rocm") {
        if (ndev <= 0) return fail(BH_ERR_NO_DEVICE, "HIP provider requested but no HIP device is available");
        if (device_ordinal >= ndev) return fail(BH_ERR_NO_DEVICE, "device %d out of range (0..%d)", device_ordinal, ndev - 1);
        use_hip();
        return BH_OK;
    }
    return fail(BH_ERR_INVALID, "select_provider: '%s' is not served by this backend (auto, gpu, hip, rocm, cpu)", requested);
} catch (...) { return on_exception(); }

int bh_classifier_provider_status(const bh_classifier *c, bh_provider_status *out) try {
    if (!c || !out) return fail(BH_ERR_INVALID, "provider_status: null argument");
    memset(out, 0, sizeof *out);
    snprintf(out->requested, sizeof out->requested, "hip");
    snprintf(out->actual, sizeof out->actual, "HIP");
    out->device = c->device;
    out->device_count = (uint32_t)std::max(bh_device_count(), 0);
    fill_device_facts(c->device, out);
    // BH_FLAG_AUTO degraded some rows to the f32 kernels: recorded as the reference records a provider fall-back (classifier.rs:742-754)
    if (const unsigned long long nfb = c->fallback_segments.load())
        snprintf(out->fallback_reason, sizeof out->fallback_reason,
                 "%llu segment(s) re-run on the f32 kernels: an activation left the f16 operand range of the split-f16 path", nfb);
    return BH_OK;
} catch (...) { return on_exception(); }

uint64_t bh_classifier_fallback_segments(const bh_classifier *c) { return c ? (uint64_t)c->fallback_segments.load() : 0; }

size_t bh_default_batch_size(uint32_t model_type, const char *provider_actual) {
    const char *p = provider_actual ? provider_actual : "HIP";
    if (!strcmp(p, "CPU")) return 8;                                                                  // batch_size::CPU
    if (!strcmp(p, "CUDA")) return (model_type == BH_MODEL_BIRDNET_V24 || model_type == BH_MODEL_BSG_FINLAND) ? 64 : 32;
    if (!strcmp(p, "TensorRT")) return 32;
    if (!strcmp(p, "HIP")) return 256;
    return 16;                                                                                        // batch_size::OTHER_GPU
}

size_t bh_classifier_default_batch_size(const bh_classifier *c) {
    return c ? bh_default_batch_size(c->model.h.family, "HIP") : 0;
}
const char *bh_last_error(void) { return g_err.c_str(); }

int bh_classifier_create(const bh_config *cfg, bh_classifier **out) try {
    if (!cfg || !out || !cfg->model_path) return fail(BH_ERR_INVALID, "classifier_create: null config/model_path");
    *out = nullptr;
    if (cfg->top_k == 0 || cfg->top_k > BH_MAX_TOP_K) return fail(BH_ERR_INVALID, "top_k must be 1..%d", BH_MAX_TOP_K);
    auto c = std::make_unique<bh_classifier>();
    std::string err;
    if (!bh::load_model(cfg->model_path, c->model, err)) return fail(BH_ERR_IO, "%s", err.c_str());
    const auto &m = c->model;
    if (cfg->labels_path) {
        int rc = read_labels(cfg->labels_path, c->labels);
        if (rc != BH_OK) return rc;
        if (c->labels.size() != m.h.n_classes)
            return fail(BH_ERR_LABELS, "label count %zu does not match model output width %u", c->labels.size(), m.h.n_classes);
    }
    int ndev = bh_device_count();
    if (ndev <= 0) return fail(BH_ERR_NO_DEVICE, "no HIP device available (libbirda_hip has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(BH_ERR_NO_DEVICE, "device %d out of range (0..%d)", cfg->device, ndev - 1);
    c->device = cfg->device;
    c->top_k = cfg->top_k;
    c->min_conf = cfg->min_confidence;
    HIPCHK(hipSetDevice(c->device));
    if (m.h.n_branches > bh::MAX_BRANCHES) return fail(BH_ERR_UNSUPPORTED, "too many front-end branches");
    // front-end operators
    // GEMM operand precision (decided here: the front-end operator layout depends on it)
    // the spectrogram front-end keeps f32-grade products in every mode: f32 MFMA, or split f16 in the f16 modes
    {
        const uint32_t pf = cfg->flags & BH_FLAG_PRECISION_MASK;
        c->precision = pf == BH_FLAG_F32 ? 0 : pf == BH_FLAG_F16 ? 1 : 3;
        c->auto_fallback = pf == BH_FLAG_AUTO;
        c->model_path = cfg->model_path;
    }
    if (const char *pe = (cfg->flags & FLAG_INTERNAL_NO_ENV) ? nullptr : getenv("BIRDA_HIP_PRECISION")) {
        if (!strcmp(pe, "f32")) { c->precision = 0; c->auto_fallback = false; }
        else if (!strcmp(pe, "f16x3")) { c->precision = 3; c->auto_fallback = false; }
        else if (!strcmp(pe, "auto")) { c->precision = 3; c->auto_fallback = true; }
        else if (!strcmp(pe, "f16")) { c->precision = 1; c->auto_fallback = false; }
        else return fail(BH_ERR_INVALID, "BIRDA_HIP_PRECISION must be auto, f32, f16x3 or f16");
    }
    // f16x3 and f16: the front-end GEMM on the split-f16 MFMA (f32-grade products, faster than the f32 MFMA;
    // BIRDA_HIP_MEL_F32=1 keeps it on the f32 MFMA: A/B aid)
    int fe_prec = (c->precision != 0 && !(getenv("BIRDA_HIP_MEL_F32") && getenv("BIRDA_HIP_MEL_F32")[0] == '1')) ? 3 : 0;
    for (uint32_t b = 0; b < m.h.n_branches; b++)
        if (m.branches[b].frame_length % 256) fe_prec = 0;   // 32-deep steps split over 4 waves
    if (fe_prec == 3) {
        // mel32_kernel (Y rows staged along k: no LDS bank conflicts whatever the hop) where mel_kernel's frame-strided
        // reads collapse onto a few banks: hops that are multiples of 16 samples (Perch: 320 -> every frame on one bank;
        // 483 -> 258 us per 600 segments).  BirdNET's hops (278 / 280) stay on mel_kernel (790 vs 1006 us per 1000).
        // BIRDA_HIP_MEL32=0/1 forces the choice (A/B aid).
        bool want = false, can = true;
        for (uint32_t b = 0; b < m.h.n_branches; b++) {
            if (m.branches[b].frame_step % 16 == 0) want = true;
            if (align_up(m.branches[b].n_mels, 16) % 32 || m.branches[b].frame_length % 512) can = false;   // 32-mel tiles, 64-k chunks per wave
        }
        if (const char *e = getenv("BIRDA_HIP_MEL32")) want = e[0] == '1';
        if (want && can) fe_prec = 32;
    }
    if (m.h.sample_count % 4) return fail(BH_ERR_UNSUPPORTED, "front-end: sample_count %u must be a multiple of 4 (16-byte span loads)", m.h.sample_count);
    c->fe.prec = fe_prec;
    c->fe.n_branches = (int)m.h.n_branches;
    c->fe.sample_count = (int)m.h.sample_count;
    c->fe.norm_eps = m.h.norm_eps;
    for (uint32_t b = 0; b < m.h.n_branches; b++) {
        const auto &br = m.branches[b];
        const int nm_pad = (int)align_up(br.n_mels, 16);
        if (nm_pad != 32 && nm_pad != 96 && nm_pad != 128)
            return fail(BH_ERR_UNSUPPORTED, "front-end: n_mels %u not built (32/96/128)", br.n_mels);
        if (b > 0 && nm_pad != c->fe.br[0].nm_pad) return fail(BH_ERR_UNSUPPORTED, "front-end: branches differ in n_mels");
        if (br.frame_length % 128 || br.fft_length != br.frame_length)
            return fail(BH_ERR_UNSUPPORTED, "front-end: frame_length %u must be a multiple of 128 and equal fft_length", br.frame_length);
        if ((64 * br.frame_step) % 4) return fail(BH_ERR_UNSUPPORTED, "front-end: hop %u unsupported", br.frame_step);
        int gf_s = 0;
        std::vector<float> gf = build_gf(br, m.blob.data() + br.mel_w_off, nm_pad, fe_prec, &gf_s);
        float *d = nullptr;
        int rc = upload(gf.data(), gf.size() * sizeof(float), &d);
        if (rc != BH_OK) return rc;
        c->d_owned.push_back(d);
        auto &p = c->fe.br[b];
        p.gf = d; p.L = (int)br.frame_length; p.H = (int)br.frame_step; p.K = p.L / 2;
        p.n_mels = (int)br.n_mels; p.nm_pad = nm_pad; p.n_frames = (int)br.n_frames;
        p.expo = 1.0f / (1.0f + expf(br.mag_scale));
        p.log2_bias = -2.0f * (float)gf_s * p.expo;
        p.out_scale = br.out_scale; p.out_shift = br.out_shift; p.flip = (int)(br.flags & 1u);
        c->mel_flops += 2ull * (uint64_t)p.K * nm_pad * br.n_frames;
    }
    {
        float *d = nullptr;
        int rcf = upload(&c->fe, sizeof c->fe, &d);
        if (rcf != BH_OK) return rcf;
        c->d_fe = reinterpret_cast<bh::FrontendParams *>(d);
        c->d_owned.push_back(d);
    }
    // weights
    int rc = upload(m.blob.data(), m.blob.size() * sizeof(float), &c->d_blob);
    if (rc != BH_OK) return rc;
    c->d_w.resize(m.layers.size());
    c->ldw.assign(m.layers.size(), 0);
    for (size_t i = 0; i < m.layers.size(); i++) {
        const auto &L = m.layers[i];
        c->d_w[i] = c->d_blob + L.w_off;
        if (L.op == bh::OP_PWCONV || L.op == bh::OP_DENSE) {
            if (L.cin % 4) return fail(BH_ERR_UNSUPPORTED, "layer %zu: cin %u not a multiple of 4", i, L.cin);
            const int ld = (int)align_up(L.cout, 4);
            c->ldw[i] = ld;
            if (ld != (int)L.cout) {  // pad rows so 16-B loads stay aligned
                std::vector<float> w((size_t)L.cin * ld, 0.0f);
                for (uint32_t k = 0; k < L.cin; k++)
                    memcpy(&w[(size_t)k * ld], m.blob.data() + L.w_off + (size_t)k * L.cout, L.cout * sizeof(float));
                float *d = nullptr;
                rc = upload(w.data(), w.size() * sizeof(float), &d);
                if (rc != BH_OK) return rc;
                c->d_owned.push_back(d);
                c->d_w[i] = d;
            }
        } else if (L.op == bh::OP_DWCONV) {
            if (L.cout % 4 || L.kh != L.kw || L.sh != L.sw || !((L.kh == 3 || L.kh == 5) && (L.sh == 1 || L.sh == 2)))
                return fail(BH_ERR_UNSUPPORTED, "layer %zu: depthwise %ux%u stride %u channels %u not built", i, L.kh, L.kw, L.sh, L.cout);
        } else if (L.op == bh::OP_CONV) {
            if (L.cout % 8 || (size_t)L.kh * L.kw * L.cin * L.cout * 4 > 64 * 1024)
                return fail(BH_ERR_UNSUPPORTED, "layer %zu: direct conv shape not built", i);
        } else if (L.op == bh::OP_GAP || L.op == bh::OP_SCALE) {
            if (L.cout % 4) return fail(BH_ERR_UNSUPPORTED, "layer %zu: channels %u not a multiple of 4", i, L.cout);
        }
    }
    if (m.layers.empty() || m.layers.back().cout != m.h.n_classes)
        return fail(BH_ERR_IO, "model: last layer width != n_classes");
    rc = plan_fusion(c.get());
    if (rc != BH_OK) return rc;
    // f16 operand planes for the GEMM layers that stay outside the fused blocks (head conv, dense)
    c->d_w16.assign(m.layers.size(), nullptr);
    c->w16_unscale.assign(m.layers.size(), 1.0f);
    if (c->precision != 0) {
        std::vector<char> in_block(m.layers.size(), 0);
        for (size_t i = 0; i < m.layers.size(); i++)
            if (c->fused_at[i] >= 0) {
                in_block[i] = in_block[i + 1] = 1;
                if (!c->mb[c->fused_at[i]].noexp) in_block[i + 2] = 1;
            }
        for (size_t i = 0; i < m.layers.size(); i++) {
            const auto &L = m.layers[i];
            if (in_block[i] || (L.op != bh::OP_PWCONV && L.op != bh::OP_DENSE) || !bh::pw_gemm16_supports((int)L.cin, (int)L.act)) continue;
            const int K = (int)L.cin, N = (int)L.cout, nt = (N + 15) / 16;
            std::vector<uint16_t> planes((size_t)(K / 32) * nt * 2 * 64 * 8, 0);
            const float *W = m.blob.data() + L.w_off;
            float wmax = 0.0f;
            for (size_t q = 0; q < (size_t)K * N; q++) wmax = std::max(wmax, std::fabs(W[q]));
            const int ws = bh::f16_scale_exponent(wmax);
            c->w16_unscale[i] = std::ldexp(1.0f, -ws);
            for (int st = 0; st < K / 32; st++)
                for (int t = 0; t < nt; t++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int jj = 0; jj < 8; jj++) {
                            const int k = 32 * st + 8 * (lane >> 4) + jj, n = 16 * t + (lane & 15);
                            const float v = n < N ? std::ldexp(W[(size_t)k * N + n], ws) : 0.0f;
                            const uint16_t hi = f32_to_f16(v);
                            const size_t base = (((size_t)st * nt + t) * 2) * 64 * 8;
                            planes[base + (size_t)lane * 8 + jj] = hi;
                            planes[base + 64 * 8 + (size_t)lane * 8 + jj] = f32_to_f16(v - f16_to_f32(hi));
                        }
            float *d = nullptr;
            rc = upload(planes.data(), planes.size() * sizeof(uint16_t), &d);
            if (rc != BH_OK) return rc;
            c->d_owned.push_back(d);
            c->d_w16[i] = d;
        }
    }
    // head conv + GELU followed by the global average pool (and read by nothing else): one launch
    c->head_gap.assign(m.layers.size(), 0);
    {
        const char *hg = getenv("BIRDA_HIP_HEAD_GAP");
        for (size_t i = 0; i + 1 < m.layers.size() && !(hg && hg[0] == '0'); i++) {
            const auto &L = m.layers[i], &G = m.layers[i + 1];
            if (!c->d_w16[i] || L.op != bh::OP_PWCONV || L.res_tensor != bh::NO_TENSOR || G.op != bh::OP_GAP ||
                G.in_tensor != i + 1 || !bh::head_gap16_supports((int)(L.out_h * L.out_w), (int)L.cin, (int)L.cout, (int)L.act))
                continue;
            bool other_reader = m.h.embedding_tensor == i + 1;
            for (size_t j = 0; j < m.layers.size(); j++)
                if (j != i + 1 && (m.layers[j].in_tensor == i + 1 || m.layers[j].res_tensor == i + 1)) other_reader = true;
            if (!other_reader) c->head_gap[i] = 1;
        }
    }
    if (const char *st = getenv("BIRDA_HIP_MB_STAMPS"); st && st[0] == '1' && !c->mb.empty()) {
        HIPCHK(hipMalloc((void **)&c->d_stamps, c->mb.size() * 8 * sizeof(unsigned long long)));
        HIPCHK(hipMemset(c->d_stamps, 0, c->mb.size() * 8 * sizeof(unsigned long long)));
        for (size_t i = 0; i < c->mb.size(); i++) { c->mb[i].stamps = c->d_stamps + i * 8; c->mb_small[i].stamps = c->mb[i].stamps; }
    }
    *out = c.release();
    return BH_OK;
} catch (...) { return on_exception(); }

void bh_classifier_destroy(bh_classifier *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->fb) { bh_classifier_destroy(c->fb); c->fb = nullptr; }
    if (c->internal_ctx) ctx_destroy(c->internal_ctx);
    for (bh_batch_context *p : c->parked_ctx)
        if (p) ctx_destroy(p);
    for (float *d : c->d_owned) (void)hipFree(d);
    (void)hipFree(c->d_blob);
    (void)hipFree(c->d_stamps);
    (void)hipFree(c->d_class_score);
    (void)hipFree(c->d_species_keep);
    (void)hipFree(c->d_bsg);
    delete c;
}

// ---- range filter / species list (reference classifier.rs:587-645) -------------------------
int bh_classifier_set_range_filter(bh_classifier *c, const float *scores, size_t n_classes, float threshold,
                                   int keep_unmatched, int rerank) try {
    if (!c || !scores) return fail(BH_ERR_INVALID, "set_range_filter: null argument");
    if (n_classes != c->model.h.n_classes)
        return fail(BH_ERR_INVALID, "set_range_filter: %zu scores for %u classes", n_classes, c->model.h.n_classes);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());   // no launch may still read the table being replaced
    if (!c->d_class_score) HIPCHK(hipMalloc((void **)&c->d_class_score, n_classes * sizeof(float)));
    HIPCHK(hipMemcpy(c->d_class_score, scores, n_classes * sizeof(float), hipMemcpyHostToDevice));
    c->filter.class_score = c->d_class_score;
    c->filter.threshold = threshold;
    c->filter.keep_unmatched = keep_unmatched ? 1 : 0;
    c->filter.rerank = rerank ? 1 : 0;
    return BH_OK;
} catch (...) { return on_exception(); }

int bh_classifier_set_species_list(bh_classifier *c, const uint8_t *keep, size_t n_classes) try {
    if (!c || !keep) return fail(BH_ERR_INVALID, "set_species_list: null argument");
    if (n_classes != c->model.h.n_classes)
        return fail(BH_ERR_INVALID, "set_species_list: %zu flags for %u classes", n_classes, c->model.h.n_classes);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    if (!c->d_species_keep) HIPCHK(hipMalloc((void **)&c->d_species_keep, n_classes));
    HIPCHK(hipMemcpy(c->d_species_keep, keep, n_classes, hipMemcpyHostToDevice));
    c->filter.species_keep = c->d_species_keep;
    return BH_OK;
} catch (...) { return on_exception(); }

int bh_topk_from_logits(bh_classifier *c, const float *logits, size_t n, bh_result *out) try {
    if (!c || (n && (!logits || !out))) return fail(BH_ERR_INVALID, "topk_from_logits: null argument");
    if (!n) return BH_OK;
    HIPCHK(hipSetDevice(c->device));
    const size_t nc = c->model.h.n_classes, tk = c->top_k;
    float *d_l = nullptr, *d_c = nullptr;
    int32_t *d_i = nullptr;
    std::vector<int32_t> hi(n * tk);
    std::vector<float> hc(n * tk);
    hipError_t e = hipMalloc((void **)&d_l, n * nc * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&d_i, n * tk * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&d_c, n * tk * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d_l, logits, n * nc * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        bh::launch_topk(d_l, (int)n, (int)nc, (int)c->model.h.output_activation, (int)tk, c->min_conf, c->filter, d_i, d_c, nullptr, nullptr, nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(hi.data(), d_i, n * tk * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(hc.data(), d_c, n * tk * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_l); (void)hipFree(d_i); (void)hipFree(d_c);
    if (e != hipSuccess) return fail(BH_ERR_HIP, "topk_from_logits: %s", hipGetErrorString(e));
    for (size_t i = 0; i < n; i++) {
        bh_result &r = out[i];
        r.n_pred = 0;
        for (size_t k = 0; k < tk; k++) {
            if (hi[i * tk + k] < 0) break;
            r.index[r.n_pred] = hi[i * tk + k];
            r.confidence[r.n_pred] = hc[i * tk + k];
            r.n_pred++;
        }
    }
    return BH_OK;
} catch (...) { return on_exception(); }

int bh_classifier_clear_filters(bh_classifier *c) try {
    if (!c) return fail(BH_ERR_INVALID, "clear_filters: null classifier");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    {   // the range filter / species list only: BSG tables have their own clear
        bh::TopkFilter f{};
        f.bsg_intercept = c->filter.bsg_intercept; f.bsg_slope = c->filter.bsg_slope; f.bsg_prior = c->filter.bsg_prior;
        c->filter = f;
    }
    return BH_OK;
} catch (...) { return on_exception(); }

// ---- BSG post-processing tables (reference classifier.rs:508-545) ------------------------------------------
int bh_classifier_set_bsg(bh_classifier *c, const float *intercept, const float *slope, const float *prior, size_t n_classes) try {
    if (!c || !intercept || !slope) return fail(BH_ERR_INVALID, "set_bsg: null argument");
    if (n_classes != c->model.h.n_classes) return fail(BH_ERR_INVALID, "set_bsg: %zu entries for %u classes", n_classes, c->model.h.n_classes);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    if (!c->d_bsg) HIPCHK(hipMalloc((void **)&c->d_bsg, 3 * n_classes * sizeof(float)));
    HIPCHK(hipMemcpy(c->d_bsg, intercept, n_classes * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_bsg + n_classes, slope, n_classes * sizeof(float), hipMemcpyHostToDevice));
    if (prior) HIPCHK(hipMemcpy(c->d_bsg + 2 * n_classes, prior, n_classes * sizeof(float), hipMemcpyHostToDevice));
    c->filter.bsg_intercept = c->d_bsg;
    c->filter.bsg_slope = c->d_bsg + n_classes;
    c->filter.bsg_prior = prior ? c->d_bsg + 2 * n_classes : nullptr;
    return BH_OK;
} catch (...) { return on_exception(); }

int bh_classifier_clear_bsg(bh_classifier *c) try {
    if (!c) return fail(BH_ERR_INVALID, "clear_bsg: null classifier");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    c->filter.bsg_intercept = c->filter.bsg_slope = c->filter.bsg_prior = nullptr;
    return BH_OK;
} catch (...) { return on_exception(); }

int bh_classifier_info(const bh_classifier *c, bh_model_info *info) {
    if (!c || !info) return fail(BH_ERR_INVALID, "classifier_info: null argument");
    const auto &h = c->model.h;
    info->sample_rate = h.sample_rate; info->segment_duration = h.segment_duration; info->sample_count = h.sample_count;
    info->n_classes = h.n_classes; info->embedding_dim = h.embedding_dim; info->output_activation = h.output_activation;
    info->spec_channels = h.n_branches; info->spec_h = h.spec_h; info->spec_w = h.spec_w;
    info->n_layers = h.n_layers; info->macs_per_segment = c->model.macs_per_segment();
    info->mel_flops_per_segment = c->mel_flops;
    info->model_type = h.family;
    info->precision = c->auto_fallback ? BH_FLAG_AUTO : c->precision == 3 ? BH_FLAG_F16X3 : c->precision == 1 ? BH_FLAG_F16 : BH_FLAG_F32;
    return BH_OK;
}

const char *bh_classifier_label(const bh_classifier *c, uint32_t index) {
    if (!c || index >= c->labels.size()) return nullptr;
    return c->labels[index].c_str();
}

int bh_classifier_is_warm(const bh_classifier *c, size_t batch_size) {
    if (!c) return 0;
    auto *cc = const_cast<bh_classifier *>(c);
    std::lock_guard<std::mutex> g(cc->warm_mu);
    return cc->warmed.count(batch_size) ? 1 : 0;
}

int bh_classifier_ensure_warm(bh_classifier *c, size_t batch_size) try {
    if (!c || batch_size == 0) return fail(BH_ERR_INVALID, "ensure_warm: bad arguments");
    if (bh_classifier_is_warm(c, batch_size)) return BH_OK;
    // warmup(batch_size): all-zero segments through the real path (classifier.rs:443-466)
    std::vector<float> zero(c->model.h.sample_count, 0.0f);
    std::vector<const float *> segs(batch_size, zero.data());
    std::vector<bh_result> res(batch_size);
    bool had_ctx;
    { std::lock_guard<std::mutex> g(c->internal_mu); had_ctx = c->internal_ctx != nullptr; }
    int rc = bh_predict_batch(c, segs.data(), batch_size, zero.size(), res.data());
    if (!had_ctx) {
        // the warm-up's context is not kept: callers that batch create their own (process_file does, right after this
        // call), and a second full-size arena for the classifier's lifetime would only double the memory
        std::lock_guard<std::mutex> g(c->internal_mu);
        if (c->internal_ctx) { ctx_destroy(c->internal_ctx); c->internal_ctx = nullptr; }
    }
    if (rc != BH_OK) return rc;  // recorded only after success (classifier.rs:424)
    std::lock_guard<std::mutex> g(c->warm_mu);
    c->warmed.insert(batch_size);
    return BH_OK;
} catch (...) { return on_exception(); }

int bh_batch_context_create(bh_classifier *c, size_t max_batch, bh_batch_context **out) try {
    const char *keep = getenv("BIRDA_HIP_KEEP_TENSORS");
    const bool keep_t = keep && keep[0] == '1';
    if (c && out && !keep_t) {
        std::lock_guard<std::mutex> g(c->parked_mu);
        // a parked context of this size, or failing that one up to twice as large (files of slightly different lengths cap their
        // effective batch differently, processor.rs:531-545: an exact match alone would rebuild 4 GB of arena for each of them)
        bh_batch_context **pick = nullptr;
        for (bh_batch_context *&p : c->parked_ctx)
            if (p && !p->keep_tensors && p->max_batch == max_batch) { pick = &p; break; }
        if (!pick)
            for (bh_batch_context *&p : c->parked_ctx)
                if (p && !p->keep_tensors && p->max_batch > max_batch && p->max_batch <= 2 * max_batch && (!pick || p->max_batch < (*pick)->max_batch)) pick = &p;
        if (pick) {
            *out = *pick;
            *pick = nullptr;
            (*out)->asked_batch = max_batch;
            return BH_OK;
        }
    }
    return ctx_create(c, max_batch, keep_t, out);
} catch (...) { return on_exception(); }

size_t bh_classifier_trim(bh_classifier *c) {
    if (!c) return 0;
    size_t freed = 0;
    bh_batch_context *gone[bh_classifier::N_PARKED + 1] = {};
    int n = 0;
    {
        std::lock_guard<std::mutex> g(c->parked_mu);
        for (bh_batch_context *&p : c->parked_ctx)
            if (p) { gone[n++] = p; p = nullptr; }
    }
    {
        std::lock_guard<std::mutex> g(c->internal_mu);
        if (c->internal_ctx) { gone[n++] = c->internal_ctx; c->internal_ctx = nullptr; }
    }
    (void)hipSetDevice(c->device);
    for (int i = 0; i < n; i++) { freed += gone[i]->device_bytes; ctx_destroy(gone[i]); }
    return freed;
}

void bh_batch_context_destroy(bh_batch_context *ctx) {
    if (!ctx) return;
    bh_classifier *c = ctx->c;
    if (!ctx->keep_tensors && !ctx->profiling) {   // parked for the next create of this size (see bh_classifier::parked_ctx)
        ctx->forced_sub_slices = 0;
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(ctx->stream);
        *ctx->h_nonfinite = 0;
        (void)hipMemsetAsync(ctx->d_nonfinite, 0, sizeof(unsigned), ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
        bh_batch_context *old = nullptr;
        {
            std::lock_guard<std::mutex> g(c->parked_mu);
            bh_batch_context **slot = nullptr;
            for (bh_batch_context *&p : c->parked_ctx)
                if (!p) { slot = &p; break; }
            if (!slot) {      // all taken: one of a different size goes, else the oldest (first) one; the newcomer is the youngest (last)
                int gone = 0;
                for (int i = 0; i < bh_classifier::N_PARKED; i++)
                    if (c->parked_ctx[i]->max_batch != ctx->max_batch) { gone = i; break; }
                old = c->parked_ctx[gone];
                for (int i = gone; i + 1 < bh_classifier::N_PARKED; i++) c->parked_ctx[i] = c->parked_ctx[i + 1];
                slot = &c->parked_ctx[bh_classifier::N_PARKED - 1];
            }
            *slot = ctx;
        }
        if (old) ctx_destroy(old);
        return;
    }
    ctx_destroy(ctx);
}
size_t bh_batch_context_bytes(const bh_batch_context *ctx) {
    return ctx ? ctx->max_batch * (size_t)ctx->c->model.h.sample_count * sizeof(float) : 0;
}
size_t bh_batch_context_device_bytes(const bh_batch_context *ctx) { return ctx ? ctx->device_bytes : 0; }
int bh_batch_context_set_sub_slices(bh_batch_context *ctx, uint32_t n) {
    if (!ctx) return fail(BH_ERR_INVALID, "null batch context");
    ctx->forced_sub_slices = n;
    return BH_OK;
}
void *bh_batch_context_host_buffer(bh_batch_context *ctx, size_t *bytes) {
    if (bytes) *bytes = ctx ? bh_batch_context_bytes(ctx) : 0;
    return ctx ? ctx->h_input : nullptr;
}

int bh_predict(bh_classifier *c, const float *segment, size_t n_samples, bh_result *out) try {
    const float *segs[1] = {segment};
    return bh_predict_batch(c, segs, 1, n_samples, out);
} catch (...) { return on_exception(); }

int bh_predict_batch(bh_classifier *c, const float *const *segments, size_t n, size_t n_samples, bh_result *out) try {
    if (!c || !segments || !out) return fail(BH_ERR_INVALID, "predict_batch: null argument");
    if (n == 0) return BH_OK;
    if (n_samples != c->model.h.sample_count)
        return fail(BH_ERR_INVALID, "segment has %zu samples, model expects %u", n_samples, c->model.h.sample_count);
    std::lock_guard<std::mutex> g(c->internal_mu);
    bh_batch_context *ctx = nullptr;
    int rc = internal_ctx(c, n, &ctx);
    if (rc != BH_OK) return rc;
    return predict_slices(c, ctx, segments, nullptr, n, out, nullptr, nullptr);
} catch (...) { return on_exception(); }

int bh_predict_batch_with_context(bh_classifier *c, bh_batch_context *ctx, const float *const *segments, size_t n,
                                  size_t n_samples, bh_result *out) try {
    int rc = check_ctx(c, ctx);
    if (rc != BH_OK) return rc;
    if (!segments || !out) return fail(BH_ERR_INVALID, "predict_batch_with_context: null argument");
    if (n > ctx->asked_batch) return fail(BH_ERR_INVALID, "batch of %zu exceeds context capacity %zu", n, ctx->asked_batch);
    if (n_samples != c->model.h.sample_count)
        return fail(BH_ERR_INVALID, "segment has %zu samples, model expects %u", n_samples, c->model.h.sample_count);
    return predict_slices(c, ctx, segments, nullptr, n, out, nullptr, nullptr);
} catch (...) { return on_exception(); }

int bh_host_alloc(size_t bytes, void **out) try {
    if (!out || bytes == 0) return fail(BH_ERR_INVALID, "host_alloc: null argument or zero size");
    *out = nullptr;
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fail(BH_ERR_HIP, "hipHostMalloc of %zu bytes failed", bytes); }
    return BH_OK;
} catch (...) { return on_exception(); }

void bh_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int bh_host_register(void *p, size_t bytes) try {
    if (!p || bytes == 0) return fail(BH_ERR_INVALID, "host_register: null argument or zero size");
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return fail(BH_ERR_HIP, "hipHostRegister of %zu bytes failed", bytes); }
    return BH_OK;
} catch (...) { return on_exception(); }

int bh_host_unregister(void *p) try {
    if (!p) return BH_OK;
    if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return fail(BH_ERR_HIP, "hipHostUnregister failed"); }
    return BH_OK;
} catch (...) { return on_exception(); }

int bh_predict_batch_contig(bh_classifier *c, bh_batch_context *ctx, const float *base, size_t n, bh_result *out) try {
    int rc = check_ctx(c, ctx);
    if (rc != BH_OK) return rc;
    if (!base || !out) return fail(BH_ERR_INVALID, "predict_batch_contig: null argument");
    return predict_slices(c, ctx, nullptr, base, n, out, nullptr, nullptr);
} catch (...) { return on_exception(); }

int bh_predict_batch_logits(bh_classifier *c, bh_batch_context *ctx, const float *base, size_t n, float *logits,
                            float *embeddings) try {
    int rc = check_ctx(c, ctx);
    if (rc != BH_OK) return rc;
    if (!base || !logits) return fail(BH_ERR_INVALID, "predict_batch_logits: null argument");
    return predict_slices(c, ctx, nullptr, base, n, nullptr, logits, embeddings);
} catch (...) { return on_exception(); }

int bh_forward_device(bh_classifier *c, bh_batch_context *ctx, const float *d_segments, size_t n, float *d_logits,
                      int32_t *d_topk_index, float *d_topk_conf) try {
    int rc = check_ctx(c, ctx);
    if (rc != BH_OK) return rc;
    if (!d_segments || !d_logits) return fail(BH_ERR_INVALID, "forward_device: null device pointer");
    HIPCHK(hipSetDevice(c->device));
    const auto &h = c->model.h;
    for (size_t b0 = 0; b0 < n; b0 += ctx->max_batch) {
        const size_t nb = std::min(ctx->max_batch, n - b0);
        rc = forward_slice(c, ctx, d_segments + b0 * h.sample_count, nb, d_logits + b0 * h.n_classes,
                           d_topk_index ? d_topk_index + b0 * c->top_k : nullptr,
                           d_topk_conf ? d_topk_conf + b0 * c->top_k : nullptr);
        if (rc != BH_OK) return rc;
        if (c->auto_fallback && d_topk_index && d_topk_conf) {   // what bh_batch_context_synchronize re-runs marked rows from
            const bh_batch_context::Pending p{d_segments + b0 * h.sample_count, nb, d_logits + b0 * h.n_classes,
                                              d_topk_index + b0 * c->top_k, d_topk_conf + b0 * c->top_k};
            bool known = false;
            for (const auto &q : ctx->pending) known |= q.d_seg == p.d_seg && q.n == p.n && q.d_logits == p.d_logits && q.d_idx == p.d_idx && q.d_conf == p.d_conf;
            if (!known) {
                if (ctx->pending.size() < 256) ctx->pending.push_back(p);
                else ctx->pending_overflow = true;
            }
        }
    }
    return BH_OK;
} catch (...) { return on_exception(); }

int bh_batch_context_synchronize(bh_batch_context *ctx) try {
    if (!ctx) return fail(BH_ERR_INVALID, "synchronize: null context");
    bh_classifier *c = ctx->c;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(fetch_nonfinite(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (*ctx->h_nonfinite && c->auto_fallback && !ctx->pending_overflow) {
        // BH_FLAG_AUTO: the marked rows of every forward enqueued since the last synchronise, again on the f32 kernels.  (Forwards
        // that shared their buffers have been overwritten by the last one of them, whose rows these are.)
        std::vector<int32_t> h_idx;
        std::vector<size_t> bad;
        for (const auto &p : ctx->pending) {
            h_idx.resize(p.n * c->top_k);
            HIPCHK(hipMemcpy(h_idx.data(), p.d_idx, h_idx.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            bad.clear();
            for (size_t i = 0; i < p.n; i++)
                if (h_idx[i * c->top_k] == BH_TOPK_NONFINITE) bad.push_back(i);
            const int rc = redo_rows_f32(c, p.d_seg, bad, p.d_logits, p.d_idx, p.d_conf, nullptr);
            if (rc != BH_OK) { ctx->pending.clear(); clear_nonfinite(ctx); return rc; }
        }
        clear_nonfinite(ctx);
    }
    ctx->pending.clear();
    ctx->pending_overflow = false;
    return nonfinite_status(c, ctx);   // BH_ERR_NONFINITE once per occurrence: the counter is cleared
} catch (...) { return on_exception(); }
void *bh_batch_context_stream(bh_batch_context *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

uint64_t bh_tensor_floats(const bh_classifier *c, uint32_t tensor) {
    if (!c || tensor >= c->model.tensor_floats.size()) return 0;
    return c->model.tensor_floats[tensor];
}

int bh_debug_read_tensor(bh_classifier *c, bh_batch_context *ctx, uint32_t tensor, float *host, size_t max_floats) try {
    int rc = check_ctx(c, ctx);
    if (rc != BH_OK) return rc;
    if (!ctx->keep_tensors) return fail(BH_ERR_INVALID, "context was not created with BIRDA_HIP_KEEP_TENSORS=1");
    if (tensor >= c->model.tensor_floats.size()) return fail(BH_ERR_INVALID, "tensor %u out of range", tensor);
    const size_t nfl = c->model.tensor_floats[tensor] * ctx->last_n;
    if (nfl > max_floats) return fail(BH_ERR_INVALID, "host buffer too small (%zu < %zu)", max_floats, nfl);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const float *src = (tensor == c->model.layers.size()) ? ctx->last_logits : ctx->d_arena + ctx->t_off[tensor];
    HIPCHK(hipMemcpy(host, src, nfl * sizeof(float), hipMemcpyDeviceToHost));
    return BH_OK;
} catch (...) { return on_exception(); }

int bh_batch_context_layer_ms(bh_batch_context *ctx, float *ms, uint32_t *launches, size_t n_layers) try {
    if (!ctx || !ms) return fail(BH_ERR_INVALID, "layer_ms: null argument");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < n_layers; i++) { ms[i] = 0.f; if (launches) launches[i] = 0; }
    for (size_t i = 1; i < ctx->ev.size(); i++) {
        const int ly = ctx->ev_layer[i];
        if (ly < 0 || (size_t)ly >= n_layers) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, ctx->ev[i - 1], ctx->ev[i]) == hipSuccess) { ms[ly] += t; if (launches) launches[ly]++; }
    }
    return BH_OK;
} catch (...) { return on_exception(); }

int bh_classifier_fused_blocks(const bh_classifier *c, int32_t *cfgs, size_t cap) {
    if (!c) return 0;
    for (size_t i = 0; i < c->mb.size() && cfgs && i < cap; i++) cfgs[i] = c->mb[i].cfg;
    return (int)c->mb.size();
}

int bh_plan_fused_blocks(const char *model_path, uint32_t flags, int32_t *cfgs, int32_t *layers, size_t cap) try {
    if (!model_path) return fail(BH_ERR_INVALID, "plan_fused_blocks: null model path");
    bh::Model m;
    std::string err;
    if (!bh::load_model(model_path, m, err)) return fail(BH_ERR_IO, "%s", err.c_str());
    const uint32_t p = flags & BH_FLAG_PRECISION_MASK;
    const int precision = p == BH_FLAG_F32 ? 0 : p == BH_FLAG_F16 ? 1 : 3;
    const std::vector<int> readers = tensor_readers(m);
    size_t n = 0;
    for (size_t i = 0; i + 2 < m.layers.size(); i++) {
        bh::MbDesc d{};
        if (!describe_fused_block(m, readers, i, precision, -1, d)) continue;
        if (n < cap) { if (cfgs) cfgs[n] = d.cfg; if (layers) layers[n] = (int32_t)i; }
        n++;
        bh::MbDesc tw{};
        if (bh::mb_plan_twin(d, tw)) {   // the block's small-launch twin is part of what the library must ship: listed behind it
            if (n < cap) { if (cfgs) cfgs[n] = tw.cfg; if (layers) layers[n] = (int32_t)i; }
            n++;
        }
        i += d.noexp ? 1 : 2;
    }
    return (int)n;
} catch (...) { return on_exception(); }

int bh_mb_config_name(int32_t cfg, char *out, size_t cap) {
    return bh::mb_config_name(cfg, out, cap);
}

int bh_classifier_frontend_kernel(const bh_classifier *c, char *out, size_t cap) {
    if (!c) return 0;
    char buf[64];
    const int nmp = c->fe.br[0].nm_pad;
    if (c->fe.prec == 32) snprintf(buf, sizeof buf, "bh::mel32_kernel<%d>", nmp / 32);
    else snprintf(buf, sizeof buf, "bh::mel_kernel<%d, %d>", nmp / 16, c->fe.prec);
    const int n = (int)strlen(buf);
    if (out && cap > (size_t)n) memcpy(out, buf, (size_t)n + 1);
    return n;
}

int bh_debug_mb_stamps(bh_classifier *c, uint64_t *out, size_t cap) {
    if (!c || !c->d_stamps) return 0;
    const size_t n = std::min(cap, c->mb.size() * 8);
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (out && hipMemcpy(out, c->d_stamps, n * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    (void)hipMemset(c->d_stamps, 0, c->mb.size() * 8 * sizeof(unsigned long long));
    return (int)(n / 8);
}

int bh_batch_context_set_profiling(bh_batch_context *ctx, int enabled) try {
    if (!ctx) return fail(BH_ERR_INVALID, "set_profiling: null context");
    if (enabled) {   // a new measurement: drop the events of the previous one (after they have fired)
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (auto e : ctx->ev) (void)hipEventDestroy(e);
        ctx->ev.clear(); ctx->ev_stage.clear(); ctx->ev_layer.clear();
    }
    ctx->profiling = enabled != 0;
    return BH_OK;
} catch (...) { return on_exception(); }

int bh_batch_context_stage_ms(bh_batch_context *ctx, float *ms, uint32_t *launches) try {
    if (!ctx || !ms) return fail(BH_ERR_INVALID, "stage_ms: null argument");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < BH_N_STAGES; i++) { ctx->stage_ms[i] = 0.f; ctx->stage_launches[i] = 0; }
    for (size_t i = 1; i < ctx->ev.size(); i++) {
        const int st = ctx->ev_stage[i];
        if (st < 0) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, ctx->ev[i - 1], ctx->ev[i]) == hipSuccess) {
            ctx->stage_ms[st] += t;
            ctx->stage_launches[st]++;
        }
    }
    for (int i = 0; i < BH_N_STAGES; i++) { ms[i] = ctx->stage_ms[i]; if (launches) launches[i] = ctx->stage_launches[i]; }
    return BH_OK;
} catch (...) { return on_exception(); }

int bh_predict_batch_source_rate(bh_classifier *c, bh_batch_context *ctx, const float *const *segments, size_t n,
                                 size_t n_src_samples, uint32_t source_rate, bh_result *out) try {
    if (!c || !segments || !out) return fail(BH_ERR_INVALID, "predict_batch_source_rate: null argument");
    if (n == 0) return BH_OK;
    const auto &h = c->model.h;
    if (source_rate == h.sample_rate) {
        if (ctx) return bh_predict_batch_with_context(c, ctx, segments, n, n_src_samples, out);
        return bh_predict_batch(c, segments, n, n_src_samples, out);
    }
    std::unique_lock<std::mutex> lock(c->internal_mu, std::defer_lock);
    if (!ctx) {
        lock.lock();
        int rc0 = internal_ctx(c, n, &ctx);
        if (rc0 != BH_OK) return rc0;
    } else {
        int rc0 = check_ctx(c, ctx);
        if (rc0 != BH_OK) return rc0;
        if (n > ctx->asked_batch) return fail(BH_ERR_INVALID, "batch of %zu exceeds context capacity %zu", n, ctx->asked_batch);
    }
    HIPCHK(hipSetDevice(c->device));
    if (ctx->raw_len < n_src_samples) {
        (void)hipFree(ctx->d_raw); (void)hipHostFree(ctx->h_raw);
        ctx->d_raw = nullptr; ctx->h_raw = nullptr; ctx->raw_len = 0;
        HIPCHK(hipMalloc((void **)&ctx->d_raw, ctx->max_batch * n_src_samples * sizeof(float)));
        HIPCHK(hipHostMalloc((void **)&ctx->h_raw, ctx->max_batch * n_src_samples * sizeof(float), hipHostMallocDefault));
        ctx->raw_len = n_src_samples;
    }
    for (size_t b0 = 0; b0 < n; b0 += ctx->max_batch) {
        const size_t nb = std::min(ctx->max_batch, n - b0);
        for (size_t i = 0; i < nb; i++) {
            if (!segments[b0 + i]) return fail(BH_ERR_INVALID, "segment %zu is null", b0 + i);
            memcpy(ctx->h_raw + i * n_src_samples, segments[b0 + i], n_src_samples * sizeof(float));
        }
        HIPCHK(hipMemcpyAsync(ctx->d_raw, ctx->h_raw, nb * n_src_samples * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        // resample_chunk + resize(segment_samples, 0.0) per segment (processor.rs:86-87), on device
        int rc = bh_resample_device(c, ctx, ctx->d_raw, n_src_samples, n_src_samples, source_rate, h.sample_rate,
                                    ctx->d_input, h.sample_count, h.sample_count, nb);
        if (rc != BH_OK) return rc;
        rc = forward_slice(c, ctx, ctx->d_input, nb, ctx->d_logits, ctx->d_topk_idx, ctx->d_topk_conf);
        if (rc != BH_OK) return rc;
        HIPCHK(hipMemcpyAsync(ctx->h_topk_idx, ctx->d_topk_idx, nb * c->top_k * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->h_topk_conf, ctx->d_topk_conf, nb * c->top_k * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(fetch_nonfinite(ctx));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (*ctx->h_nonfinite && c->auto_fallback) {
            bool redone = false;
            const int rr = settle_rows(c, ctx, 0, nb, nullptr, &redone);
            if (rr != BH_OK) return rr;
            clear_nonfinite(ctx);
        }
        for (size_t i = 0; i < nb; i++) {
            bh_result &r = out[b0 + i];
            r.n_pred = 0;
            for (uint32_t k = 0; k < c->top_k; k++) {
                const int32_t id = ctx->h_topk_idx[i * c->top_k + k];
                if (id < 0) break;
                r.index[r.n_pred] = id;
                r.confidence[r.n_pred] = ctx->h_topk_conf[i * c->top_k + k];
                r.n_pred++;
            }
        }
        const int nf = nonfinite_status(c, ctx);
        if (nf != BH_OK) return nf;
    }
    return BH_OK;
} catch (...) { return on_exception(); }

size_t bh_segment_starts(size_t n_frames, size_t segment_samples, size_t overlap_samples, uint64_t *starts, size_t cap) {
    // StreamingDecoder::next_segment over a stream of n_frames (decode.rs:150-202): take = min(seg, left);
    // emit at the running start; advance take - overlap, or stop when that is <= 0 (the buffer is cleared)
    if (overlap_samples >= segment_samples || segment_samples == 0) return 0;
    size_t n = 0, pos = 0;
    while (pos < n_frames) {
        const size_t take = std::min(segment_samples, n_frames - pos);
        if (starts && n < cap) starts[n] = pos;
        n++;
        if (take <= overlap_samples) break;
        pos += take - overlap_samples;
    }
    return n;
}

static int predict_pcm16_core(bh_classifier *c, bh_batch_context *ctx, const void *pcm, uint32_t fmt, size_t n_frames, uint32_t channels,
                              uint32_t source_rate, const std::vector<uint64_t> &starts, size_t seg, bh_result *out,
                              bh_rows_fn on_rows = nullptr, void *user = nullptr);
static inline size_t pcm_bytes_per_sample(uint32_t fmt) { return fmt == BH_PCM_S16 ? 2 : fmt == BH_PCM_S24 ? 3 : (fmt == BH_PCM_S32 || fmt == BH_PCM_F32) ? 4 : 0; }

int bh_predict_pcm16(bh_classifier *c, bh_batch_context *ctx, const int16_t *pcm, size_t n_frames, uint32_t channels,
                     uint32_t source_rate, size_t overlap_samples, bh_result *out, size_t out_cap, size_t *n_segments,
                     uint64_t *start_samples) {
    return bh_predict_pcm(c, ctx, pcm, BH_PCM_S16, n_frames, channels, source_rate, overlap_samples, out, out_cap, n_segments, start_samples);
}

int bh_predict_pcm(bh_classifier *c, bh_batch_context *ctx, const void *pcm, uint32_t sample_format, size_t n_frames, uint32_t channels,
                   uint32_t source_rate, size_t overlap_samples, bh_result *out, size_t out_cap, size_t *n_segments,
                   uint64_t *start_samples) {
    return bh_predict_pcm_rows(c, ctx, pcm, sample_format, n_frames, channels, source_rate, overlap_samples, out, out_cap, n_segments,
                               start_samples, nullptr, nullptr);
}

int bh_predict_pcm_rows(bh_classifier *c, bh_batch_context *ctx, const void *pcm, uint32_t sample_format, size_t n_frames, uint32_t channels,
                        uint32_t source_rate, size_t overlap_samples, bh_result *out, size_t out_cap, size_t *n_segments,
                        uint64_t *start_samples, bh_rows_fn on_rows, void *user) try {
    int rc = check_ctx(c, ctx);
    if (rc != BH_OK) return rc;
    if (!pcm || !out || !n_segments || channels == 0 || !pcm_bytes_per_sample(sample_format)) return fail(BH_ERR_INVALID, "predict_pcm: bad arguments");
    const auto &h = c->model.h;
    const bool resampling = source_rate != h.sample_rate;
    // segment and overlap lengths at the source rate (processor.rs:67-82)
    const size_t seg = resampling ? (size_t)std::ceil((double)h.sample_count * source_rate / h.sample_rate) : h.sample_count;
    const size_t ovl = resampling ? (size_t)std::ceil((double)overlap_samples * source_rate / h.sample_rate) : overlap_samples;
    if (ovl >= seg) return fail(BH_ERR_INVALID, "overlap (%zu) must be shorter than the segment (%zu)", ovl, seg);
    const size_t nseg = bh_segment_starts(n_frames, seg, ovl, nullptr, 0);
    *n_segments = nseg;
    if (nseg > out_cap) return fail(BH_ERR_INVALID, "predict_pcm16: %zu segments, room for %zu", nseg, out_cap);
    if (nseg == 0) return BH_OK;
    std::vector<uint64_t> starts(nseg);
    bh_segment_starts(n_frames, seg, ovl, starts.data(), nseg);
    if (start_samples) memcpy(start_samples, starts.data(), nseg * sizeof(uint64_t));
    return predict_pcm16_core(c, ctx, pcm, sample_format, n_frames, channels, source_rate, starts, seg, out, on_rows, user);
} catch (...) { return on_exception(); }

int bh_predict_pcm16_at(bh_classifier *c, bh_batch_context *ctx, const int16_t *pcm, size_t n_frames, uint32_t channels,
                        uint32_t source_rate, const uint64_t *start_samples, size_t n_segments, bh_result *out) {
    return bh_predict_pcm_at(c, ctx, pcm, BH_PCM_S16, n_frames, channels, source_rate, start_samples, n_segments, out);
}

int bh_predict_pcm_at(bh_classifier *c, bh_batch_context *ctx, const void *pcm, uint32_t sample_format, size_t n_frames, uint32_t channels,
                      uint32_t source_rate, const uint64_t *start_samples, size_t n_segments, bh_result *out) try {
    int rc = check_ctx(c, ctx);
    if (rc != BH_OK) return rc;
    if (!pcm || !out || !start_samples || channels == 0 || !pcm_bytes_per_sample(sample_format)) return fail(BH_ERR_INVALID, "predict_pcm_at: bad arguments");
    if (n_segments == 0) return BH_OK;
    const auto &h = c->model.h;
    const size_t seg = source_rate != h.sample_rate ? (size_t)std::ceil((double)h.sample_count * source_rate / h.sample_rate) : h.sample_count;
    std::vector<uint64_t> starts(start_samples, start_samples + n_segments);
    for (size_t i = 0; i < n_segments; i++)
        if (starts[i] >= n_frames || (i && starts[i] < starts[i - 1]))
            return fail(BH_ERR_INVALID, "predict_pcm16_at: segment %zu starts at %llu (stream of %zu frames; starts must not decrease)", i,
                        (unsigned long long)starts[i], n_frames);
    return predict_pcm16_core(c, ctx, pcm, sample_format, n_frames, channels, source_rate, starts, seg, out);
} catch (...) { return on_exception(); }

static int predict_pcm16_core(bh_classifier *c, bh_batch_context *ctx, const void *pcm, uint32_t fmt, size_t n_frames, uint32_t channels,
                              uint32_t source_rate, const std::vector<uint64_t> &starts, size_t seg, bh_result *out,
                              bh_rows_fn on_rows, void *user) {
    int rc = BH_OK;
    const auto &h = c->model.h;
    const bool resampling = source_rate != h.sample_rate;
    const size_t nseg = starts.size();
    HIPCHK(hipSetDevice(c->device));
    // The stream travels once, as int16 (a quarter of the f32 segments when they overlap by half), slice by
    // slice: worker threads gather 8-MiB pieces of the slice's span into the pinned staging buffer, each
    // piece's H2D copy is enqueued on the copy stream as soon as it is complete, and a sub-slice of
    // segments is cut, resampled and classified as soon as the frames it needs are on their way.
    if (ctx->starts_cap < nseg) {
        (void)hipFree(ctx->d_starts); ctx->d_starts = nullptr; ctx->starts_cap = 0;
        HIPCHK(hipMalloc((void **)&ctx->d_starts, nseg * sizeof(unsigned long long)));
        ctx->starts_cap = nseg;
    }
    HIPCHK(hipMemcpyAsync(ctx->d_starts, starts.data(), nseg * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    if (resampling && ctx->raw_len < seg) {
        (void)hipFree(ctx->d_raw); (void)hipHostFree(ctx->h_raw);
        ctx->d_raw = nullptr; ctx->h_raw = nullptr; ctx->raw_len = 0;
        if (hipMalloc((void **)&ctx->d_raw, ctx->max_batch * seg * sizeof(float)) != hipSuccess ||
            hipHostMalloc((void **)&ctx->h_raw, ctx->max_batch * seg * sizeof(float), hipHostMallocDefault) != hipSuccess)
            return fail(BH_ERR_HIP, "predict_pcm16: scratch allocation failed");
        ctx->raw_len = seg;
    }
    const size_t frame_bytes = (size_t)channels * pcm_bytes_per_sample(fmt);
    const size_t stage_cap = ctx->max_batch * (size_t)h.sample_count * sizeof(float);   // the pinned input staging buffer
    bool pcm_pinned = false;
    {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, pcm) == hipSuccess) pcm_pinned = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError();
    }
    for (size_t b0 = 0; b0 < nseg; b0 += ctx->max_batch) {
        const size_t nb = std::min(ctx->max_batch, nseg - b0);
        const size_t f0 = starts[b0], f1 = std::min<size_t>(n_frames, starts[b0 + nb - 1] + seg);   // frames of this slice
        const size_t bytes = (f1 - f0) * frame_bytes;
        if (ctx->pcm_cap < bytes) {
            (void)hipFree(ctx->d_pcm); ctx->d_pcm = nullptr; ctx->pcm_cap = 0;
            HIPCHK(hipMalloc((void **)&ctx->d_pcm, bytes));
            ctx->pcm_cap = bytes;
        }
        // the segment kernel indexes the stream by absolute frame: hand it the slice buffer's virtual origin
        const char *d_origin = reinterpret_cast<const char *>(ctx->d_pcm) - f0 * frame_bytes;   // (never dereferenced below frame f0)
        const char *src = reinterpret_cast<const char *>(pcm) + f0 * frame_bytes;
        const bool staged = !pcm_pinned && bytes <= stage_cap;   // (more than two channels: the span can exceed the staging buffer)
        char *stage = reinterpret_cast<char *>(ctx->h_input);
        const size_t PIECE = (size_t)8 << 20;
        const size_t npieces = (staged || pcm_pinned) ? (bytes + PIECE - 1) / PIECE : 1;
        const size_t bps = (size_t)((double)bytes / (double)nb);
        const bool can_lane = !resampling && lanes_possible(ctx);   // (the resampler's scratch is one per context)
        std::vector<size_t> cuts = sub_slice_cuts(c, nb, 1, false, bps, can_lane, ctx->forced_sub_slices);
        std::vector<SliceLane> lanes;
        if (can_lane && !lanes_begin(c, ctx, cuts, lanes) && cuts.size() > 1) cuts = sub_slice_cuts(c, nb, 1, false, bps, false, ctx->forced_sub_slices);
        const size_t nsub = cuts.size();
        while (ctx->copy_ev.size() < nsub) {
            hipEvent_t e;
            HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->copy_ev.push_back(e);
        }
        while (ctx->done_ev.size() < nsub) {
            hipEvent_t e;
            HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->done_ev.push_back(e);
        }
        const unsigned nthreads = staged ? (unsigned)std::min<size_t>(copy_threads(), npieces) : 1;
        std::vector<std::atomic<int>> done(npieces);
        for (auto &d : done) d.store(0, std::memory_order_relaxed);
        std::atomic<size_t> next{0};
        auto gather = [&](size_t j) { const size_t o = j * PIECE; memcpy(stage + o, src + o, std::min(PIECE, bytes - o)); };
        std::vector<std::thread> workers;
        if (nthreads > 1)
            for (unsigned t = 0; t < nthreads; t++)
                workers.emplace_back([&] {
                    for (size_t j; (j = next.fetch_add(1)) < npieces;) { gather(j); done[j].store(1, std::memory_order_release); }
                });
        size_t si = 0;   // next sub-slice to launch
        rc = BH_OK;
        for (size_t j = 0; j < npieces && rc == BH_OK; j++) {
            size_t sent;   // bytes of the span on the copy stream after this piece
            if (staged) {
                if (nthreads > 1) while (!done[j].load(std::memory_order_acquire)) std::this_thread::yield();
                else gather(j);
                const size_t o = j * PIECE, len = std::min(PIECE, bytes - o);
                if (hipMemcpyAsync(reinterpret_cast<char *>(ctx->d_pcm) + o, stage + o, len, hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess)
                    { rc = fail(BH_ERR_HIP, "predict_pcm16: upload failed"); break; }
                sent = o + len;
            } else if (pcm_pinned) {   // pinned caller memory (bh_host_alloc / bh_host_register): piece by piece straight from it
                const size_t o = j * PIECE, len = std::min(PIECE, bytes - o);
                if (hipMemcpyAsync(reinterpret_cast<char *>(ctx->d_pcm) + o, src + o, len, hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess)
                    { rc = fail(BH_ERR_HIP, "predict_pcm16: upload failed"); break; }
                sent = o + len;
            } else {
                if (hipMemcpyAsync(ctx->d_pcm, src, bytes, hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess)
                    { rc = fail(BH_ERR_HIP, "predict_pcm16: upload failed"); break; }
                sent = bytes;
            }
            const size_t frames_sent = f0 + sent / frame_bytes;
            while (si < nsub && rc == BH_OK) {
                const size_t s0 = si ? cuts[si - 1] : 0, ns = cuts[si] - s0;
                const size_t need = std::min<size_t>(n_frames, starts[b0 + s0 + ns - 1] + seg);
                if (need > frames_sent && sent < bytes) break;   // its last frames are not on their way yet
                const SliceLane *lane = lanes.empty() ? nullptr : &lanes[si];
                hipStream_t ls = lane ? lane->s : ctx->stream;
                if (hipEventRecord(ctx->copy_ev[si], ctx->copy_stream) != hipSuccess ||
                    hipStreamWaitEvent(ls, ctx->copy_ev[si], 0) != hipSuccess) { rc = fail(BH_ERR_HIP, "stream event failed"); break; }
                float *d_in = ctx->d_input + s0 * h.sample_count;
                if (resampling) {
                    float *d_rw = ctx->d_raw + s0 * seg;
                    bh::launch_segment_pcm(d_origin, (int)fmt, n_frames, (int)channels, ctx->d_starts + b0 + s0, (int)ns, (int)seg, d_rw, seg, ctx->stream);
                    rc = bh_resample_device(c, ctx, d_rw, seg, seg, source_rate, h.sample_rate, d_in, h.sample_count, h.sample_count, ns);
                    if (rc != BH_OK) break;
                } else {
                    bh::launch_segment_pcm(d_origin, (int)fmt, n_frames, (int)channels, ctx->d_starts + b0 + s0, (int)ns, (int)seg, d_in, seg, ls);
                }
                rc = forward_slice(c, ctx, d_in, ns, ctx->d_logits + s0 * h.n_classes, ctx->d_topk_idx + s0 * c->top_k,
                                   ctx->d_topk_conf + s0 * c->top_k, lane);
                // the sub-slice's rows follow its forward down the same stream: they are on the host -- and handed to the caller,
                // below -- while the later sub-slices are still being computed
                if (rc == BH_OK &&
                    (hipMemcpyAsync(ctx->h_topk_idx + s0 * c->top_k, ctx->d_topk_idx + s0 * c->top_k, ns * c->top_k * sizeof(int32_t), hipMemcpyDeviceToHost, ls) != hipSuccess ||
                     hipMemcpyAsync(ctx->h_topk_conf + s0 * c->top_k, ctx->d_topk_conf + s0 * c->top_k, ns * c->top_k * sizeof(float), hipMemcpyDeviceToHost, ls) != hipSuccess ||
                     hipEventRecord(ctx->done_ev[si], ls) != hipSuccess))
                    rc = fail(BH_ERR_HIP, "predict_pcm16: result download failed");
                si++;
            }
        }
        for (auto &w : workers) w.join();
        if (rc == BH_OK && lanes_end(ctx, lanes) != hipSuccess) rc = fail(BH_ERR_HIP, "stream event failed");
        if (rc != BH_OK) {
            (void)hipStreamSynchronize(ctx->copy_stream); lanes_sync(ctx); (void)hipStreamSynchronize(ctx->stream);
            return rc;
        }
        if (fetch_nonfinite(ctx) != hipSuccess) return fail(BH_ERR_HIP, "predict_pcm16: result download failed");
        bool poisoned = false;   // a sub-slice with marked rows and no BH_FLAG_AUTO: its rows and the later ones are not handed out
        for (size_t sj = 0; sj < nsub; sj++) {
            const size_t s0 = sj ? cuts[sj - 1] : 0, ns = cuts[sj] - s0;
            if (hipEventSynchronize(ctx->done_ev[sj]) != hipSuccess) {
                lanes_sync(ctx); (void)hipStreamSynchronize(ctx->stream);
                return fail(BH_ERR_HIP, "predict_pcm16: result download failed");
            }
            // Rows the top-k stage marked (logits inf / NaN from finite samples: an activation left the f16 range) are looked for
            // BEFORE the sub-slice's rows go to the caller: BH_FLAG_AUTO re-runs them on the f32 kernels here; otherwise nothing
            // from this sub-slice on is delivered and the call ends in BH_ERR_NONFINITE (what was delivered before is valid).
            bool marked = false;
            for (size_t i = s0; i < s0 + ns && !marked; i++) marked = ctx->h_topk_idx[i * c->top_k] == BH_TOPK_NONFINITE;
            if (marked && c->auto_fallback) {
                bool redone = false;
                const int rr = settle_rows(c, ctx, s0, ns, nullptr, &redone);
                if (rr != BH_OK) { lanes_sync(ctx); (void)hipStreamSynchronize(ctx->stream); clear_nonfinite(ctx); return rr; }
            } else if (marked) poisoned = true;
            for (size_t i = s0; i < s0 + ns; i++) {
                bh_result &r = out[b0 + i];
                r.n_pred = 0;
                for (uint32_t k = 0; k < c->top_k; k++) {
                    const int32_t id = ctx->h_topk_idx[i * c->top_k + k];
                    if (id < 0) break;
                    r.index[r.n_pred] = id;
                    r.confidence[r.n_pred] = ctx->h_topk_conf[i * c->top_k + k];
                    r.n_pred++;
                }
            }
            if (on_rows && !poisoned) on_rows(user, b0 + s0, ns, out + b0 + s0, starts.data() + b0 + s0);
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(BH_ERR_HIP, "predict_pcm16: result download failed");
        if (c->auto_fallback && *ctx->h_nonfinite) clear_nonfinite(ctx);   // (every marked row was re-run above)
        const int nf = nonfinite_status(c, ctx);
        if (nf != BH_OK) return nf;
    }
    return BH_OK;
}

// ---- custom classifier on embeddings (reference birdnet_onnx::CustomClassifier; lib.rs:883-901, processor.rs:319-360) ----
}  // extern "C"

struct bh_custom_classifier {
    bh::CustomModel model;
    int device = 0;
    uint32_t top_k = 0;
    std::vector<std::string> labels;
    float *d_blob = nullptr;
    std::vector<float *> d_w;     // per layer: W with rows padded to a multiple of 4 (or a pointer into d_blob)
    std::vector<int> ldw;
    std::vector<float *> d_owned;
    // scratch, grown on demand: activations of the two widest layers, logits, top-k rows, host copies
    float *d_act[2] = {nullptr, nullptr};
    float *d_in = nullptr;
    int32_t *d_idx = nullptr;
    float *d_conf = nullptr;
    size_t cap_rows = 0;
    uint32_t max_width = 0;
    uint32_t k0 = 0;              // input width padded to a multiple of 4 (the GEMM's k step): rows of d_in have this stride
    hipStream_t stream = nullptr;
    std::mutex mu;
};

namespace {

int cc_reserve(bh_custom_classifier *cc, size_t rows) {
    if (rows <= cc->cap_rows) return BH_OK;
    for (float *&p : cc->d_act) { (void)hipFree(p); p = nullptr; }
    (void)hipFree(cc->d_in); (void)hipFree(cc->d_idx); (void)hipFree(cc->d_conf);
    cc->d_in = nullptr; cc->d_idx = nullptr; cc->d_conf = nullptr; cc->cap_rows = 0;
    for (float *&p : cc->d_act) HIPCHK(hipMalloc((void **)&p, rows * (size_t)cc->max_width * sizeof(float)));
    HIPCHK(hipMalloc((void **)&cc->d_in, rows * (size_t)cc->k0 * sizeof(float)));
    HIPCHK(hipMalloc((void **)&cc->d_idx, rows * (size_t)cc->top_k * sizeof(int32_t)));
    HIPCHK(hipMalloc((void **)&cc->d_conf, rows * (size_t)cc->top_k * sizeof(float)));
    cc->cap_rows = rows;
    return BH_OK;
}

// dense stack + activation / top-k on device rows [n][input_dim] (row stride in_stride); results to the host
int cc_run(bh_custom_classifier *cc, const float *d_emb, size_t in_stride, size_t n, hipStream_t s, bh_result *out, float *logits_out) {
    const auto &m = cc->model;
    int rc = cc_reserve(cc, n);
    if (rc != BH_OK) return rc;
    const float *cur = d_emb;
    if (in_stride != cc->k0) return fail(BH_ERR_INVALID, "custom classifier: input rows must be contiguous and %u wide", cc->k0);
    for (size_t i = 0; i < m.layers.size(); i++) {
        const auto &L = m.layers[i];
        float *dst = cc->d_act[i & 1];
        bh::launch_pw_gemm(cur, cc->d_w[i], cc->d_blob + L.b_off, nullptr, dst, (int)n, (int)(i == 0 ? cc->k0 : L.in_dim), (int)L.out_dim, cc->ldw[i], (int)L.act, s);
        cur = dst;
    }
    if (!out) {   // every class's activated output, no ranking (the range filter: its last layer carries the sigmoid)
        HIPCHK(hipGetLastError());
        if (logits_out) HIPCHK(hipMemcpyAsync(logits_out, cur, n * (size_t)m.h.n_classes * sizeof(float), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        return BH_OK;
    }
    const uint32_t TK = cc->top_k;
    bh::launch_topk(cur, (int)n, (int)m.h.n_classes, (int)m.h.output_activation, (int)TK, 0.0f, bh::TopkFilter{}, cc->d_idx, cc->d_conf, nullptr, nullptr, s);
    HIPCHK(hipGetLastError());
    std::vector<int32_t> hi(n * TK);
    std::vector<float> hc(n * TK);
    HIPCHK(hipMemcpyAsync(hi.data(), cc->d_idx, n * TK * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hc.data(), cc->d_conf, n * TK * sizeof(float), hipMemcpyDeviceToHost, s));
    if (logits_out) HIPCHK(hipMemcpyAsync(logits_out, cur, n * (size_t)m.h.n_classes * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (size_t i = 0; i < n; i++) {
        bh_result &r = out[i];
        r.n_pred = 0;
        for (uint32_t k = 0; k < TK; k++) {
            if (hi[i * TK + k] < 0) break;
            r.index[r.n_pred] = hi[i * TK + k];
            r.confidence[r.n_pred] = hc[i * TK + k];
            r.n_pred++;
        }
    }
    return BH_OK;
}

}  // namespace

extern "C" {

void bh_custom_classifier_destroy(bh_custom_classifier *cc) {
    if (!cc) return;
    (void)hipSetDevice(cc->device);
    if (cc->stream) { (void)hipStreamSynchronize(cc->stream); (void)hipStreamDestroy(cc->stream); }
    for (float *p : cc->d_owned) (void)hipFree(p);
    for (float *p : cc->d_act) (void)hipFree(p);
    (void)hipFree(cc->d_in); (void)hipFree(cc->d_idx); (void)hipFree(cc->d_conf); (void)hipFree(cc->d_blob);
    delete cc;
}

// labels, device, stream, weights on the device: everything after the model itself has been read into cc->model
static int cc_build(bh_custom_classifier *cc, const char *labels_path, int32_t device, uint32_t top_k, bool drop_blank_labels) {
    const auto &m = cc->model;
    if (labels_path) {
        int rc = read_labels(labels_path, cc->labels);
        if (rc != BH_OK) return rc;
        if (drop_blank_labels) {   // (a geomodel label file: lines trimmed, blank ones skipped, as the reference's loader reads it)
            std::vector<std::string> kept;
            for (auto &l : cc->labels) {
                size_t a = 0, b = l.size();
                while (a < b && isspace((unsigned char)l[a])) a++;
                while (b > a && isspace((unsigned char)l[b - 1])) b--;
                if (b > a) kept.push_back(l.substr(a, b - a));
            }
            cc->labels.swap(kept);
        }
        if (cc->labels.size() != m.h.n_classes)
            return fail(BH_ERR_LABELS, "label count %zu does not match the model's output width %u", cc->labels.size(), m.h.n_classes);
    }
    const int ndev = bh_device_count();
    if (ndev <= 0) return fail(BH_ERR_NO_DEVICE, "no HIP device available (libbirda_hip has no CPU path)");
    if (device < 0 || device >= ndev) return fail(BH_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, ndev - 1);
    cc->device = device;
    cc->top_k = top_k ? std::min<uint32_t>(top_k, BH_MAX_TOP_K) : std::min<uint32_t>(m.h.n_classes, BH_MAX_TOP_K);
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamCreateWithFlags(&cc->stream, hipStreamNonBlocking));
    int rc = upload(m.blob.data(), m.blob.size() * sizeof(float), &cc->d_blob);
    if (rc != BH_OK) return rc;
    cc->k0 = (uint32_t)align_up(m.h.input_dim, 4);
    cc->max_width = cc->k0;
    for (size_t li = 0; li < m.layers.size(); li++) {
        const auto &L = m.layers[li];
        // the GEMM steps k by 4: the FIRST layer's input rows are zero-padded to k0 (a geomodel has 3 inputs); a hidden width
        // that is not a multiple of 4 would need padded activation rows and is refused
        if (li > 0 && L.in_dim % 4) return fail(BH_ERR_UNSUPPORTED, "dense stack: hidden width %u not a multiple of 4", L.in_dim);
        const uint32_t k_rows = li == 0 ? cc->k0 : L.in_dim;
        cc->max_width = std::max(cc->max_width, L.out_dim);
        const int ld = (int)align_up(L.out_dim, 4);
        float *w = cc->d_blob + L.w_off;
        if (ld != (int)L.out_dim || (L.w_off % 4) || k_rows != L.in_dim) {   // rows padded / 16-byte aligned for the GEMM's loads
            std::vector<float> wp((size_t)k_rows * ld, 0.0f);
            for (uint32_t k = 0; k < L.in_dim; k++) memcpy(&wp[(size_t)k * ld], m.blob.data() + L.w_off + (size_t)k * L.out_dim, L.out_dim * sizeof(float));
            float *d = nullptr;
            rc = upload(wp.data(), wp.size() * sizeof(float), &d);
            if (rc != BH_OK) return rc;
            cc->d_owned.push_back(d);
            w = d;
        }
        cc->d_w.push_back(w);
        cc->ldw.push_back(ld);
    }
    return BH_OK;
}

// host rows [n][input_dim] -> d_in rows [n][k0] (zero padded)
static int cc_upload_rows(bh_custom_classifier *cc, const float *rows, size_t n) {
    int rc = cc_reserve(cc, n);
    if (rc != BH_OK) return rc;
    const size_t in = cc->model.h.input_dim;
    if (cc->k0 == in) {
        HIPCHK(hipMemcpyAsync(cc->d_in, rows, n * in * sizeof(float), hipMemcpyHostToDevice, cc->stream));
    } else {
        HIPCHK(hipMemsetAsync(cc->d_in, 0, n * (size_t)cc->k0 * sizeof(float), cc->stream));
        HIPCHK(hipMemcpy2DAsync(cc->d_in, (size_t)cc->k0 * sizeof(float), rows, in * sizeof(float), in * sizeof(float), n, hipMemcpyHostToDevice, cc->stream));
    }
    return BH_OK;
}

int bh_custom_classifier_create(const char *model_path, const char *labels_path, int32_t device, uint32_t top_k,
                                bh_custom_classifier **out) try {
    if (!model_path || !out) return fail(BH_ERR_INVALID, "custom_classifier_create: null argument");
    *out = nullptr;
    std::unique_ptr<bh_custom_classifier, void (*)(bh_custom_classifier *)> cc(new bh_custom_classifier(), bh_custom_classifier_destroy);
    std::string err;
    if (!bh::load_custom_model(model_path, cc->model, err)) return fail(BH_ERR_IO, "%s", err.c_str());
    int rc = cc_build(cc.get(), labels_path, device, top_k, false);
    if (rc != BH_OK) return rc;
    *out = cc.release();
    return BH_OK;
} catch (...) { return on_exception(); }

uint32_t bh_custom_classifier_num_classes(const bh_custom_classifier *cc) { return cc ? cc->model.h.n_classes : 0; }
uint32_t bh_custom_classifier_input_dim(const bh_custom_classifier *cc) { return cc ? cc->model.h.input_dim : 0; }
const char *bh_custom_classifier_label(const bh_custom_classifier *cc, uint32_t index) {
    if (!cc || index >= cc->labels.size()) return nullptr;
    return cc->labels[index].c_str();
}

int bh_custom_classifier_predict_batch(bh_custom_classifier *cc, const float *embeddings, size_t n, bh_result *out) try {
    if (!cc || (n && (!embeddings || !out))) return fail(BH_ERR_INVALID, "custom_classifier_predict_batch: null argument");
    if (n == 0) return BH_OK;
    std::lock_guard<std::mutex> g(cc->mu);
    HIPCHK(hipSetDevice(cc->device));
    int rc = cc_upload_rows(cc, embeddings, n);
    if (rc != BH_OK) return rc;
    return cc_run(cc, cc->d_in, cc->k0, n, cc->stream, out, nullptr);
} catch (...) { return on_exception(); }

int bh_predict_batch_two_stage(bh_classifier *c, bh_batch_context *ctx, bh_custom_classifier *cc, const float *const *segments,
                               size_t n, size_t n_samples, bh_result *out, float *logits_out) try {
    int rc = check_ctx(c, ctx);
    if (rc != BH_OK) return rc;
    if (!cc || !segments || !out) return fail(BH_ERR_INVALID, "predict_batch_two_stage: null argument");
    const auto &h = c->model.h;
    if (n_samples != h.sample_count) return fail(BH_ERR_INVALID, "segment has %zu samples, model expects %u", n_samples, h.sample_count);
    if (cc->device != c->device) return fail(BH_ERR_INVALID, "two-stage: backbone and custom classifier live on different devices");
    if (h.embedding_dim != cc->model.h.input_dim || cc->k0 != cc->model.h.input_dim)
        return fail(BH_ERR_INVALID, "bat mode requires %u-d embeddings from the backbone, the model exposes %u", cc->model.h.input_dim, h.embedding_dim);
    std::lock_guard<std::mutex> g(cc->mu);
    HIPCHK(hipSetDevice(c->device));
    std::vector<bh_result> backbone(std::min(n, ctx->max_batch));
    for (size_t b0 = 0; b0 < n; b0 += ctx->max_batch) {
        const size_t nb = std::min(ctx->max_batch, n - b0);
        // the backbone on this slice (results discarded: the custom classifier's replace them, processor.rs:369-372); the
        // embedding tensor of the slice then sits in the context's arena
        rc = predict_slices(c, ctx, segments + b0, nullptr, nb, backbone.data(), nullptr, nullptr, true);
        if (rc != BH_OK) return rc;
        const float *d_emb = ctx->d_arena + ctx->t_off[h.embedding_tensor];
        rc = cc_run(cc, d_emb, h.embedding_dim, nb, ctx->stream, out + b0, logits_out ? logits_out + b0 * cc->model.h.n_classes : nullptr);
        if (rc != BH_OK) return rc;
    }
    return BH_OK;
} catch (...) { return on_exception(); }

// ---- range filter: the geomodel query (reference src/inference/range_filter.rs:19-51 over birdnet_onnx::RangeFilter) ----
}  // extern "C"

struct bh_range_filter {
    bh_custom_classifier *cc = nullptr;
    float threshold = 0.0f;
};

extern "C" {

void bh_range_filter_destroy(bh_range_filter *rf) {
    if (!rf) return;
    bh_custom_classifier_destroy(rf->cc);
    delete rf;
}

int bh_range_filter_create(const char *model_path, const char *labels_path, int32_t device, float threshold, bh_range_filter **out) try {
    if (!model_path || !labels_path || !out) return fail(BH_ERR_INVALID, "range_filter_create: null argument (a geomodel is built from ITS OWN labels)");
    *out = nullptr;
    if (!(threshold >= 0.0f && threshold <= 1.0f)) return fail(BH_ERR_INVALID, "range filter threshold %g outside 0..1", (double)threshold);
    std::unique_ptr<bh_range_filter, void (*)(bh_range_filter *)> rf(new bh_range_filter(), bh_range_filter_destroy);
    rf->cc = new bh_custom_classifier();
    rf->threshold = threshold;
    char magic[4] = {0, 0, 0, 0};
    if (FILE *f = fopen(model_path, "rb")) { (void)!fread(magic, 1, 4, f); fclose(f); }
    else return fail(BH_ERR_IO, "cannot open geomodel file %s", model_path);
    std::string err;
    const bool ok = memcmp(magic, "BHC1", 4) == 0 ? bh::load_custom_model(model_path, rf->cc->model, err)
                                                  : bh::onnxd::load_dense_onnx(model_path, rf->cc->model, err);
    if (!ok) return fail(BH_ERR_IO, "%s", err.c_str());
    const auto &h = rf->cc->model.h;
    if (h.input_dim != 3) return fail(BH_ERR_UNSUPPORTED, "a geomodel takes (latitude, longitude, week): this model has %u inputs", h.input_dim);
    // the scores must leave the last layer activated: a sigmoid folded into it (onnx_dense.hpp) or written there by the converter
    if (h.output_activation != 0 || rf->cc->model.layers.back().act != bh::ACT_SIGMOID)
        return fail(BH_ERR_UNSUPPORTED, "a geomodel ends in a sigmoid over its species (output activation %u, last layer activation %u)",
                    h.output_activation, rf->cc->model.layers.back().act);
    int rc = cc_build(rf->cc, labels_path, device, 1, true);
    if (rc != BH_OK) return rc;
    *out = rf.release();
    return BH_OK;
} catch (...) { return on_exception(); }

uint32_t bh_range_filter_num_species(const bh_range_filter *rf) { return rf ? rf->cc->model.h.n_classes : 0; }
const char *bh_range_filter_label(const bh_range_filter *rf, uint32_t index) { return rf ? bh_custom_classifier_label(rf->cc, index) : nullptr; }

uint32_t bh_birdnet_week(uint32_t month, uint32_t day) {
    if (month < 1) month = 1;
    if (month > 12) month = 12;
    if (day < 1) day = 1;
    // no clamp on the week inside the month: days 29-31 belong to the next month's first week (capped at 48 for the last days of
    // December) -- the one form birda's week -> start day -> (month, day) round trip inverts for all 48 weeks (birda_hip.h)
    return std::min<uint32_t>(48, (month - 1) * 4 + (day - 1) / 7 + 1);
}

int bh_range_filter_predict_week(bh_range_filter *rf, float latitude, float longitude, float week, float *scores, size_t cap,
                                 uint32_t *indices, size_t *n_kept) try {
    if (!rf || !scores) return fail(BH_ERR_INVALID, "range_filter_predict: null argument");
    const size_t n = rf->cc->model.h.n_classes;
    if (cap < n) return fail(BH_ERR_INVALID, "range_filter_predict: room for %zu scores, the geomodel has %zu species", cap, n);
    bh_custom_classifier *cc = rf->cc;
    std::lock_guard<std::mutex> g(cc->mu);
    HIPCHK(hipSetDevice(cc->device));
    const float row[3] = {latitude, longitude, week};
    int rc = cc_upload_rows(cc, row, 1);
    if (rc != BH_OK) return rc;
    rc = cc_run(cc, cc->d_in, cc->k0, 1, cc->stream, nullptr, scores);
    if (rc != BH_OK) return rc;
    size_t kept = 0;
    for (size_t i = 0; i < n; i++)
        if (scores[i] >= rf->threshold) { if (indices) indices[kept] = (uint32_t)i; kept++; }
    if (n_kept) *n_kept = kept;
    return BH_OK;
} catch (...) { return on_exception(); }

int bh_range_filter_predict(bh_range_filter *rf, double latitude, double longitude, uint32_t month, uint32_t day, float *scores,
                            size_t cap, uint32_t *indices, size_t *n_kept) {
    if (month < 1 || month > 12 || day < 1 || day > 31) return fail(BH_ERR_INVALID, "range_filter_predict: month %u / day %u is not a date", month, day);
    // range_filter.rs:46: `latitude as f32, longitude as f32`
    return bh_range_filter_predict_week(rf, (float)latitude, (float)longitude, (float)bh_birdnet_week(month, day), scores, cap, indices, n_kept);
}

int bh_resample_output_len(size_t n_in, uint32_t from_rate, uint32_t to_rate, size_t *n_out) try {
    if (!n_out || from_rate == 0 || to_rate == 0) return fail(BH_ERR_INVALID, "resample_output_len: bad arguments");
    *n_out = bh::resample_output_len(n_in, from_rate, to_rate);
    return BH_OK;
} catch (...) { return on_exception(); }

int bh_resample_device(bh_classifier *c, bh_batch_context *ctx, const float *d_in, size_t in_stride, size_t src_len,
                       uint32_t from_rate, uint32_t to_rate, float *d_out, size_t out_stride, size_t out_len,
                       size_t n_seg) try {
    int rc = check_ctx(c, ctx);
    if (rc != BH_OK) return rc;
    if (!d_in || !d_out || from_rate == 0 || to_rate == 0 || src_len > in_stride || out_len > out_stride)
        return fail(BH_ERR_INVALID, "resample_device: bad arguments");
    if (n_seg == 0) return BH_OK;
    HIPCHK(hipSetDevice(c->device));
    if (from_rate == to_rate) {  // resample.rs:11-13: identity; then resize(segment_samples, 0.0)
        const size_t ncopy = std::min(src_len, out_len);
        HIPCHK(hipMemcpy2DAsync(d_out, out_stride * sizeof(float), d_in, in_stride * sizeof(float), ncopy * sizeof(float),
                                n_seg, hipMemcpyDeviceToDevice, ctx->stream));
        if (out_len > ncopy)
            HIPCHK(hipMemset2DAsync(d_out + ncopy, out_stride * sizeof(float), 0, (out_len - ncopy) * sizeof(float), n_seg,
                                    ctx->stream));
        return BH_OK;
    }
    const char *err = nullptr;
    const bh::ResamplePlan *pl = bh::resample_plan(from_rate, to_rate, &err);
    if (!pl) return fail(BH_ERR_UNSUPPORTED, "%s (%u -> %u Hz)", err ? err : "resampler", from_rate, to_rate);
    bh::launch_resample(*pl, d_in, in_stride, (int)src_len, d_out, out_stride, (int)out_len, (int)n_seg,
                        c->precision != 0 && !(getenv("BIRDA_HIP_RESAMPLE_F32") && getenv("BIRDA_HIP_RESAMPLE_F32")[0] == '1'), ctx->stream);
    HIPCHK(hipGetLastError());
    return BH_OK;
} catch (...) { return on_exception(); }

int bh_resample(bh_classifier *c, const float *in, size_t n_in, uint32_t from_rate, uint32_t to_rate, float *out,
                size_t out_cap, size_t *n_out) try {
    if (!c || !in || !out || !n_out) return fail(BH_ERR_INVALID, "resample: null argument");
    const size_t need = bh::resample_output_len(n_in, from_rate, to_rate);
    if (need > out_cap) return fail(BH_ERR_INVALID, "resample: output buffer too small (%zu < %zu)", out_cap, need);
    *n_out = need;
    if (need == 0) return BH_OK;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> g(c->internal_mu);
    bh_batch_context *ctx = nullptr;
    int rc = internal_ctx(c, 1, &ctx);
    if (rc != BH_OK) return rc;
    float *d_in = nullptr, *d_out = nullptr;
    HIPCHK(hipMalloc((void **)&d_in, n_in * sizeof(float)));
    if (hipMalloc((void **)&d_out, need * sizeof(float)) != hipSuccess) { (void)hipFree(d_in); return fail(BH_ERR_HIP, "resample: hipMalloc failed"); }
    rc = BH_OK;
    if (hipMemcpyAsync(d_in, in, n_in * sizeof(float), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = fail(BH_ERR_HIP, "resample: H2D failed");
    if (rc == BH_OK) rc = bh_resample_device(c, ctx, d_in, n_in, n_in, from_rate, to_rate, d_out, need, need, 1);
    if (rc == BH_OK && hipMemcpyAsync(out, d_out, need * sizeof(float), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = fail(BH_ERR_HIP, "resample: D2H failed");
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == BH_OK) rc = fail(BH_ERR_HIP, "resample: stream sync failed");
    (void)hipFree(d_in); (void)hipFree(d_out);
    return rc;
} catch (...) { return on_exception(); }

}  // extern "C"

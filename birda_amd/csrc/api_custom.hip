// api_custom.hip -- custom classifiers on embeddings (reference birdnet_onnx::CustomClassifier; lib.rs:883-901, processor.rs:319-360)
// and the geomodel range filter (reference src/inference/range_filter.rs:19-51); split out of api.hip in round 4 (api_internal.hpp).
#include "api_internal.hpp"

using namespace bhi;

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
}  // extern "C"

// bh_multi_*: one process, several shards of one segment list, behind the C ABI.
//
// Every segment is independent through every stage of the path (reference src/pipeline/processor.rs:363-367 treats the
// rows of a batch independently), so shard g of G owns the contiguous block [g N / G, (g + 1) N / G) of the global list
// -- or, for mixed-rate input (BASELINE config 5), the contiguous block whose SOURCE SAMPLES add up to 1 / G of the
// total -- and there is no data-path exchange at all.  A shard = one HIP device ordinal + one batch context (own
// stream, own arena) + one host thread per call.  An ordinal may appear more than once: "logical devices", which is how
// the 8-shard layout of config 3 runs on a box with a single GPU.  Shards on one ordinal share the classifier (weights).
//
// The only exchange is the result gather.  Host entry points hand every shard its slice of the caller's result array
// (hipMemcpyDtoH per device: the results are wanted on the host anyway).  bh_multi_forward_device can also assemble the
// packed top-k rows of all shards on shard 0's device with one RCCL all-gather over xGMI (ncclCommInitAll, one
// communicator per distinct device; ~50 KB per GPU for config 3), falling back to the per-device copies when RCCL is
// not loadable, the communicator does not come up, or two shards share a device.  RCCL is opened at run time (the
// 570-MB library is not a load-time dependency of libbirda_hip.so).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/birda_hip.h"

namespace {

thread_local std::string m_err;
int mfail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    m_err = buf;
    return code;
}
int m_on_exception() noexcept {
    try { throw; }
    catch (const std::bad_alloc &) { return mfail(BH_ERR_INTERNAL, "out of host memory"); }
    catch (const std::exception &e) { return mfail(BH_ERR_INTERNAL, "internal error: %s", e.what()); }
    catch (...) { return mfail(BH_ERR_INTERNAL, "internal error (unknown exception)"); }
}

// ---- RCCL, opened at run time --------------------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **comm, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*AllGather)(const void *send, void *recv, size_t count, int datatype, void *comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // only ever librccl: the collective library of the ROCm install this library was built against
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        r.ok = r.CommInitAll && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd;
    });
    return r;
}

struct Shard {
    int device = 0;
    bh_classifier *clf = nullptr;     // shared by the shards of one device (owner: bh_multi::clfs)
    bh_batch_context *ctx = nullptr;
    // device-resident path: logits scratch + packed top-k rows [rows][top_k]{int32 index}[top_k]{f32 confidence}
    float *d_logits = nullptr;
    size_t logits_rows = 0;
    char *d_pack = nullptr;           // this shard's rows, padded to pack_rows
    char *d_all = nullptr;            // RCCL: every shard's rows
    size_t pack_rows = 0, all_rows = 0;
    void *comm = nullptr;
};

}  // namespace

struct bh_multi {
    std::vector<Shard> shards;
    std::map<int, bh_classifier *> clfs;   // one classifier per distinct device
    uint32_t top_k = 5;
    size_t max_batch = 0;
    bool use_rccl = false;
    std::string gather_note;
    bh_model_info info{};
};

namespace {

// run f(shard index) on one host thread per shard; the first failure (lowest shard) is reported
template <class F> int for_each_shard(bh_multi *m, F &&f) {
    const size_t G = m->shards.size();
    std::vector<int> rc(G, BH_OK);
    std::vector<std::string> msg(G);
    auto body = [&](size_t g) {
        try {
            m_err.clear();      // G == 1 runs on the caller's thread, whose thread-local text may be an earlier call's
            rc[g] = f(g);
            if (rc[g] != BH_OK) msg[g] = m_err.empty() ? bh_last_error() : m_err;
        } catch (...) { rc[g] = m_on_exception(); msg[g] = m_err; }
    };
    if (G == 1) body(0);
    else {
        std::vector<std::thread> th;
        th.reserve(G);
        for (size_t g = 0; g < G; g++) th.emplace_back(body, g);
        for (auto &t : th) t.join();
    }
    for (size_t g = 0; g < G; g++)
        if (rc[g] != BH_OK) return mfail(rc[g], "shard %zu (device %d): %s", g, m->shards[g].device, msg[g].c_str());
    return BH_OK;
}

}  // namespace

extern "C" {

const char *bh_multi_last_error(void) { return m_err.c_str(); }

void bh_shard_range(size_t n_total, uint32_t shard, uint32_t n_shards, size_t *lo, size_t *hi) {
    if (n_shards == 0) n_shards = 1;
    // 128-bit product: n_total * shard cannot overflow for any size_t list length
    const unsigned __int128 n = n_total;
    if (lo) *lo = (size_t)(n * shard / n_shards);
    if (hi) *hi = (size_t)(n * (shard + 1u) / n_shards);
}

int bh_shard_ranges_weighted(const uint64_t *weights, size_t n, uint32_t n_shards, size_t *bounds) try {
    if (!bounds || n_shards == 0 || (n && !weights)) return mfail(BH_ERR_INVALID, "shard_ranges_weighted: bad arguments");
    long double total = 0;
    for (size_t i = 0; i < n; i++) total += (long double)weights[i];
    if (total <= 0) {   // nothing to balance by: equal counts
        for (uint32_t g = 0; g <= n_shards; g++) bh_shard_range(n, g, n_shards, &bounds[g], nullptr);
        return BH_OK;
    }
    // item i goes where its midpoint falls on the cumulative weight axis: contiguous, order-preserving
    std::fill(bounds, bounds + n_shards + 1, n);
    bounds[0] = 0;
    long double acc = 0;
    uint32_t cur = 0;
    for (size_t i = 0; i < n; i++) {
        const long double mid = acc + 0.5L * (long double)weights[i];
        uint32_t g = (uint32_t)std::min<long double>((long double)(n_shards - 1), mid * n_shards / total);
        if (g < cur) g = cur;
        while (cur < g) bounds[++cur] = i;
        acc += (long double)weights[i];
    }
    while (cur < n_shards) bounds[++cur] = n;
    return BH_OK;
} catch (...) { return m_on_exception(); }

void bh_multi_destroy(bh_multi *m) {
    if (!m) return;
    for (auto &s : m->shards) {
        (void)hipSetDevice(s.device);
        if (s.ctx) bh_batch_context_destroy(s.ctx);
        (void)hipFree(s.d_logits); (void)hipFree(s.d_pack); (void)hipFree(s.d_all);
        if (s.comm && rccl().ok) rccl().CommDestroy(s.comm);
    }
    for (auto &kv : m->clfs) bh_classifier_destroy(kv.second);
    delete m;
}

int bh_multi_create(const bh_multi_config *cfg, bh_multi **out) try {
    if (!cfg || !out || !cfg->model_path) return mfail(BH_ERR_INVALID, "multi_create: null config / model_path");
    *out = nullptr;
    const int ndev = bh_device_count();
    if (ndev <= 0) return mfail(BH_ERR_NO_DEVICE, "no HIP device available (libbirda_hip has no CPU path)");
    std::vector<int> devs;
    if (cfg->n_devices == 0) for (int d = 0; d < ndev; d++) devs.push_back(d);
    else {
        if (!cfg->devices) return mfail(BH_ERR_INVALID, "multi_create: n_devices > 0 but devices is null");
        for (uint32_t i = 0; i < cfg->n_devices; i++) {
            if (cfg->devices[i] < 0 || cfg->devices[i] >= ndev)
                return mfail(BH_ERR_NO_DEVICE, "multi_create: device %d out of range (0..%d)", cfg->devices[i], ndev - 1);
            devs.push_back(cfg->devices[i]);
        }
    }
    std::unique_ptr<bh_multi, void (*)(bh_multi *)> m(new bh_multi(), bh_multi_destroy);
    m->top_k = cfg->top_k ? cfg->top_k : 5;
    for (int d : devs) {
        if (!m->clfs.count(d)) {
            bh_config c{cfg->model_path, cfg->labels_path, m->top_k, cfg->min_confidence, d, cfg->flags};
            bh_classifier *clf = nullptr;
            int rc = bh_classifier_create(&c, &clf);
            if (rc != BH_OK) return mfail(rc, "device %d: %s", d, bh_last_error());
            m->clfs[d] = clf;
        }
    }
    bh_classifier_info(m->clfs.begin()->second, &m->info);
    m->max_batch = cfg->max_batch ? cfg->max_batch : bh_classifier_default_batch_size(m->clfs.begin()->second);
    for (int d : devs) {
        Shard s;
        s.device = d;
        s.clf = m->clfs[d];
        int rc = bh_batch_context_create(s.clf, m->max_batch, &s.ctx);
        if (rc != BH_OK) return mfail(rc, "device %d: %s", d, bh_last_error());
        m->shards.push_back(s);
    }
    // result gather: RCCL needs one rank per DISTINCT device
    const bool distinct = m->clfs.size() == devs.size();
    m->use_rccl = false;
    if (cfg->gather == BH_GATHER_HOST) m->gather_note = "host (requested)";
    else if (!distinct) m->gather_note = "host (several shards share a device: one RCCL rank per device only)";
    else if (!rccl().ok) m->gather_note = "host (librccl not loadable)";
    else {
        std::vector<void *> comms(devs.size(), nullptr);
        const int r = rccl().CommInitAll(comms.data(), (int)devs.size(), devs.data());
        if (r != 0) m->gather_note = std::string("host (ncclCommInitAll failed: ") + (rccl().GetErrorString ? rccl().GetErrorString(r) : "?") + ")";
        else {
            for (size_t g = 0; g < devs.size(); g++) m->shards[g].comm = comms[g];
            m->use_rccl = true;
            m->gather_note = "rccl";
        }
    }
    if (cfg->gather == BH_GATHER_RCCL && !m->use_rccl) return mfail(BH_ERR_UNSUPPORTED, "multi_create: RCCL gather requested but unavailable: %s", m->gather_note.c_str());
    *out = m.release();
    return BH_OK;
} catch (...) { return m_on_exception(); }

uint32_t bh_multi_shards(const bh_multi *m) { return m ? (uint32_t)m->shards.size() : 0; }
const char *bh_multi_gather_backend(const bh_multi *m) { return m ? m->gather_note.c_str() : ""; }
bh_classifier *bh_multi_classifier(bh_multi *m, uint32_t shard) { return (m && shard < m->shards.size()) ? m->shards[shard].clf : nullptr; }
bh_batch_context *bh_multi_context(bh_multi *m, uint32_t shard) { return (m && shard < m->shards.size()) ? m->shards[shard].ctx : nullptr; }
int bh_multi_shard_device(const bh_multi *m, uint32_t shard) { return (m && shard < m->shards.size()) ? m->shards[shard].device : -1; }

int bh_multi_predict_batch_contig(bh_multi *m, const float *base, size_t n, bh_result *out) try {
    if (!m || (n && (!base || !out))) return mfail(BH_ERR_INVALID, "multi_predict_batch_contig: null argument");
    const uint32_t G = (uint32_t)m->shards.size();
    const size_t S = m->info.sample_count;
    return for_each_shard(m, [&](size_t g) -> int {
        size_t lo, hi;
        bh_shard_range(n, (uint32_t)g, G, &lo, &hi);
        if (hi == lo) return BH_OK;
        return bh_predict_batch_contig(m->shards[g].clf, m->shards[g].ctx, base + lo * S, hi - lo, out + lo);
    });
} catch (...) { return m_on_exception(); }

int bh_multi_predict_batch_source_rate(bh_multi *m, const float *const *segments, const uint32_t *source_rates,
                                       const size_t *n_src_samples, size_t n, bh_result *out, size_t *bounds_out) try {
    if (!m || (n && (!segments || !source_rates || !n_src_samples || !out))) return mfail(BH_ERR_INVALID, "multi_predict_batch_source_rate: null argument");
    const uint32_t G = (uint32_t)m->shards.size();
    std::vector<uint64_t> w(n);
    for (size_t i = 0; i < n; i++) w[i] = n_src_samples[i];
    std::vector<size_t> bounds(G + 1);
    int rc = bh_shard_ranges_weighted(w.data(), n, G, bounds.data());   // balance by source samples
    if (rc != BH_OK) return rc;
    if (bounds_out) memcpy(bounds_out, bounds.data(), (G + 1) * sizeof(size_t));
    return for_each_shard(m, [&](size_t g) -> int {
        const size_t lo = bounds[g], hi = bounds[g + 1];
        // one batch per (rate, length) present in the shard; results scatter back to list order
        std::map<std::pair<uint32_t, size_t>, std::vector<size_t>> groups;
        for (size_t i = lo; i < hi; i++) groups[{source_rates[i], n_src_samples[i]}].push_back(i);
        for (auto &kv : groups) {
            const auto &ids = kv.second;
            std::vector<const float *> ptrs(ids.size());
            std::vector<bh_result> res(ids.size());
            for (size_t j = 0; j < ids.size(); j++) ptrs[j] = segments[ids[j]];
            for (size_t j0 = 0; j0 < ids.size(); j0 += m->max_batch) {   // the context's capacity per call
                const size_t nb = std::min(m->max_batch, ids.size() - j0);
                int r = bh_predict_batch_source_rate(m->shards[g].clf, m->shards[g].ctx, ptrs.data() + j0, nb, kv.first.second, kv.first.first,
                                                     res.data() + j0);
                if (r != BH_OK) return r;
            }
            for (size_t j = 0; j < ids.size(); j++) out[ids[j]] = res[j];
        }
        return BH_OK;
    });
} catch (...) { return m_on_exception(); }

int bh_multi_forward_device(bh_multi *m, const float *const *d_segments, const size_t *n_per_shard, bh_result *out) try {
    if (!m || !d_segments || !n_per_shard || !out) return mfail(BH_ERR_INVALID, "multi_forward_device: null argument");
    const size_t G = m->shards.size();
    const uint32_t TK = m->top_k;
    const size_t row_bytes = (size_t)TK * 8, NC = m->info.n_classes;
    size_t total = 0, max_n = 0;
    std::vector<size_t> first(G + 1, 0);
    for (size_t g = 0; g < G; g++) {
        if (n_per_shard[g] && !d_segments[g]) return mfail(BH_ERR_INVALID, "multi_forward_device: shard %zu has no input pointer", g);
        first[g + 1] = first[g] + n_per_shard[g];
        max_n = std::max(max_n, n_per_shard[g]);
    }
    total = first[G];
    if (total == 0) return BH_OK;
    std::vector<char> host_rows(m->use_rccl ? G * max_n * row_bytes : total * row_bytes);
    // enqueue every shard's forward + the packing of its top-k rows on the shard's own stream
    int rc = for_each_shard(m, [&](size_t g) -> int {
        Shard &s = m->shards[g];
        const size_t n = n_per_shard[g];
        if (hipSetDevice(s.device) != hipSuccess) return mfail(BH_ERR_HIP, "hipSetDevice(%d) failed", s.device);
        const size_t lrows = std::min(std::max<size_t>(n, 1), m->max_batch);   // logits are scratch: one micro-batch of rows
        if (s.logits_rows < lrows) {
            (void)hipFree(s.d_logits); s.d_logits = nullptr; s.logits_rows = 0;
            if (hipMalloc((void **)&s.d_logits, lrows * NC * sizeof(float)) != hipSuccess) return mfail(BH_ERR_HIP, "hipMalloc(logits scratch) failed");
            s.logits_rows = lrows;
        }
        if (s.pack_rows < max_n) {
            (void)hipFree(s.d_pack); s.d_pack = nullptr; s.pack_rows = 0;
            if (hipMalloc((void **)&s.d_pack, max_n * row_bytes) != hipSuccess) return mfail(BH_ERR_HIP, "hipMalloc(packed rows) failed");
            s.pack_rows = max_n;
        }
        if (m->use_rccl && s.all_rows < G * max_n) {
            (void)hipFree(s.d_all); s.d_all = nullptr; s.all_rows = 0;
            if (hipMalloc((void **)&s.d_all, G * max_n * row_bytes) != hipSuccess) return mfail(BH_ERR_HIP, "hipMalloc(gather buffer) failed");
            s.all_rows = G * max_n;
        }
        hipStream_t st = (hipStream_t)bh_batch_context_stream(s.ctx);
        // packed row layout: the int32 indices of a slice, then its confidences, written by the top-k kernel into two
        // planes of the slice's piece of d_pack: [slice]{ idx[rows][TK] ; conf[rows][TK] } -- re-ordered on the host
        for (size_t b0 = 0; b0 < n; b0 += m->max_batch) {
            const size_t nb = std::min(m->max_batch, n - b0);
            char *piece = s.d_pack + b0 * row_bytes;
            int r = bh_forward_device(s.clf, s.ctx, d_segments[g] + b0 * m->info.sample_count, nb, s.d_logits,
                                      reinterpret_cast<int32_t *>(piece), reinterpret_cast<float *>(piece + nb * TK * 4));
            if (r != BH_OK) return r;
        }
        if (!m->use_rccl) {
            if (n && hipMemcpyAsync(host_rows.data() + first[g] * row_bytes, s.d_pack, n * row_bytes, hipMemcpyDeviceToHost, st) != hipSuccess)
                return mfail(BH_ERR_HIP, "result download failed");
        }
        return BH_OK;
    });
    if (rc != BH_OK) return rc;
    if (m->use_rccl) {
        // one all-gather of the padded row blocks; every rank receives all of them, shard 0's copy goes to the host
        Rccl &R = rccl();
        R.GroupStart();
        int r = 0;
        for (size_t g = 0; g < G && r == 0; g++) {
            Shard &s = m->shards[g];
            (void)hipSetDevice(s.device);
            r = R.AllGather(s.d_pack, s.d_all, max_n * row_bytes, /* ncclInt8 */ 0, s.comm, (hipStream_t)bh_batch_context_stream(s.ctx));
        }
        const int r2 = R.GroupEnd();
        if (r != 0 || r2 != 0) return mfail(BH_ERR_HIP, "ncclAllGather failed: %s", R.GetErrorString ? R.GetErrorString(r ? r : r2) : "?");
        Shard &s0 = m->shards[0];
        (void)hipSetDevice(s0.device);
        if (hipMemcpyAsync(host_rows.data(), s0.d_all, G * max_n * row_bytes, hipMemcpyDeviceToHost, (hipStream_t)bh_batch_context_stream(s0.ctx)) != hipSuccess)
            return mfail(BH_ERR_HIP, "result download failed");
    }
    // (BH_FLAG_AUTO: bh_batch_context_synchronize re-runs rows beyond the f16 range on the f32 kernels, in place on the device --
    //  AFTER the rows above were copied / gathered.  A shard whose classifier's fall-back count moved is downloaded again.)
    std::vector<uint64_t> fb_before(G);
    for (size_t g = 0; g < G; g++) fb_before[g] = bh_classifier_fallback_segments(m->shards[g].clf);
    rc = for_each_shard(m, [&](size_t g) -> int { return bh_batch_context_synchronize(m->shards[g].ctx); });
    if (rc != BH_OK) return rc;
    for (size_t g = 0; g < G; g++) {
        Shard &s = m->shards[g];
        if (!n_per_shard[g] || bh_classifier_fallback_segments(s.clf) == fb_before[g]) continue;
        (void)hipSetDevice(s.device);
        if (hipMemcpy(host_rows.data() + (m->use_rccl ? g * max_n : first[g]) * row_bytes, s.d_pack, n_per_shard[g] * row_bytes, hipMemcpyDeviceToHost) != hipSuccess)
            return mfail(BH_ERR_HIP, "result download failed");
    }
    // unpack: per shard, per micro-batch slice {idx plane, conf plane} -> bh_result rows in list order
    for (size_t g = 0; g < G; g++) {
        const char *rows = host_rows.data() + (m->use_rccl ? g * max_n : first[g]) * row_bytes;
        const size_t n = n_per_shard[g];
        for (size_t b0 = 0; b0 < n; b0 += m->max_batch) {
            const size_t nb = std::min(m->max_batch, n - b0);
            const int32_t *idx = reinterpret_cast<const int32_t *>(rows + b0 * row_bytes);
            const float *conf = reinterpret_cast<const float *>(rows + b0 * row_bytes + nb * TK * 4);
            for (size_t i = 0; i < nb; i++) {
                bh_result &r = out[first[g] + b0 + i];
                r.n_pred = 0;
                for (uint32_t k = 0; k < TK; k++) {
                    if (idx[i * TK + k] < 0) break;
                    r.index[r.n_pred] = idx[i * TK + k];
                    r.confidence[r.n_pred] = conf[i * TK + k];
                    r.n_pred++;
                }
            }
        }
    }
    return BH_OK;
} catch (...) { return m_on_exception(); }

}  // extern "C"

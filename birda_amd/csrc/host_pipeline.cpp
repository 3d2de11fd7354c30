// Host side of the hot path, mirroring the reference's per-file pipeline over the C ABI.
//
//   StreamingDecoder        reference src/audio/decode.rs:34-245 (WAV subset of symphonia)
//   decode_and_stream       reference src/pipeline/processor.rs:49-108 (producer thread)
//   run_streaming_inference reference src/pipeline/processor.rs:114-190 (consumer, batcher, sort)
//   process_batch           reference src/pipeline/processor.rs:220-410 (padding, watchdog, dispatch, threshold)
//   process_file            reference src/pipeline/processor.rs:418-796
//   CsvWriter               reference src/output/csv.rs:17-132, Detection::from_label types.rs:58-79
//   inference watchdog      reference src/gpu/watchdog.rs:22-66
//
// The reference is Rust; no cargo/rustc exists in this image, so the host side is C++ above
// the same C ABI a Rust `extern "C"` binding would use.  All model compute
// goes through bh_predict* (HIP); nothing here computes logits on the CPU.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <limits>
#include <vector>

#include "../../include/birda_hip.h"
#include "../../include/birda_host.h"

namespace {

thread_local std::string h_err;
int hfail(int code, const std::string &msg) { h_err = msg; return code; }

// ---------------------------------------------------------------------------------------
// WAV container -> interleaved PCM packets (the symphonia role, WAV only)
// ---------------------------------------------------------------------------------------
enum SampleFmt { FMT_NONE = 0, FMT_S16, FMT_S24, FMT_S32, FMT_F32, FMT_U8 };

uint32_t rd32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

}  // namespace

struct bh_decoder {
    FILE *f = nullptr;
    std::string path;
    uint32_t sample_rate = 0;
    int channels = 1;
    SampleFmt fmt = FMT_NONE;
    int bytes_per_sample = 2;
    uint64_t data_bytes = 0, data_read = 0;
    bool has_duration = false;
    double duration_secs = 0.0;
    // StreamingDecoder state (decode.rs:34-50)
    std::vector<float> buffer;
    size_t samples_emitted = 0;
    bool eof = false;
    static constexpr size_t PACKET_FRAMES = 1152;

    // append_samples -- decode.rs:353-411
    void append(const unsigned char *raw, size_t frames) {
        const int ch = channels;
        for (size_t i = 0; i < frames; i++) {
            float sum = 0.0f;
            for (int c = 0; c < ch; c++) {
                const unsigned char *p = raw + (i * ch + c) * bytes_per_sample;
                float v;
                switch (fmt) {
                case FMT_S16: v = (float)(int16_t)rd16(p) / 32768.0f; break;                     // :372-374
                case FMT_S32: v = (float)(int32_t)rd32(p) / 2147483648.0f; break;                 // :388-391
                case FMT_S24: {  // symphonia widens 24-bit PCM into its S32 buffer (value << 8)  [EXT]
                    int32_t s = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24);
                    v = (float)s / 2147483648.0f;
                } break;
                case FMT_F32: { uint32_t u = rd32(p); memcpy(&v, &u, 4); } break;               // :356-358
                default: return;  // other formats are silently skipped (:407-409)
                }
                if (ch == 1) { buffer.push_back(v); goto next; }
                sum += v;
            }
            buffer.push_back(sum / (float)ch);
        next:;
        }
    }

    // decode_next_packet -- decode.rs:205-245
    void decode_next_packet() {
        const size_t frame_bytes = (size_t)channels * bytes_per_sample;
        uint64_t left = data_bytes - data_read;
        size_t want = (size_t)std::min<uint64_t>(left / frame_bytes, PACKET_FRAMES);
        if (want == 0) { eof = true; return; }
        std::vector<unsigned char> raw(want * frame_bytes);
        size_t got = fread(raw.data(), frame_bytes, want, f);
        if (got == 0) { eof = true; return; }
        data_read += (uint64_t)got * frame_bytes;
        append(raw.data(), got);
    }
};

extern "C" {

const char *bhh_last_error(void) { return h_err.c_str(); }

// StreamingDecoder::open -- decode.rs:54-128
int bhh_decoder_open(const char *path, bh_decoder **out) {
    if (!path || !out) return hfail(BH_ERR_INVALID, "decoder_open: null argument");
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return hfail(BH_ERR_IO, std::string("AudioOpen: cannot open ") + path);
    auto d = std::make_unique<bh_decoder>();
    d->f = f;
    d->path = path;
    unsigned char hdr[12];
    if (fread(hdr, 1, 12, f) != 12 || memcmp(hdr, "RIFF", 4) != 0 || memcmp(hdr + 8, "WAVE", 4) != 0) {
        fclose(f); d->f = nullptr;
        return hfail(BH_ERR_UNSUPPORTED, std::string("AudioOpen: ") + path + " is not a RIFF/WAVE file (only WAV is decoded here)");
    }
    bool have_fmt = false, have_data = false;
    while (!have_data) {
        unsigned char ch[8];
        if (fread(ch, 1, 8, f) != 8) break;
        uint32_t sz = rd32(ch + 4);
        if (memcmp(ch, "fmt ", 4) == 0) {
            std::vector<unsigned char> b(sz);
            if (fread(b.data(), 1, sz, f) != sz || sz < 16) break;
            uint16_t tag = rd16(&b[0]);
            d->channels = rd16(&b[2]);
            d->sample_rate = rd32(&b[4]);
            uint16_t bits = rd16(&b[14]);
            if (tag == 0xFFFE && sz >= 26) tag = rd16(&b[24]);  // WAVE_FORMAT_EXTENSIBLE sub-format
            d->bytes_per_sample = bits / 8;
            if (tag == 1) d->fmt = bits == 16 ? FMT_S16 : bits == 24 ? FMT_S24 : bits == 32 ? FMT_S32 : bits == 8 ? FMT_U8 : FMT_NONE;
            else if (tag == 3 && bits == 32) d->fmt = FMT_F32;
            else d->fmt = FMT_NONE;
            have_fmt = true;
            if (sz & 1) fseek(f, 1, SEEK_CUR);
        } else if (memcmp(ch, "data", 4) == 0) {
            long pos = ftell(f);
            fseek(f, 0, SEEK_END);
            long end = ftell(f);
            fseek(f, pos, SEEK_SET);
            uint64_t avail = (uint64_t)(end - pos);
            d->data_bytes = (sz == 0xFFFFFFFFu || sz == 0 || sz > avail) ? avail : sz;
            have_data = true;
        } else {
            fseek(f, (long)sz + (sz & 1), SEEK_CUR);
        }
    }
    if (!have_fmt || !have_data || d->channels <= 0)
        { fclose(f); d->f = nullptr; return hfail(BH_ERR_IO, std::string("NoAudioTracks: ") + path); }
    if (d->sample_rate == 0) { fclose(f); d->f = nullptr; return hfail(BH_ERR_IO, std::string("AudioDecode: missing sample rate in ") + path); }
    if (d->fmt == FMT_NONE || d->bytes_per_sample == 0)
        { fclose(f); d->f = nullptr; return hfail(BH_ERR_UNSUPPORTED, std::string("AudioDecode: unsupported WAV sample format in ") + path); }
    const uint64_t n_frames = d->data_bytes / ((uint64_t)d->channels * d->bytes_per_sample);
    d->has_duration = true;  // n_frames / sample_rate, decode.rs:102-105
    d->duration_secs = (double)n_frames / (double)d->sample_rate;
    *out = d.release();
    return BH_OK;
}

void bhh_decoder_close(bh_decoder *d) {
    if (!d) return;
    if (d->f) fclose(d->f);
    delete d;
}

uint32_t bhh_decoder_sample_rate(const bh_decoder *d) { return d ? d->sample_rate : 0; }
int bhh_decoder_duration_hint(const bh_decoder *d, double *secs) {
    if (!d || !d->has_duration) return 0;
    if (secs) *secs = d->duration_secs;
    return 1;
}

// StreamingDecoder::next_segment -- decode.rs:150-202.  returns 1 segment / 0 exhausted / <0 error
int bhh_decoder_next_segment(bh_decoder *d, size_t segment_samples, size_t overlap_samples, float *out,
                             size_t *start_sample) {
    if (!d || !out || !start_sample) return hfail(BH_ERR_INVALID, "next_segment: null argument");
    if (overlap_samples >= segment_samples)  // :156-162
        return hfail(BH_ERR_INVALID, "overlap_samples (" + std::to_string(overlap_samples) + ") must be less than segment_samples (" +
                                         std::to_string(segment_samples) + ")");
    while (d->buffer.size() < segment_samples && !d->eof) d->decode_next_packet();  // :165-167
    if (d->buffer.empty()) return 0;                                                  // :170-172
    const size_t take = std::min(segment_samples, d->buffer.size());                  // :175
    memcpy(out, d->buffer.data(), take * sizeof(float));
    std::fill(out + take, out + segment_samples, 0.0f);                               // :178-181
    *start_sample = d->samples_emitted;                                               // :183
    const size_t advance = take > overlap_samples ? take - overlap_samples : 0;       // :186
    if (advance > 0) {
        d->buffer.erase(d->buffer.begin(), d->buffer.begin() + (long)advance);
        d->samples_emitted += advance;
    } else {                                                                          // :191-196
        d->buffer.clear();
        d->samples_emitted += take;
    }
    return 1;
}

// estimate_segment_count -- src/output/progress.rs:80-92 (-1 = None)
int64_t bhh_estimate_segment_count(int has_duration, double duration_secs, float segment_duration, float overlap) {
    if (!has_duration) return -1;
    const float step = segment_duration - overlap;
    if (step <= 0.0f) return -1;
    return (int64_t)std::ceil(duration_secs / (double)step);
}

// effective batch size -- processor.rs:531-545
size_t bhh_effective_batch_size(size_t batch_size, int64_t estimated_segments) {
    if (estimated_segments < 0) return batch_size;
    const size_t est = (size_t)estimated_segments;
    if (est == 0) return batch_size;
    return batch_size > est ? est : batch_size;
}

// source segment sizing -- processor.rs:67-82
size_t bhh_source_samples(size_t target_samples, uint32_t source_rate, uint32_t target_rate) {
    if (source_rate == target_rate) return target_samples;
    return (size_t)std::ceil((double)target_samples * (double)source_rate / (double)target_rate);
}

// (seconds * rate as f32) as usize -- processor.rs:514,520
size_t bhh_duration_to_samples(float seconds, uint32_t rate) {
    const float v = seconds * (float)rate;
    return v <= 0.0f ? 0 : (size_t)v;
}

}  // extern "C"

namespace {

// escape_csv -- csv.rs:126-132
std::string escape_csv(const std::string &v) {
    if (v.find(',') != std::string::npos || v.find('"') != std::string::npos || v.find('\n') != std::string::npos) {
        std::string o = "\"";
        for (char c : v) { if (c == '"') o += '"'; o += c; }
        return o + "\"";
    }
    return v;
}

struct Detection {  // output/types.rs:8-23
    float start_time, end_time, confidence;
    std::string scientific_name, common_name;
};

// Detection::from_label -- types.rs:58-79
Detection detection_from_label(const std::string &label, float conf, float start, float end) {
    Detection d{start, end, conf, label, label};
    const size_t us = label.find('_');
    if (us != std::string::npos) { d.scientific_name = label.substr(0, us); d.common_name = label.substr(us + 1); }
    return d;
}

std::string csv_row(const Detection &d, const std::string &path) {  // csv.rs:55-66, DECIMAL_PLACES = 4
    char num[96];
    snprintf(num, sizeof num, "%.1f,%.1f,", (double)d.start_time, (double)d.end_time);
    std::string row = num;
    row += escape_csv(d.scientific_name) + "," + escape_csv(d.common_name) + ",";
    snprintf(num, sizeof num, "%.4f,", (double)d.confidence);
    row += num;
    row += escape_csv(path) + "\n";
    return row;
}

const char *CSV_HEADER = "Start (s),End (s),Scientific name,Common name,Confidence,File\n";  // csv.rs:41-52

// ---- watchdog (gpu/watchdog.rs:22-66) ----
struct Watchdog {
    std::shared_ptr<std::atomic<bool>> cancelled;
};

// inference_watchdog_timeout -- processor.rs:194-211
uint64_t watchdog_timeout_secs() {
    const char *v = getenv("BIRDA_INFERENCE_TIMEOUT");
    if (v && *v) {
        char *end = nullptr;
        unsigned long long t = strtoull(v, &end, 10);
        if (end && *end == 0 && t >= 1 && t <= 3600) return t;
    }
    return 10;
}

struct AudioChunk {  // chunker.rs:4-12
    std::vector<float> samples;
    float start_time, end_time;
};

// bounded channel == sync_channel(capacity) (processor.rs:640-641)
struct Channel {
    std::mutex mu;
    std::condition_variable not_full, not_empty;
    std::deque<AudioChunk> q;
    size_t cap;
    bool closed = false, receiver_gone = false;
    int err_code = 0;
    std::string err;
    explicit Channel(size_t c) : cap(c) {}
    bool send(AudioChunk &&c) {
        std::unique_lock<std::mutex> l(mu);
        not_full.wait(l, [&] { return q.size() < cap || receiver_gone; });
        if (receiver_gone) return false;  // DecodeChannelClosed
        q.push_back(std::move(c));
        not_empty.notify_one();
        return true;
    }
    void close(int code = 0, const std::string &e = "") {
        std::lock_guard<std::mutex> l(mu);
        closed = true; err_code = code; err = e;
        not_empty.notify_all();
    }
    // 1 = chunk, 0 = end of stream, <0 error
    int recv(AudioChunk &out) {
        std::unique_lock<std::mutex> l(mu);
        not_empty.wait(l, [&] { return !q.empty() || closed; });
        if (!q.empty()) { out = std::move(q.front()); q.pop_front(); not_full.notify_one(); return 1; }
        return err_code;
    }
    void drop_receiver() {
        std::lock_guard<std::mutex> l(mu);
        receiver_gone = true;
        not_full.notify_all();
    }
};

}  // namespace

extern "C" {

uint64_t bhh_watchdog_timeout_secs(void) { return watchdog_timeout_secs(); }

// start_inference_watchdog -- gpu/watchdog.rs:22-52.  The returned handle is the WatchdogGuard;
// bhh_watchdog_cancel == drop(guard) (:61-65).
void *bhh_watchdog_start(uint64_t timeout_ms, size_t batch_size) {
    auto *w = new Watchdog{std::make_shared<std::atomic<bool>>(false)};
    auto flag = w->cancelled;
    std::thread([flag, timeout_ms, batch_size] {
        std::this_thread::sleep_for(std::chrono::milliseconds(timeout_ms));
        if (!flag->load(std::memory_order_seq_cst)) {
            const size_t suggested = std::max<size_t>(batch_size / 2, 1);
            fprintf(stderr, "\nFATAL: Inference timeout after %llus (batch size: %zu)\n\n"
                            "The GPU inference operation did not complete within the expected time.\n"
                            "Recommendations:\n  1. Reduce batch size: birda -b %zu <input>\n"
                            "  2. Close other GPU applications and try again\n\n"
                            "Terminating process to prevent system lockup.\n",
                    (unsigned long long)(timeout_ms / 1000), batch_size, suggested);
            _Exit(1);
        }
    }).detach();
    return w;
}
void bhh_watchdog_cancel(void *guard) {
    auto *w = static_cast<Watchdog *>(guard);
    if (!w) return;
    w->cancelled->store(true, std::memory_order_seq_cst);
    delete w;
}

size_t bhh_csv_header(int bom, char *out, size_t cap) {
    std::string s;
    if (bom) s = "\xEF\xBB\xBF";  // UTF8_BOM, constants.rs:437
    s += CSV_HEADER;
    if (s.size() + 1 > cap) return 0;
    memcpy(out, s.c_str(), s.size() + 1);
    return s.size();
}

size_t bhh_csv_row(const char *label, float start, float end, float conf, const char *path, char *out, size_t cap) {
    std::string s = csv_row(detection_from_label(label, conf, start, end), path);
    if (s.size() + 1 > cap) return 0;
    memcpy(out, s.c_str(), s.size() + 1);
    return s.size();
}

// output_path_for(csv): <output_dir>/<stem>.BirdNET.results.csv (constants.rs:265; coordinator.rs:63-94)
static std::string csv_output_path(const std::string &input, const std::string &out_dir) {
    size_t slash = input.find_last_of('/');
    std::string name = slash == std::string::npos ? input : input.substr(slash + 1);
    size_t dot = name.find_last_of('.');
    std::string stem = (dot == std::string::npos || dot == 0) ? name : name.substr(0, dot);
    for (char &c : stem) if (c == '/' || c == '\\') c = '_';  // coordinator.rs:55-57
    std::string dir = out_dir.empty() ? (slash == std::string::npos ? "." : input.substr(0, slash)) : out_dir;
    return dir + "/" + stem + ".BirdNET.results.csv";
}

// process_file -- processor.rs:418-796 (CSV format, no lock files / progress bars / reporters)
int bhh_process_file(bh_classifier *clf, const bhh_processing_config *cfg, bhh_process_result *res) {
    if (!clf || !cfg || !cfg->input_path || !res) return hfail(BH_ERR_INVALID, "process_file: null argument");
    memset(res, 0, sizeof *res);
    const auto t_start = std::chrono::steady_clock::now();
    bh_model_info info;
    if (bh_classifier_info(clf, &info) != BH_OK) return hfail(BH_ERR_INVALID, bh_last_error());

    // open once for metadata (:457-459)
    bh_decoder *probe = nullptr;
    int rc = bhh_decoder_open(cfg->input_path, &probe);
    if (rc != BH_OK) return rc;
    const uint32_t source_rate = bhh_decoder_sample_rate(probe);
    double duration = 0.0;
    const int has_duration = bhh_decoder_duration_hint(probe, &duration);
    bhh_decoder_close(probe);

    const uint32_t target_rate = info.sample_rate;          // :474
    const float segment_duration = info.segment_duration;
    const size_t segment_samples = bhh_duration_to_samples(segment_duration, target_rate);   // :514
    const size_t overlap_samples = bhh_duration_to_samples(cfg->overlap, target_rate);       // :520
    const int64_t estimated = bhh_estimate_segment_count(has_duration, duration, segment_duration, cfg->overlap);  // :525
    const size_t batch_size = cfg->batch_size ? cfg->batch_size : 8;
    const size_t effective = bhh_effective_batch_size(batch_size, estimated);                // :531-545
    res->effective_batch = effective;

    if (segment_samples != info.sample_count)
        return hfail(BH_ERR_INVALID, "segment_duration * sample_rate != model sample_count");
    const bool resampling = source_rate != target_rate;   // raw source-rate segments go to the device resampler
    const size_t src_segment_samples = bhh_source_samples(segment_samples, source_rate, target_rate);

    rc = bh_classifier_ensure_warm(clf, effective);                                          // :577
    if (rc != BH_OK) return hfail(rc, bh_last_error());
    bh_batch_context *ctx = nullptr;                                                         // :582-603
    if (effective > 1) {
        rc = bh_batch_context_create(clf, effective, &ctx);
        if (rc != BH_OK) ctx = nullptr;  // fall back to predict_batch like the reference does for Perch
    }

    Channel chan(std::max<size_t>(4, effective * 2));                                        // :640-641
    const std::string path = cfg->input_path;
    const float overlap = cfg->overlap;
    (void)overlap;
    std::thread producer([&chan, path, source_rate, target_rate, segment_samples, overlap_samples] {  // :23-108
        bh_decoder *dec = nullptr;
        int r = bhh_decoder_open(path.c_str(), &dec);                                        // :59
        if (r != BH_OK) { chan.close(r, h_err); return; }
        const size_t src_seg = bhh_source_samples(segment_samples, source_rate, target_rate);   // :67-71
        const size_t src_ovl = bhh_source_samples(overlap_samples, source_rate, target_rate);   // :78-82
        std::vector<float> raw(src_seg);
        size_t start_sample = 0;
        while ((r = bhh_decoder_next_segment(dec, src_seg, src_ovl, raw.data(), &start_sample)) == 1) {  // :84
            AudioChunk c;
            // equal rates: resample_chunk is the identity (resample.rs:98-100) and resize pads (:87);
            // otherwise the raw segment travels on and both steps run on the device (bh_resample_device)
            c.samples.assign(raw.begin(), raw.end());
            if (source_rate == target_rate) c.samples.resize(segment_samples, 0.0f);
            c.start_time = (float)start_sample / (float)source_rate;                         // :91
            const float seg_dur = (float)segment_samples / (float)target_rate;               // :93
            c.end_time = c.start_time + seg_dur;                                             // :94
            if (!chan.send(std::move(c))) break;                                             // :103
        }
        bhh_decoder_close(dec);
        chan.close(r < 0 ? r : 0, r < 0 ? h_err : "");
    });

    // run_streaming_inference -- :114-190
    std::vector<Detection> detections;
    std::vector<AudioChunk> batch;
    batch.reserve(effective);
    size_t segment_count = 0;
    std::vector<float> padding;
    std::vector<bh_result> results(effective);
    std::string fail_msg;
    int fail_code = 0;

    auto process_batch = [&](std::vector<AudioChunk> &b) -> int {                            // :220-410
        const size_t valid = b.size();
        std::vector<const float *> segs;
        for (auto &c : b) segs.push_back(c.samples.data());
        if (valid < effective) {                                                             // :240-258
            if (padding.empty()) padding.assign(resampling ? src_segment_samples : (size_t)info.sample_count, 0.0f);
            res->padded_rows += effective - valid;
            while (segs.size() < effective) segs.push_back(padding.data());
        }
        const size_t bs = segs.size();
        void *guard = bhh_watchdog_start(watchdog_timeout_secs() * 1000, bs);                // :263-266
        int r;
        if (resampling) r = bh_predict_batch_source_rate(clf, ctx, segs.data(), bs, src_segment_samples, source_rate, results.data());
        else if (bs == 1) r = bh_predict(clf, segs[0], info.sample_count, &results[0]);       // :269-277
        else if (ctx) r = bh_predict_batch_with_context(clf, ctx, segs.data(), bs, info.sample_count, results.data());
        else r = bh_predict_batch(clf, segs.data(), bs, info.sample_count, results.data());
        bhh_watchdog_cancel(guard);
        if (r != BH_OK) { fail_msg = std::string("Inference: ") + bh_last_error(); return r; }
        res->batches++;
        for (size_t i = 0; i < valid; i++)                                                   // :363-385
            for (uint32_t k = 0; k < results[i].n_pred; k++)
                if (results[i].confidence[k] >= cfg->min_confidence) {                       // :375
                    const char *label = bh_classifier_label(clf, (uint32_t)results[i].index[k]);
                    detections.push_back(detection_from_label(label ? label : std::to_string(results[i].index[k]),
                                                              results[i].confidence[k], b[i].start_time, b[i].end_time));
                }
        return BH_OK;
    };

    AudioChunk chunk;
    int r;
    while ((r = chan.recv(chunk)) == 1) {                                                    // :132-155
        batch.push_back(std::move(chunk));
        segment_count++;
        if (batch.size() >= effective) {
            if ((fail_code = process_batch(batch)) != BH_OK) break;
            batch.clear();
        }
    }
    if (r < 0 && !fail_code) { fail_code = r; fail_msg = chan.err; }
    if (!fail_code && !batch.empty()) fail_code = process_batch(batch);                      // :158-174
    chan.drop_receiver();
    producer.join();                                                                         // :676
    if (ctx) bh_batch_context_destroy(ctx);
    if (fail_code) return hfail(fail_code, fail_msg);

    // sort: start_time asc then confidence desc (:178-187); stable here (ties keep batch order)
    std::stable_sort(detections.begin(), detections.end(), [](const Detection &a, const Detection &b) {
        if (a.start_time < b.start_time) return true;
        if (a.start_time > b.start_time) return false;
        return a.confidence > b.confidence;
    });

    // write_output(csv) -- :819-873
    const std::string out_path = csv_output_path(path, cfg->output_dir ? cfg->output_dir : "");
    FILE *o = fopen(out_path.c_str(), "wb");
    if (!o) return hfail(BH_ERR_IO, "cannot create " + out_path);
    if (cfg->csv_bom) fwrite("\xEF\xBB\xBF", 1, 3, o);
    fputs(CSV_HEADER, o);
    const std::string shown = cfg->display_path ? cfg->display_path : path;
    for (const auto &d : detections) { const std::string row = csv_row(d, shown); fwrite(row.data(), 1, row.size(), o); }
    fclose(o);

    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    res->detections = detections.size();
    res->segments = segment_count;
    res->duration_secs = wall;
    res->audio_duration_secs = has_duration ? duration
                               : (segment_count ? (double)segment_duration + (segment_count - 1.0) * ((double)segment_duration - cfg->overlap) : 0.0);  // :692-703
    res->segments_per_sec = wall > 0 && segment_count ? (double)segment_count / wall : 0.0;  // :771-778
    snprintf(res->output_path, sizeof res->output_path, "%s", out_path.c_str());
    return BH_OK;
}

}  // extern "C"


// ---- range filter tables (reference src/inference/geomodel.rs) ---------------------------------------------
extern "C" BH_API size_t bhh_scientific_name_len(const char *label) {
    if (!label) return 0;
    const char *us = std::strchr(label, '_');
    if (!us) return std::strlen(label);
    // the prefix is a scientific name only when it contains a space (binomials do, FSD50K classes do not): :28-33
    return std::memchr(label, ' ', (size_t)(us - label)) ? (size_t)(us - label) : std::strlen(label);
}

static std::string species_key(const char *label) {   // :36-38
    std::string k(label, bhh_scientific_name_len(label));
    for (char &ch : k)
        if (ch >= 'A' && ch <= 'Z') ch = (char)(ch + 32);
    return k;
}

extern "C" BH_API int bhh_project_scores(const char *const *geomodel_labels, size_t n_geomodel, const char *const *score_species,
                                         const float *score_values, size_t n_scores, const char *const *classifier_labels,
                                         size_t n_classes, float threshold, float *out_scores, size_t *mapped, size_t *in_range) {
    if ((n_geomodel && !geomodel_labels) || (n_scores && (!score_species || !score_values)) || (n_classes && (!classifier_labels || !out_scores))) {
        return hfail(BH_ERR_INVALID, "project_scores: null argument");
    }
    std::unordered_map<std::string, size_t> cls_by_key;   // first classifier label of a scientific name wins (:62-74)
    cls_by_key.reserve(n_classes);
    for (size_t c = 0; c < n_classes; c++) {
        cls_by_key.emplace(species_key(classifier_labels[c]), c);
        out_scores[c] = std::numeric_limits<float>::quiet_NaN();
    }
    size_t n_mapped = 0;
    for (size_t g = 0; g < n_geomodel; g++) {              // mapped species start at 0 = "out of range" (:146-149)
        auto it = cls_by_key.find(species_key(geomodel_labels[g]));
        if (it != cls_by_key.end() && std::isnan(out_scores[it->second])) { out_scores[it->second] = 0.0f; n_mapped++; }
    }
    for (size_t i = 0; i < n_scores; i++) {                // :151-158; species outside the mapping are dropped
        auto it = cls_by_key.find(species_key(score_species[i]));
        if (it != cls_by_key.end() && !std::isnan(out_scores[it->second])) out_scores[it->second] = score_values[i];
    }
    if (mapped) *mapped = n_mapped;
    if (in_range) {                                        // :173-178
        size_t n = 0;
        for (size_t c = 0; c < n_classes; c++)
            if (out_scores[c] >= threshold) n++;
        *in_range = n;
    }
    return BH_OK;
}


// ---- directory mode (reference src/pipeline/coordinator.rs:146-190) ------------------------------------------
#include <dirent.h>
#include <sys/stat.h>

extern "C" BH_API int bhh_is_audio_file(const char *path) {   // :179-190
    if (!path) return 0;
    const char *base = std::strrchr(path, '/');
    base = base ? base + 1 : path;
    const char *dot = std::strrchr(base, '.');
    if (!dot || dot == base) return 0;                         // Path::extension(): ".wav" alone has none
    std::string ext(dot + 1);
    for (char &ch : ext)
        if (ch >= 'A' && ch <= 'Z') ch = (char)(ch + 32);
    for (const char *e : {"wav", "flac", "mp3", "m4a", "aac"})
        if (ext == e) return 1;
    return 0;
}

static bool walk_audio_files(const std::string &dir, std::vector<std::string> &files) {   // :164-176
    DIR *d = opendir(dir.c_str());
    if (!d) return false;
    std::vector<std::string> names;
    while (dirent *e = readdir(d)) {
        if (!std::strcmp(e->d_name, ".") || !std::strcmp(e->d_name, "..")) continue;
        names.emplace_back(e->d_name);
    }
    closedir(d);
    std::sort(names.begin(), names.end());
    for (const std::string &n : names) {
        const std::string p = dir + "/" + n;
        struct stat st;
        if (stat(p.c_str(), &st) != 0) continue;
        if (S_ISDIR(st.st_mode)) {
            if (!walk_audio_files(p, files)) return false;
        } else if (bhh_is_audio_file(p.c_str())) {
            files.push_back(p);
        }
    }
    return true;
}

extern "C" BH_API size_t bhh_collect_input_files(const char *const *paths, size_t n_paths, char *out, size_t cap, size_t *n_files) {
    std::vector<std::string> files;
    for (size_t i = 0; i < n_paths; i++) {                    // :146-161
        struct stat st;
        if (!paths || !paths[i] || stat(paths[i], &st) != 0) continue;   // "Skipping non-existent path"
        if (S_ISREG(st.st_mode)) {
            if (bhh_is_audio_file(paths[i])) files.emplace_back(paths[i]);
        } else if (S_ISDIR(st.st_mode)) {
            if (!walk_audio_files(paths[i], files)) { hfail(BH_ERR_IO, std::string("cannot read directory ") + paths[i]); return (size_t)-1; }
        }
    }
    std::string joined;
    for (size_t i = 0; i < files.size(); i++) { if (i) joined += '\n'; joined += files[i]; }
    if (n_files) *n_files = files.size();
    if (out && cap > joined.size()) std::memcpy(out, joined.c_str(), joined.size() + 1);
    return joined.size() + 1;
}

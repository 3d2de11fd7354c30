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
#include <pthread.h>
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

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "host_internal.hpp"

#if defined(__x86_64__)
#include <immintrin.h>
#endif

// Pageable caller memory -> the context's pinned staging buffer (bh_predict_batch*, bh_predict_pcm*: api.hip).  Non-temporal
// stores where the host has AVX2: the staged bytes are read next by the DMA engine, not by a core, and a streaming store neither
// fetches the destination line first nor evicts the source's neighbours -- 8 threads gather 295 MB (512 segments) in 1.8 ms
// against 2.7 ms with glibc's memcpy on the pool's hosts (tools/microbench/host_gather.cpp, profiles/r5_*_host_gather.txt).
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void stream_copy_avx2(char *dst, const char *src, size_t n) {
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(src + i)), b = _mm256_loadu_si256((const __m256i *)(src + i + 32)),
                      c = _mm256_loadu_si256((const __m256i *)(src + i + 64)), d = _mm256_loadu_si256((const __m256i *)(src + i + 96));
        _mm256_stream_si256((__m256i *)(dst + i), a); _mm256_stream_si256((__m256i *)(dst + i + 32), b);
        _mm256_stream_si256((__m256i *)(dst + i + 64), c); _mm256_stream_si256((__m256i *)(dst + i + 96), d);
    }
    _mm_sfence();
    if (i < n) memcpy(dst + i, src + i, n - i);
}
#endif
extern "C" void bh_internal_stream_copy(void *dst, const void *src, size_t bytes) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && bytes >= 4096) {
        char *d = static_cast<char *>(dst);
        const char *s = static_cast<const char *>(src);
        const size_t head = (32 - (reinterpret_cast<uintptr_t>(d) & 31)) & 31;     // streaming stores want a 32-byte aligned destination
        if (head) { memcpy(d, s, head); d += head; s += head; bytes -= head; }
        stream_copy_avx2(d, s, bytes);
        return;
    }
#endif
    memcpy(dst, src, bytes);
}

using bhh::Detection;
using bhh::detection_from_label;
using bhh::escape_csv;

namespace {

thread_local std::string h_err;
int hfail(int code, const std::string &msg) { h_err = msg; return code; }
}  // namespace
void bhh::set_host_error(const std::string &msg) { h_err = msg; }
namespace {
// catch (...) handler of the extern "C" entry points: exceptions never cross the C ABI
int h_on_exception() noexcept {
    try {
        try { throw; }
        catch (const std::bad_alloc &) { return hfail(BH_ERR_INTERNAL, "out of host memory"); }
        catch (const std::exception &e) { return hfail(BH_ERR_INTERNAL, std::string("internal error: ") + e.what()); }
        catch (...) { return hfail(BH_ERR_INTERNAL, "internal error (unknown exception)"); }
    } catch (...) { return BH_ERR_INTERNAL; }   // building the message failed as well
}

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
    uint64_t data_bytes = 0, data_read = 0, data_offset = 0;
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
int bhh_decoder_open(const char *path, bh_decoder **out) try {
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
            // the chunk size comes from an untrusted header: read the 40 bytes WAVE_FORMAT_EXTENSIBLE defines at most
            unsigned char b[40] = {0};
            const uint32_t take = sz < sizeof b ? sz : (uint32_t)sizeof b;
            if (sz < 16 || fread(b, 1, take, f) != take) break;
            if (sz > take && fseek(f, (long)(sz - take), SEEK_CUR) != 0) break;
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
            d->data_offset = (uint64_t)pos;
            fseek(f, 0, SEEK_END);
            long end = ftell(f);
            fseek(f, pos, SEEK_SET);
            uint64_t avail = (uint64_t)(end - pos);
            // a streaming writer leaves 0xFFFFFFFF (or a length past the end of the file): take what is there.
            // A zero-length data chunk is an empty stream, not "to end of file".
            d->data_bytes = (sz == 0xFFFFFFFFu || sz > avail) ? avail : sz;
            have_data = true;
        } else {
            if (fseek(f, (long)sz + (sz & 1), SEEK_CUR) != 0) break;
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
} catch (...) { return h_on_exception(); }

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
                             size_t *start_sample) try {
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
} catch (...) { return h_on_exception(); }

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

std::string csv_row(const Detection &d, const std::string &path) {  // csv.rs:55-66, DECIMAL_PLACES = 4 (bhh_csv_row; files go through host_output.cpp)
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
// The reference spawns one detached thread per batch that sleeps the whole timeout whether or not the guard was
// dropped; at ORT batch rates that is a handful of threads.  A batch takes well under a millisecond here, so ONE
// process-wide thread serves every guard: start arms a deadline, cancel disarms it and wakes the thread, which
// terminates the process (exit code 1, the reference's message) only for a deadline that passes while still armed.
struct WatchdogHub {
    struct Armed { std::chrono::steady_clock::time_point deadline; uint64_t timeout_ms; size_t batch_size; };
    std::mutex mu;
    std::condition_variable cv;
    std::unordered_map<uint64_t, Armed> armed;
    uint64_t next_id = 1;
    bool thread_started = false;

    void loop() {
        std::unique_lock<std::mutex> l(mu);
        for (;;) {
            if (armed.empty()) { cv.wait(l); continue; }
            auto first = armed.begin();
            for (auto it = armed.begin(); it != armed.end(); ++it)
                if (it->second.deadline < first->second.deadline) first = it;
            const Armed a = first->second;
            if (std::chrono::steady_clock::now() >= a.deadline) {
                const size_t suggested = std::max<size_t>(a.batch_size / 2, 1);
                fprintf(stderr, "\nFATAL: Inference timeout after %llus (batch size: %zu)\n\n"
                                "The GPU inference operation did not complete within the expected time.\n"
                                "Recommendations:\n  1. Reduce batch size: birda -b %zu <input>\n"
                                "  2. Close other GPU applications and try again\n\n"
                                "Terminating process to prevent system lockup.\n",
                        (unsigned long long)(a.timeout_ms / 1000), a.batch_size, suggested);
                _Exit(1);
            }
            cv.wait_until(l, a.deadline);
        }
    }
    uint64_t arm(uint64_t timeout_ms, size_t batch_size) {
        std::lock_guard<std::mutex> l(mu);
        if (!thread_started) {   // created once; never per batch.  A bare pthread: std::thread's heap-allocated state block of a thread
            pthread_t th;        // that never ends is what LeakSanitizer reports once the process has created other threads before it
            if (pthread_create(&th, nullptr, [](void *p) -> void * { static_cast<WatchdogHub *>(p)->loop(); return nullptr; }, this) == 0) pthread_detach(th);
            thread_started = true;
        }
        const uint64_t id = next_id++;
        armed[id] = Armed{std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms), timeout_ms, batch_size};
        cv.notify_all();
        return id;
    }
    void disarm(uint64_t id) {
        std::lock_guard<std::mutex> l(mu);
        armed.erase(id);
        cv.notify_all();
    }
};
WatchdogHub &watchdog_hub() {
    static WatchdogHub *hub = new WatchdogHub();   // never destroyed: the detached thread may outlive static destructors
    return *hub;
}

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
// bhh_watchdog_cancel == drop(guard) (:61-65).  NULL when the guard could not be armed (out of memory / threads):
// the batch then runs unguarded instead of taking the process down.
void *bhh_watchdog_start(uint64_t timeout_ms, size_t batch_size) {
    try {
        return new uint64_t(watchdog_hub().arm(timeout_ms, batch_size));
    } catch (...) {
        return nullptr;
    }
}
void bhh_watchdog_cancel(void *guard) {
    auto *id = static_cast<uint64_t *>(guard);
    if (!id) return;
    watchdog_hub().disarm(*id);
    delete id;
}

size_t bhh_csv_header(int bom, char *out, size_t cap) try {
    std::string s;
    if (bom) s = "\xEF\xBB\xBF";  // UTF8_BOM, constants.rs:437
    s += CSV_HEADER;
    if (s.size() + 1 > cap) return 0;
    memcpy(out, s.c_str(), s.size() + 1);
    return s.size();
} catch (...) { return (h_on_exception(), (size_t)0); }

size_t bhh_csv_row(const char *label, float start, float end, float conf, const char *path, char *out, size_t cap) try {
    std::string s = csv_row(detection_from_label(label, conf, start, end, path ? path : ""), path ? path : "");
    if (s.size() + 1 > cap) return 0;
    memcpy(out, s.c_str(), s.size() + 1);
    return s.size();
} catch (...) { return (h_on_exception(), (size_t)0); }

}  // extern "C"

// ---- process_file -- processor.rs:418-796 ---------------------------------------------------------------
namespace {

struct FilePlan {
    std::string path, shown;
    uint32_t source_rate = 0, target_rate = 0;
    float segment_duration = 0.f;
    size_t segment_samples = 0, overlap_samples = 0, src_segment_samples = 0, src_overlap_samples = 0;
    size_t effective = 0;
    int64_t estimated = -1;
    bool resampling = false;
    float min_confidence = 0.1f;
    bhh_reporter *reporter = nullptr;
    bh_custom_classifier *custom = nullptr;   // bat mode
};

// (start_sample / source_rate as f32, start + segment_samples / target_rate) -- processor.rs:91-94
void chunk_times(const FilePlan &pl, size_t start_sample, float &start_time, float &end_time) {
    start_time = (float)start_sample / (float)pl.source_rate;
    end_time = start_time + (float)pl.segment_samples / (float)pl.target_rate;
}

// threshold + Detection::from_label for one segment's predictions -- processor.rs:363-385
void collect_detections(bh_classifier *clf, const FilePlan &pl, const bh_result &r, float start_time, float end_time,
                        std::vector<Detection> &detections) {
    for (uint32_t k = 0; k < r.n_pred; k++)
        if (r.confidence[k] >= pl.min_confidence) {                                          // :375
            // bat mode: the predictions (and their labels) are the custom classifier's (:369-372)
            const char *label = pl.custom ? bh_custom_classifier_label(pl.custom, (uint32_t)r.index[k]) : bh_classifier_label(clf, (uint32_t)r.index[k]);
            detections.push_back(detection_from_label(label ? label : std::to_string(r.index[k]), r.confidence[k], start_time, end_time, pl.shown));
        }
}

// reporter.progress(None, Some(file_progress)) after every segment -- processor.rs:388-406
void report_progress(const FilePlan &pl, size_t &segments_done) {
    if (!pl.reporter) return;
    segments_done++;
    const size_t total = pl.estimated > 0 ? (size_t)pl.estimated : 0;   // unwrap_or(0)
    const float percent = total > 0 ? std::min((float)segments_done / (float)total * 100.0f, 100.0f) : 0.0f;
    bhh_reporter_file_progress(pl.reporter, pl.path.c_str(), segments_done, total, percent);
}

struct RunStats { size_t segments = 0, batches = 0, padded_rows = 0; };

// HOST front end: decode thread -> bounded channel -> batcher (processor.rs:23-190, 220-410)
int run_host_front_end(bh_classifier *clf, const FilePlan &pl, const bh_model_info &info, bh_batch_context *ctx,
                       std::vector<Detection> &detections, RunStats &st, std::string &fail_msg) {
    const size_t effective = pl.effective;
    Channel chan(std::max<size_t>(4, effective * 2));                                        // :640-641
    std::thread producer([&chan, &pl] {                                                      // :23-108
        try {
            bh_decoder *dec = nullptr;
            int r = bhh_decoder_open(pl.path.c_str(), &dec);                                 // :59
            if (r != BH_OK) { chan.close(r, h_err); return; }
            std::vector<float> raw(pl.src_segment_samples);
            size_t start_sample = 0;
            while ((r = bhh_decoder_next_segment(dec, pl.src_segment_samples, pl.src_overlap_samples, raw.data(), &start_sample)) == 1) {  // :84
                AudioChunk c;
                // equal rates: resample_chunk is the identity (resample.rs:98-100) and resize pads (:87);
                // otherwise the raw segment travels on and both steps run on the device (bh_resample_device)
                c.samples.assign(raw.begin(), raw.end());
                if (!pl.resampling) c.samples.resize(pl.segment_samples, 0.0f);
                chunk_times(pl, start_sample, c.start_time, c.end_time);
                if (!chan.send(std::move(c))) break;                                         // :103
            }
            bhh_decoder_close(dec);
            chan.close(r < 0 ? r : 0, r < 0 ? h_err : "");
        } catch (...) {   // an exception must not leave the thread (std::terminate)
            chan.close(BH_ERR_INTERNAL, "decode thread: out of memory or internal error");
        }
    });

    std::vector<AudioChunk> batch;
    batch.reserve(effective);
    std::vector<float> padding;
    std::vector<bh_result> results(effective);
    int fail_code = 0;
    size_t segments_done = 0;

    auto process_batch = [&](std::vector<AudioChunk> &b) -> int {                            // :220-410
        const size_t valid = b.size();
        std::vector<const float *> segs;
        for (auto &c : b) segs.push_back(c.samples.data());
        if (valid < effective) {                                                             // :240-258
            if (padding.empty()) padding.assign(pl.resampling ? pl.src_segment_samples : (size_t)info.sample_count, 0.0f);
            st.padded_rows += effective - valid;
            while (segs.size() < effective) segs.push_back(padding.data());
        }
        const size_t bs = segs.size();
        void *guard = bhh_watchdog_start(watchdog_timeout_secs() * 1000, bs);                // :263-266
        int r;
        if (pl.custom) r = bh_predict_batch_two_stage(clf, ctx, pl.custom, segs.data(), bs, info.sample_count, results.data(), nullptr);   // :319-360
        else if (pl.resampling) r = bh_predict_batch_source_rate(clf, ctx, segs.data(), bs, pl.src_segment_samples, pl.source_rate, results.data());
        else if (bs == 1) r = bh_predict(clf, segs[0], info.sample_count, &results[0]);       // :269-277
        else if (ctx) r = bh_predict_batch_with_context(clf, ctx, segs.data(), bs, info.sample_count, results.data());
        else r = bh_predict_batch(clf, segs.data(), bs, info.sample_count, results.data());
        bhh_watchdog_cancel(guard);
        if (r != BH_OK) { fail_msg = std::string("Inference: ") + bh_last_error(); return r; }
        st.batches++;
        for (size_t i = 0; i < valid; i++) {                                                 // :363-385
            collect_detections(clf, pl, results[i], b[i].start_time, b[i].end_time, detections);
            report_progress(pl, segments_done);
        }
        return BH_OK;
    };

    AudioChunk chunk;
    int r;
    try {
        while ((r = chan.recv(chunk)) == 1) {                                                // :132-155
            batch.push_back(std::move(chunk));
            st.segments++;
            if (batch.size() >= effective) {
                if ((fail_code = process_batch(batch)) != BH_OK) break;
                batch.clear();
            }
        }
        if (r < 0 && !fail_code) { fail_code = r; fail_msg = chan.err; }
        if (!fail_code && !batch.empty()) fail_code = process_batch(batch);                  // :158-174
    } catch (...) {
        chan.drop_receiver();
        producer.join();
        throw;
    }
    chan.drop_receiver();
    producer.join();                                                                         // :676
    return fail_code;
}

// DEVICE front end: the PCM16 frames of the file are mapped and handed to bh_predict_pcm16 span by span.
// A span starts on a multiple of the segment step, so its windows are the stream's windows (decode.rs:150-202); all
// but the last span end on a full segment, whose trailing overlap remainder belongs to the next span and is dropped here.
int run_device_front_end(bh_classifier *clf, const FilePlan &pl, const bh_model_info &info, bh_batch_context *ctx, const unsigned char *pcm,
                         uint32_t fmt, size_t bps, size_t n_frames, uint32_t channels, std::vector<Detection> &detections, RunStats &st,
                         std::string &fail_msg, int fd = -1, uint64_t data_offset = 0) {
    (void)info;
    // BIRDA_HOST_PREAD=1: the file's bytes by pread into the pinned staging buffer (bh_predict_pcm_fd_rows, round 6) instead of a gather
    // from the mapping.  Built for VERDICT r5 next #8 and measured: no gain -- WAV -> CSV 96.6 / 90.4 k segments/s by descriptor against
    // 98.0 / 99.3 k mapped (0.63-0.68 against 0.64 of the same run's device-resident rate, profiles/r6_k_pread.txt): as round 5's
    // MADV_POPULATE_READ probe said, the page faults are not what the file path costs; the inference under it is.  Off by default.
    static const bool pread_on = [] { const char *e = getenv("BIRDA_HOST_PREAD"); return e && e[0] == '1'; }();
    bool use_fd = fd >= 0 && pread_on;
    const size_t seg = pl.src_segment_samples, ovl = pl.src_overlap_samples;
    if (ovl >= seg) {   // next_segment's check, decode.rs:156-162
        fail_msg = "overlap_samples (" + std::to_string(ovl) + ") must be less than segment_samples (" + std::to_string(seg) + ")";
        return BH_ERR_INVALID;
    }
    const size_t step = seg - ovl;
    const size_t span_segments = std::max<size_t>(pl.effective, 4096 / pl.effective * pl.effective);   // whole batches, ~4096 segments
    std::vector<bh_result> results(span_segments + 1);
    std::vector<uint64_t> starts(span_segments + 1);
    size_t segments_done = 0;
    for (size_t f0 = 0; f0 < n_frames;) {
        const size_t full_span = (span_segments - 1) * step + seg;
        const bool last = n_frames - f0 <= full_span;
        const size_t frames = last ? n_frames - f0 : full_span;
        size_t n_seg = 0;
        // process_batch's per-batch work (threshold, Detection::from_label, progress; processor.rs:363-407) runs in the rows
        // callback: on this thread, for each finished run of segments, while the device computes the later ones
        struct Sink {
            bh_classifier *clf; const FilePlan *pl; std::vector<Detection> *detections; size_t f0, limit; size_t *segments_done;
            static void rows(void *user, size_t first, size_t n, const bh_result *r, const uint64_t *st) {
                Sink &s = *static_cast<Sink *>(user);
                for (size_t i = 0; i < n && first + i < s.limit; i++) {
                    float t0, t1;
                    chunk_times(*s.pl, s.f0 + (size_t)st[i], t0, t1);
                    collect_detections(s.clf, *s.pl, r[i], t0, t1, *s.detections);
                    report_progress(*s.pl, *s.segments_done);
                }
            }
        } sink{clf, &pl, &detections, f0, last ? (size_t)-1 : span_segments, &segments_done};
        void *guard = bhh_watchdog_start(watchdog_timeout_secs() * 1000, std::min(pl.effective, span_segments));
        int r;
        try {
            r = BH_ERR_UNSUPPORTED;
            if (use_fd) {
                r = bh_predict_pcm_fd_rows(clf, ctx, fd, data_offset + (uint64_t)f0 * channels * bps, fmt, frames, channels, pl.source_rate, pl.overlap_samples,
                                           results.data(), results.size(), &n_seg, starts.data(), &Sink::rows, &sink);
                if (r == BH_ERR_UNSUPPORTED) use_fd = false;     // (a slice wider than the staging buffer: nothing was delivered; the mapped route)
            }
            if (!use_fd)
            r = bh_predict_pcm_rows(clf, ctx, pcm + f0 * channels * bps, fmt, frames, channels, pl.source_rate, pl.overlap_samples, results.data(),
                                    results.size(), &n_seg, starts.data(), &Sink::rows, &sink);
        } catch (...) { bhh_watchdog_cancel(guard); throw; }
        bhh_watchdog_cancel(guard);
        if (r != BH_OK) { fail_msg = std::string("Inference: ") + bh_last_error(); return r; }
        const size_t keep = last ? n_seg : std::min(n_seg, span_segments);
        st.segments += keep;
        st.batches += (keep + pl.effective - 1) / pl.effective;
        if (last) break;
        f0 += span_segments * step;
    }
    return BH_OK;
}

// a read-only mapping of a WAV file's PCM16 data chunk
struct PcmMapping {
    void *base = MAP_FAILED;
    size_t length = 0;
    const unsigned char *pcm = nullptr;   // the data chunk: interleaved samples in the file's own layout
    uint32_t fmt = 0;                     // BH_PCM_*
    size_t bps = 0;                       // bytes per sample
    size_t n_frames = 0;
    int fd = -1;                          // kept open: bh_predict_pcm_fd_rows reads the data chunk straight into pinned staging
    uint64_t data_offset = 0;
    ~PcmMapping() { if (base != MAP_FAILED) munmap(base, length); if (fd >= 0) ::close(fd); }
    bool open(const bh_decoder &d) {
        fmt = d.fmt == FMT_S16 ? BH_PCM_S16 : d.fmt == FMT_S24 ? BH_PCM_S24 : d.fmt == FMT_S32 ? BH_PCM_S32 : d.fmt == FMT_F32 ? BH_PCM_F32 : 0;
        bps = fmt == BH_PCM_S16 ? 2 : fmt == BH_PCM_S24 ? 3 : 4;
        if (!fmt) return false;            // (8-bit PCM and everything symphonia decodes stay on the host front end)
        fd = ::open(d.path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        data_offset = d.data_offset;
        length = (size_t)(d.data_offset + d.data_bytes);
        // (no MAP_POPULATE: the minor faults are taken inside the upload workers' copies, eight threads at once; populating
        //  the 70 000 pages of a 1 000-segment file here, on one thread, was slower end to end)
        // (the mapping costs nothing until it is touched: the route of streams wider than the staging buffer, and of BIRDA_HOST_PREAD=0)
        if (d.data_bytes) base = mmap(nullptr, length, PROT_READ, MAP_PRIVATE, fd, 0);
        if (d.data_bytes && base == MAP_FAILED) return false;
        if (base != MAP_FAILED) (void)madvise(base, length, MADV_SEQUENTIAL);
        pcm = base != MAP_FAILED ? static_cast<const unsigned char *>(base) + d.data_offset : nullptr;
        n_frames = (size_t)(d.data_bytes / ((uint64_t)d.channels * bps));
        return true;
    }
};

// the rest of process_file once the detections exist: sort (:178-187), audio duration (:692-703), write_output per format
// (:721-736) or the reporter's detections event (:739-769), the result record (:771-788)
template <class Lap>
int finish_file(const bhh_processing_config *cfg, const FilePlan &pl, uint32_t formats, std::vector<Detection> &detections, const RunStats &st,
                int has_duration, double duration, float overlap_secs, std::chrono::steady_clock::time_point t_start, bhh_process_result *res,
                Lap &&lap) {
    int rc = BH_OK;
    // sort: start_time asc then confidence desc (:178-187); stable here (ties keep batch order)
    // (the segments arrive in time order and a segment's predictions by descending confidence, so the list is usually sorted
    //  already: one pass to see that instead of a merge sort's buffer and moves, 0.24 -> 0.02 ms for 5 000 detections)
    const auto before = [](const Detection &a, const Detection &b) {
        if (a.start_time < b.start_time) return true;
        if (a.start_time > b.start_time) return false;
        return a.confidence > b.confidence;
    };
    if (!std::is_sorted(detections.begin(), detections.end(), before)) std::stable_sort(detections.begin(), detections.end(), before);

    lap("sort");
    const double audio_duration = has_duration ? duration
                                  : (st.segments ? (double)pl.segment_duration + (st.segments - 1.0) * ((double)pl.segment_duration - overlap_secs) : 0.0);  // :692-703

    // write_output per format (:721-736) unless a reporter owns stdout-only mode
    const bool should_write = cfg->dual_output || !cfg->reporter;
    if (should_write) {
        bhh::WriterOptions wo;
        wo.csv_bom = cfg->csv_bom != 0;
        if (cfg->csv_columns) wo.csv_columns = cfg->csv_columns;
        if (cfg->model_name) wo.model = cfg->model_name;
        wo.min_confidence = cfg->min_confidence; wo.overlap = cfg->overlap;
        wo.audio_duration = (float)audio_duration;                                           // :706-714
        wo.has_lat = cfg->has_lat != 0; wo.has_lon = cfg->has_lon != 0; wo.lat = cfg->lat; wo.lon = cfg->lon; wo.week = cfg->week;
        for (uint32_t bit = 1; bit <= BHH_FORMAT_PARQUET; bit <<= 1) {
            if (!(formats & bit)) continue;
            std::string out_path, err;
            rc = bhh::write_output(pl.path, cfg->output_dir ? cfg->output_dir : "", bit, detections, wo, out_path, err);
            if (rc != BH_OK) return hfail(rc, err);
            if (!res->formats_written) snprintf(res->output_path, sizeof res->output_path, "%s", out_path.c_str());
            res->formats_written |= bit;
        }
    }
    if (!cfg->dual_output && cfg->reporter) bhh::reporter_detections(cfg->reporter, pl.path, detections, cfg->bsg);   // :739-769
    lap("write");

    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    res->detections = detections.size();
    res->segments = st.segments;
    res->batches = st.batches;
    res->padded_rows = st.padded_rows;
    res->duration_secs = wall;
    res->audio_duration_secs = audio_duration;
    res->segments_per_sec = wall > 0 && st.segments ? (double)st.segments / wall : 0.0;      // :771-778
    return BH_OK;
}

}  // namespace

extern "C" int bhh_process_file(bh_classifier *clf, const bhh_processing_config *cfg, bhh_process_result *res) try {
    if (!clf || !cfg || !cfg->input_path || !res) return hfail(BH_ERR_INVALID, "process_file: null argument");
    memset(res, 0, sizeof *res);
    const auto t_start = std::chrono::steady_clock::now();
    const bool timing = getenv("BIRDA_HOST_TIMING") != nullptr;   // diagnostic: phase times of this call to stderr
    auto lap = [&, last = t_start](const char *what) mutable {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "process_file %-10s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    };
    bh_model_info info;
    if (bh_classifier_info(clf, &info) != BH_OK) return hfail(BH_ERR_INVALID, bh_last_error());
    const uint32_t formats = cfg->formats ? cfg->formats : BHH_FORMAT_CSV;
    if (formats & ~BHH_FORMAT_ALL) return hfail(BH_ERR_INVALID, "InvalidOutputFormat: unknown bit in the format mask");
    if (cfg->front_end > BHH_FRONT_END_DEVICE) return hfail(BH_ERR_INVALID, "process_file: unknown front_end");

    // open once for metadata (:457-459)
    bh_decoder *probe = nullptr;
    int rc = bhh_decoder_open(cfg->input_path, &probe);
    if (rc != BH_OK) return rc;
    std::unique_ptr<bh_decoder, void (*)(bh_decoder *)> probe_own(probe, bhh_decoder_close);
    FilePlan pl;
    pl.path = cfg->input_path;
    pl.shown = cfg->display_path ? cfg->display_path : pl.path;
    pl.source_rate = bhh_decoder_sample_rate(probe);
    double duration = 0.0;
    const int has_duration = bhh_decoder_duration_hint(probe, &duration);

    pl.custom = cfg->custom_classifier;
    const bool bat_mode = pl.custom != nullptr;                                              // lib.rs:774
    float overlap_secs = cfg->overlap;
    if (bat_mode) {
        // no resampling: the model is fed source samples as if they were 48 kHz audio (:464-475); fixed 144 000-sample
        // segments overlapping by a quarter (:502-508); durations derive from bat::SAMPLE_RATE = 256 kHz (constants.rs:525-542)
        constexpr uint32_t BAT_SAMPLE_RATE = 256000;
        constexpr size_t BAT_CHUNK_SAMPLES = 144000;
        if (pl.source_rate != BAT_SAMPLE_RATE)
            fprintf(stderr, "WARN Bat mode expects %ukHz audio, source is %ukHz. Results may be unreliable.\n", BAT_SAMPLE_RATE / 1000, pl.source_rate / 1000);
        pl.target_rate = pl.source_rate;
        pl.segment_duration = (float)BAT_CHUNK_SAMPLES / (float)BAT_SAMPLE_RATE;
        pl.segment_samples = BAT_CHUNK_SAMPLES;
        pl.overlap_samples = BAT_CHUNK_SAMPLES / 4;
        overlap_secs = pl.segment_duration * 0.25f;                                          // bat::OVERLAP (lib.rs:734-738)
    } else {
        pl.target_rate = info.sample_rate;                                                   // :474
        pl.segment_duration = info.segment_duration;
        pl.segment_samples = bhh_duration_to_samples(pl.segment_duration, pl.target_rate);   // :514
        pl.overlap_samples = bhh_duration_to_samples(cfg->overlap, pl.target_rate);          // :520
    }
    pl.estimated = bhh_estimate_segment_count(has_duration, duration, pl.segment_duration, overlap_secs);  // :525
    const size_t batch_size = cfg->batch_size ? cfg->batch_size : bh_classifier_default_batch_size(clf);   // lib.rs:1035
    pl.effective = bhh_effective_batch_size(batch_size, pl.estimated);                       // :531-545
    pl.min_confidence = cfg->min_confidence;
    pl.reporter = cfg->reporter;
    res->effective_batch = pl.effective;
    if (pl.segment_samples != info.sample_count)
        return hfail(BH_ERR_INVALID, "segment_duration * sample_rate != model sample_count");
    pl.resampling = pl.source_rate != pl.target_rate;   // raw source-rate segments go to the device resampler
    // (a header may name any rate: the resampler is asked BEFORE a segment of that many source samples is sized, decoded or staged --
    //  a 1.5 GHz header had the host front end gather 4.7 G samples per segment first; tools/fuzz_wav_decoder.py)
    if (pl.resampling && bh_resample_supported(clf, pl.source_rate, pl.target_rate) != BH_OK)
        return hfail(BH_ERR_UNSUPPORTED, std::string("AudioDecode: ") + cfg->input_path + ": " + bh_last_error());
    pl.src_segment_samples = bhh_source_samples(pl.segment_samples, pl.source_rate, pl.target_rate);   // :67-71
    pl.src_overlap_samples = bhh_source_samples(pl.overlap_samples, pl.source_rate, pl.target_rate);   // :78-82

    // front end: the device takes PCM16 WAV (any channel count); everything else decodes on the host
    PcmMapping map;
    bool device = false;
    if (bat_mode && cfg->front_end == BHH_FRONT_END_DEVICE)
        return hfail(BH_ERR_UNSUPPORTED, "process_file: bat mode (custom classifier) runs on the host front end");
    if (cfg->front_end != BHH_FRONT_END_HOST && !bat_mode) {
        device = map.open(*probe);
        if (!device && cfg->front_end == BHH_FRONT_END_DEVICE)
            return hfail(BH_ERR_UNSUPPORTED, "process_file: the device front end takes PCM16 / PCM24 / PCM32 / float32 WAV files only: " + pl.path);
    }
    const uint32_t channels = (uint32_t)probe->channels;
    probe_own.reset();
    if (device && cfg->batch_size == 0) {
        // No batch size asked for: the device front end slices the stream by 1 024 segments (4 GB of context), not by the 256 of
        // determine_default_batch_size -- its rows are never padded, results do not depend on the slice, and the upload of one
        // slice overlaps the compute of the previous sub-slice only within a slice (bh_predict_pcm16)
        pl.effective = bhh_effective_batch_size(1024, pl.estimated);
        res->effective_batch = pl.effective;
    }

    lap("open");
    rc = bh_classifier_ensure_warm(clf, pl.effective);                                       // :577
    lap("warm");
    if (rc != BH_OK) return hfail(rc, bh_last_error());
    bh_batch_context *ctx = nullptr;                                                         // :582-603
    if (pl.effective > 1 || device || bat_mode) {
        rc = bh_batch_context_create(clf, pl.effective, &ctx);
        if (rc != BH_OK) {
            if (device || bat_mode) return hfail(rc, bh_last_error());
            ctx = nullptr;  // fall back to predict_batch like the reference does for Perch
        }
    }
    std::unique_ptr<bh_batch_context, void (*)(bh_batch_context *)> ctx_own(ctx, bh_batch_context_destroy);

    lap("context");
    // run_streaming_inference -- :114-190
    std::vector<Detection> detections;
    RunStats st;
    std::string fail_msg;
    const int fail_code = device ? run_device_front_end(clf, pl, info, ctx, map.pcm, map.fmt, map.bps, map.n_frames, channels, detections, st, fail_msg, map.fd, map.data_offset)
                                 : run_host_front_end(clf, pl, info, ctx, detections, st, fail_msg);
    lap("inference");
    ctx_own.reset();
    if (fail_code) return hfail(fail_code, fail_msg);

    rc = finish_file(cfg, pl, formats, detections, st, has_duration, duration, overlap_secs, t_start, res, lap);
    if (rc != BH_OK) return rc;
    res->front_end = device ? BHH_FRONT_END_DEVICE : BHH_FRONT_END_HOST;
    return BH_OK;
} catch (...) { return h_on_exception(); }


// ---- many short files: packed uploads (no counterpart in the reference; see include/birda_host.h) ---------------------------
namespace {

struct PackedFile {
    size_t index = 0;                 // into paths / results
    FilePlan pl;
    PcmMapping map;
    uint32_t channels = 0;
    int has_duration = 0;
    double duration = 0.0;
    float overlap_secs = 0.f;
    size_t n_segments = 0;
    size_t base_frame = 0;            // where its stream starts in the pack
    std::chrono::steady_clock::time_point t_start;
};

// the part of bhh_process_file's plan that the device front end needs; false = not packable (leave it to bhh_process_file)
bool plan_packable(bh_classifier *clf, const bh_model_info &info, const bhh_processing_config *cfg, const char *path, size_t pack_segments,
                   PackedFile &pf) {
    bh_decoder *probe = nullptr;
    if (bhh_decoder_open(path, &probe) != BH_OK) return false;
    std::unique_ptr<bh_decoder, void (*)(bh_decoder *)> own(probe, bhh_decoder_close);
    FilePlan &pl = pf.pl;
    pl.path = path;
    pl.shown = path;
    pl.source_rate = bhh_decoder_sample_rate(probe);
    pf.has_duration = bhh_decoder_duration_hint(probe, &pf.duration);
    pl.target_rate = info.sample_rate;
    pl.segment_duration = info.segment_duration;
    pl.segment_samples = bhh_duration_to_samples(pl.segment_duration, pl.target_rate);
    pl.overlap_samples = bhh_duration_to_samples(cfg->overlap, pl.target_rate);
    pf.overlap_secs = cfg->overlap;
    pl.estimated = bhh_estimate_segment_count(pf.has_duration, pf.duration, pl.segment_duration, pf.overlap_secs);
    pl.min_confidence = cfg->min_confidence;
    if (pl.segment_samples != info.sample_count) return false;
    pl.resampling = pl.source_rate != pl.target_rate;
    if (pl.resampling && bh_resample_supported(clf, pl.source_rate, pl.target_rate) != BH_OK) return false;      // bhh_process_file reports it
    pl.src_segment_samples = bhh_source_samples(pl.segment_samples, pl.source_rate, pl.target_rate);
    pl.src_overlap_samples = bhh_source_samples(pl.overlap_samples, pl.source_rate, pl.target_rate);
    if (pl.src_overlap_samples >= pl.src_segment_samples) return false;      // bhh_process_file reports it
    if (!pf.map.open(*probe)) return false;
    pf.channels = (uint32_t)probe->channels;
    pf.n_segments = bh_segment_starts(pf.map.n_frames, pl.src_segment_samples, pl.src_overlap_samples, nullptr, 0);
    (void)clf;
    // anything whose stream (plus its trailing silence) fits a context's staging buffer -- pack_segments f32 segments -- can go
    // through the two-deep pipeline below, a long file as a pack of its own: its forward then runs under the previous file's
    // output writing and the next file's copy
    const size_t bytes = (pf.map.n_frames + pl.src_segment_samples) * pf.channels * pf.map.bps;
    return pf.n_segments > 0 && bytes <= pack_segments * info.sample_count * sizeof(float);
}

}  // namespace

extern "C" int bhh_process_files(bh_classifier *clf, const bhh_processing_config *cfg, const char *const *paths, size_t n_files,
                                 size_t pack_segments, bhh_process_result *results, int *status) try {
    if (!clf || !cfg || (n_files && (!paths || !results))) return hfail(BH_ERR_INVALID, "process_files: null argument");
    if (pack_segments == 0) pack_segments = 1024;
    bh_model_info info;
    if (bh_classifier_info(clf, &info) != BH_OK) return hfail(BH_ERR_INVALID, bh_last_error());
    const uint32_t formats = cfg->formats ? cfg->formats : BHH_FORMAT_CSV;
    const bool packing = !cfg->custom_classifier && !cfg->reporter && cfg->front_end != BHH_FRONT_END_HOST && !(formats & ~BHH_FORMAT_ALL);
    for (size_t i = 0; i < n_files; i++) {
        memset(&results[i], 0, sizeof results[i]);
        if (status) status[i] = BH_OK;
    }
    auto single = [&](size_t i) {
        bhh_processing_config c = *cfg;
        c.input_path = paths[i];
        c.display_path = nullptr;
        const int rc = paths[i] ? bhh_process_file(clf, &c, &results[i]) : BH_ERR_INVALID;
        if (status) status[i] = rc;
    };
    // Several packs in flight: while the uploads + forwards of packs k and k - 1 run on their worker threads, this thread scatters
    // and writes pack k - 2 and plans and copies pack k + 1 into a free context's staging buffer.
    struct Pack {
        std::vector<std::unique_ptr<PackedFile>> files;
        std::vector<uint64_t> starts;
        std::vector<bh_result> rows;
        size_t total_frames = 0, total_segs = 0, seg = 0, ch = 0, bps = 2;
        uint32_t rate = 0, fmt = BH_PCM_S16;
        int rc = BH_OK;
        int slot = 0;
        std::thread worker;
        ~Pack() { if (worker.joinable()) worker.join(); }   // (an exception on the way: never leave a running thread behind)
    };
    // DEPTH packs in flight, each in its own batch context: with two, a pack's forward ran alone while this thread assembled the
    // next one and wrote the previous one's outputs (4.6 ms of every 9: the device then works one launch chain at a time); with
    // three there are always two forwards for the device to interleave (6 files of 1 000 segments: 96 k -> see DESIGN section 6)
    constexpr int MAX_DEPTH = 4;
    static const int DEPTH = [] {
        const char *e = getenv("BIRDA_HOST_PIPELINE_DEPTH");
        return e ? std::max(2, std::min(4, atoi(e))) : 3;
    }();
    bh_batch_context *ctx[MAX_DEPTH] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<std::unique_ptr<bh_batch_context, void (*)(bh_batch_context *)>> ctx_own;
    const bool timing = getenv("BIRDA_HOST_TIMING") != nullptr;   // diagnostic: where a call's time goes, to stderr
    double t_plan = 0, t_copy = 0, t_wait = 0, t_finish = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };

    std::unique_ptr<Pack> building(new Pack);
    std::deque<std::unique_ptr<Pack>> running;           // forwards on their worker threads, oldest first
    size_t n_packs = 0;

    // scatter the rows of a finished pack back to its files: detections, sort, outputs, result records
    auto complete = [&](Pack &p) {
        if (p.worker.joinable()) {
            auto t0 = now();
            p.worker.join();
            t_wait += ms(t0, now());
        }
        auto t1 = now();
        // A pack that failed as a whole -- it did not fit the staging buffer, ONE of its recordings drove an f16 operand out of
        // range (BH_ERR_NONFINITE), an upload or launch failed -- says nothing about the other files in it: the reference isolates
        // failures per file (lib.rs:1003-1100 counts the file in files_failed and goes on), so every file of the pack is re-run
        // the reference's way, one at a time, and only the offending one reports the error.
        if (p.rc != BH_OK) {
            for (auto &pfp : p.files) single(pfp->index);
            t_finish += ms(t1, now());
            return;
        }
        size_t row = 0;
        for (auto &pfp : p.files) {
            PackedFile &pf = *pfp;
            bhh_process_result *res = &results[pf.index];
            int frc = p.rc;
            if (p.rc == BH_OK) {
                std::vector<Detection> detections;
                RunStats st;
                for (size_t i = 0; i < pf.n_segments; i++, row++) {
                    float t0s, t1s;
                    chunk_times(pf.pl, (size_t)(p.starts[row] - pf.base_frame), t0s, t1s);
                    collect_detections(clf, pf.pl, p.rows[row], t0s, t1s, detections);
                }
                st.segments = pf.n_segments;
                st.batches = 1;
                res->effective_batch = p.total_segs;
                auto nolap = [](const char *) {};
                frc = finish_file(cfg, pf.pl, formats, detections, st, pf.has_duration, pf.duration, pf.overlap_secs, pf.t_start, res, nolap);
                res->front_end = BHH_FRONT_END_DEVICE;
            }
            if (status) status[pf.index] = frc;
        }
        t_finish += ms(t1, now());
    };
    auto retire = [&](size_t keep) {   // finish the oldest packs until at most `keep` are in flight (outputs stay in file order)
        while (running.size() > keep) { complete(*running.front()); running.pop_front(); }
    };
    auto drain = [&]() { retire(0); };
    // assemble `building` in its slot's staging buffer, hand it to the worker, then finish the pack that ran before it
    auto flush = [&]() {
        if (building->files.empty()) return;
        std::unique_ptr<Pack> p = std::move(building);
        building.reset(new Pack);
        auto tp0 = now();
        retire((size_t)DEPTH - 1);            // the slot this pack takes is free again
        p->slot = (int)(n_packs++ % (size_t)DEPTH);
        const PackedFile &f0 = *p->files.front();
        p->seg = f0.pl.src_segment_samples; p->ch = f0.channels; p->rate = f0.pl.source_rate; p->fmt = f0.map.fmt; p->bps = f0.map.bps;
        for (auto &pf : p->files) { pf->base_frame = p->total_frames; p->total_frames += pf->map.n_frames + p->seg; p->total_segs += pf->n_segments; }
        const size_t bytes = p->total_frames * p->ch * p->bps;
        if (!ctx[p->slot]) {
            int rc = bh_classifier_ensure_warm(clf, std::min<size_t>(pack_segments, 256));
            if (rc == BH_OK) rc = bh_batch_context_create(clf, pack_segments, &ctx[p->slot]);
            if (rc == BH_OK) ctx_own.emplace_back(ctx[p->slot], bh_batch_context_destroy);
            // several packs in flight: another pack's forward covers this one's upload, so a pack runs as few, large sub-slices
            // (measured, 8 files of 1 000 segments, three in flight: automatic split 97 k segments/s, two halves 112 k)
            static const int sub = [] { const char *e = BH_XENV("BIRDA_HOST_PACK_SUBSLICES"); return e ? atoi(e) : 2; }();
            if (rc == BH_OK) (void)bh_batch_context_set_sub_slices(ctx[p->slot], (uint32_t)std::max(0, sub));
            p->rc = rc;
        }
        size_t cap = 0;
        unsigned char *dst = p->rc == BH_OK ? static_cast<unsigned char *>(bh_batch_context_host_buffer(ctx[p->slot], &cap)) : nullptr;
        // (int16 streams with their silences fit the f32 staging buffer of pack_segments segments unless the files have many
        //  channels -- then the pack goes file by file)
        if (p->rc == BH_OK && cap < bytes) p->rc = BH_ERR_UNSUPPORTED;
        if (p->rc == BH_OK) {
            // one copy, page cache -> pinned memory, in 4-MiB pieces on a few threads (the copies fault the mapped pages in)
            struct Piece { unsigned char *d; const unsigned char *s; size_t n; };
            std::vector<Piece> pieces;
            const size_t fb = p->ch * p->bps;      // bytes per frame; zero bytes are silence in all four formats
            for (auto &pfp : p->files) {
                const PackedFile &pf = *pfp;
                unsigned char *d = dst + pf.base_frame * fb;
                const size_t nbytes = pf.map.n_frames * fb;
                for (size_t o = 0; o < nbytes; o += (size_t)4 << 20) pieces.push_back({d + o, pf.map.pcm + o, std::min<size_t>((size_t)4 << 20, nbytes - o)});
                pieces.push_back({d + nbytes, nullptr, p->seg * fb});
            }
            const unsigned nthreads = (unsigned)std::min<size_t>(8, pieces.size());
            std::atomic<size_t> next{0};
            auto work = [&next, &pieces] {
                for (size_t k; (k = next.fetch_add(1)) < pieces.size();) {
                    const Piece &q = pieces[k];
                    if (q.s) memcpy(q.d, q.s, q.n);
                    else memset(q.d, 0, q.n);
                }
            };
            std::vector<std::thread> th;
            for (unsigned t = 1; t < nthreads; t++) th.emplace_back(work);
            work();
            for (auto &t : th) t.join();
            p->starts.reserve(p->total_segs);
            std::vector<uint64_t> one;
            for (auto &pf : p->files) {
                one.resize(pf->n_segments);
                bh_segment_starts(pf->map.n_frames, p->seg, pf->pl.src_overlap_samples, one.data(), one.size());
                for (uint64_t v : one) p->starts.push_back(pf->base_frame + v);
            }
            p->rows.resize(p->total_segs);
        }
        t_copy += ms(tp0, now());
        if (p->rc == BH_OK) {
            Pack *pp = p.get();
            bh_batch_context *c = ctx[p->slot];
            p->worker = std::thread([pp, c, clf, dst] {
                void *guard = bhh_watchdog_start(watchdog_timeout_secs() * 1000, pp->total_segs);
                pp->rc = bh_predict_pcm_at(clf, c, dst, pp->fmt, pp->total_frames, (uint32_t)pp->ch, pp->rate, pp->starts.data(), pp->total_segs, pp->rows.data());
                bhh_watchdog_cancel(guard);
            });
        }
        running.push_back(std::move(p));
        retire((size_t)DEPTH - 1);
    };

    size_t pack_segs = 0;
    for (size_t i = 0; i < n_files; i++) {
        if (!paths[i]) { if (status) status[i] = BH_ERR_INVALID; continue; }
        std::unique_ptr<PackedFile> pf(new PackedFile);
        pf->index = i;
        pf->t_start = std::chrono::steady_clock::now();
        const bool can = packing && plan_packable(clf, info, cfg, paths[i], pack_segments, *pf);
        t_plan += ms(pf->t_start, now());
        if (!can) {
            flush();             // keep the order of outputs and of any side effects
            pack_segs = 0;
            drain();
            single(i);
            continue;
        }
        if (!building->files.empty()) {
            const PackedFile &f0 = *building->files.front();
            if (f0.pl.source_rate != pf->pl.source_rate || f0.channels != pf->channels || f0.map.fmt != pf->map.fmt ||
                pack_segs + pf->n_segments > pack_segments) { flush(); pack_segs = 0; }
        }
        pack_segs += pf->n_segments;
        building->files.push_back(std::move(pf));
    }
    flush();
    drain();
    if (timing)
        fprintf(stderr, "process_files: %zu packs; plan %.2f ms, assemble %.2f, waiting for forwards %.2f, detections + outputs %.2f\n", n_packs,
                t_plan, t_copy, t_wait, t_finish);
    return BH_OK;
} catch (...) { return h_on_exception(); }


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
                                         size_t n_classes, float threshold, float *out_scores, size_t *mapped, size_t *in_range) try {
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
} catch (...) { return h_on_exception(); }


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

extern "C" BH_API size_t bhh_collect_input_files(const char *const *paths, size_t n_paths, char *out, size_t cap, size_t *n_files) try {
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
} catch (...) { return (h_on_exception(), (size_t)-1); }


// ---- range-filter date arithmetic (reference src/utils/date.rs:21-70; constants.rs:324-330,399) ------------------------------
namespace {
const uint32_t kDaysInMonth[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
}
extern "C" uint32_t bhh_date_to_week(uint32_t month, uint32_t day) {
    if (month < 1) month = 1;          // (the reference subtracts 1 from an unsigned month: 0 is a panic there)
    uint32_t doy = day;
    for (uint32_t m = 0; m + 1 < month && m < 12; m++) doy += kDaysInMonth[m];
    if (doy < 1) doy = 1;
    const float q = std::floor((float)(doy - 1) / 7.6f);      // f32 arithmetic, as `(day_of_year - 1) as f32 / DAYS_PER_WEEK`
    const uint32_t week = (uint32_t)q + 1;
    return week < 48 ? week : 48;
}
extern "C" uint32_t bhh_week_to_start_day(uint32_t week) {
    if (week < 1) week = 1;
    return (uint32_t)std::fmaf((float)(week - 1), 7.6f, 1.0f);
}
extern "C" void bhh_day_of_year_to_date(uint32_t day_of_year, uint32_t *month, uint32_t *day) {
    uint32_t remaining = day_of_year, mo = 12, d = 31;
    for (uint32_t m = 0; m < 12; m++) {
        if (remaining <= kDaysInMonth[m]) { mo = m + 1; d = remaining; break; }
        remaining -= kDaysInMonth[m];
    }
    if (month) *month = mo;
    if (day) *day = d;
}

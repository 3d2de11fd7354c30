// AddressSanitizer / UBSan build of the host-side pipeline pieces that parse untrusted input or format output
// ("-fsanitize=address,undefined host build").  Compiled and run by tests/test_host_sanitizers.py on
// the CPU; the device entry points host_pipeline.cpp links against are stubbed out here (nothing below reaches them).
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/birda_hip.h"
#include "../../include/birda_host.h"
#include "../../birda_amd/csrc/model.hpp"
#include "../../birda_amd/csrc/onnx_dense.hpp"
#include "../../birda_amd/csrc/onnx_conv.hpp"
#include "../../birda_amd/csrc/host_internal.hpp"

// ---- stubs for the device library (never called by this driver) ----
extern "C" {
const char *bh_last_error(void) { return "stub"; }
int bh_classifier_info(const bh_classifier *, bh_model_info *) { return BH_ERR_NO_DEVICE; }
const char *bh_classifier_label(const bh_classifier *, uint32_t) { return nullptr; }
int bh_classifier_ensure_warm(bh_classifier *, size_t) { return BH_ERR_NO_DEVICE; }
int bh_batch_context_create(bh_classifier *, size_t, bh_batch_context **) { return BH_ERR_NO_DEVICE; }
void bh_batch_context_destroy(bh_batch_context *) {}
int bh_predict(bh_classifier *, const float *, size_t, bh_result *) { return BH_ERR_NO_DEVICE; }
int bh_predict_batch(bh_classifier *, const float *const *, size_t, size_t, bh_result *) { return BH_ERR_NO_DEVICE; }
int bh_predict_batch_with_context(bh_classifier *, bh_batch_context *, const float *const *, size_t, size_t, bh_result *) { return BH_ERR_NO_DEVICE; }
int bh_predict_batch_source_rate(bh_classifier *, bh_batch_context *, const float *const *, size_t, size_t, uint32_t, bh_result *) { return BH_ERR_NO_DEVICE; }
int bh_predict_pcm16(bh_classifier *, bh_batch_context *, const int16_t *, size_t, uint32_t, uint32_t, size_t, bh_result *, size_t, size_t *, uint64_t *) { return BH_ERR_NO_DEVICE; }
int bh_predict_pcm(bh_classifier *, bh_batch_context *, const void *, uint32_t, size_t, uint32_t, uint32_t, size_t, bh_result *, size_t, size_t *, uint64_t *) { return BH_ERR_NO_DEVICE; }
int bh_predict_pcm_at(bh_classifier *, bh_batch_context *, const void *, uint32_t, size_t, uint32_t, uint32_t, const uint64_t *, size_t, bh_result *) { return BH_ERR_NO_DEVICE; }
size_t bh_segment_starts(size_t, size_t, size_t, uint64_t *, size_t) { return 0; }
int bh_predict_pcm_rows(bh_classifier *, bh_batch_context *, const void *, uint32_t, size_t, uint32_t, uint32_t, size_t, bh_result *, size_t, size_t *, uint64_t *,
                        bh_rows_fn, void *) { return BH_ERR_NO_DEVICE; }
int bh_predict_pcm_fd_rows(bh_classifier *, bh_batch_context *, int, uint64_t, uint32_t, size_t, uint32_t, uint32_t, size_t, bh_result *, size_t, size_t *, uint64_t *,
                           bh_rows_fn, void *) { return BH_ERR_NO_DEVICE; }
int bh_resample_supported(bh_classifier *, uint32_t, uint32_t) { return BH_OK; }
int bh_batch_context_set_sub_slices(bh_batch_context *, uint32_t) { return BH_OK; }
void *bh_batch_context_host_buffer(bh_batch_context *, size_t *) { return nullptr; }
int bh_predict_batch_two_stage(bh_classifier *, bh_batch_context *, bh_custom_classifier *, const float *const *, size_t, size_t, bh_result *, float *) { return BH_ERR_NO_DEVICE; }
const char *bh_custom_classifier_label(const bh_custom_classifier *, uint32_t) { return nullptr; }
size_t bh_classifier_default_batch_size(const bh_classifier *) { return 256; }
}

static int failures = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)

static void put32(std::string &s, uint32_t v) { s.append(reinterpret_cast<const char *>(&v), 4); }
static void put16(std::string &s, uint16_t v) { s.append(reinterpret_cast<const char *>(&v), 2); }
static void write_file(const std::string &path, const std::string &bytes) {
    FILE *f = fopen(path.c_str(), "wb");
    fwrite(bytes.data(), 1, bytes.size(), f);
    fclose(f);
}

int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    // 0. the writers' number formatting (append_fixed: an integer fast path beside printf) against printf itself: random floats,
    //    exact ties of the rounding (x.x5, x.xxxx5 where representable), values around them, signs, zeros, the unusual
    {
        uint64_t rs = 0xd1b54a32d192ed03ull;
        auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; };
        auto same = [&](double v, int places) {
            std::string a;
            bhh::append_fixed(a, v, places);
            char b[400];
            snprintf(b, sizeof b, "%.*f", places, v);
            if (a != b) { fprintf(stderr, "append_fixed(%a, %d) = %s, printf says %s\n", v, places, a.c_str(), b); failures++; }
        };
        for (int it = 0; it < 200000; it++) {
            const uint32_t bits = (uint32_t)rnd();
            float f; memcpy(&f, &bits, 4);
            const int places = it % 3 == 0 ? 1 : it % 3 == 1 ? 4 : (int)(rnd() % 8);
            same((double)f, places);
            same((double)(float)((double)(rnd() % 2000000) / 1000.0), places);          // times like the pipeline's
            same((double)(float)((double)(rnd() % 100001) / 100000.0), 4);             // confidences
            const double tie = ((double)(rnd() % 100000) + 0.5) / (it % 2 ? 10.0 : 10000.0);
            same(tie, it % 2 ? 1 : 4);
            same(std::nextafter(tie, 0.0), it % 2 ? 1 : 4);
            same(std::nextafter(tie, 1e9), it % 2 ? 1 : 4);
            same(-tie, it % 2 ? 1 : 4);
        }
        for (double v : {0.0, -0.0, 0.05, -0.04, 0.25, 0.35, 2.5, 1e9, 1e15, 1e300, -1e-300, 999999999.95, 0.99995, 0.00005})
            for (int places = 0; places <= 7; places++) same(v, places);
        same(std::nan(""), 1); same(INFINITY, 4); same(-INFINITY, 1);
    }
    // 1. WAV headers: well-formed, truncated at every length, hostile chunk sizes
    std::string wav = "RIFF";
    put32(wav, 36 + 2000); wav += "WAVEfmt "; put32(wav, 16); put16(wav, 1); put16(wav, 2); put32(wav, 44100); put32(wav, 176400); put16(wav, 4); put16(wav, 16);
    wav += "LIST"; put32(wav, 5); wav += "abcde"; wav += '\0';
    wav += "data"; put32(wav, 2000);
    for (int i = 0; i < 1000; i++) put16(wav, (uint16_t)(i * 37));
    for (size_t cut = 0; cut <= wav.size(); cut += (cut < 80 ? 1 : 97)) {
        const std::string p = dir + "/cut.wav";
        write_file(p, wav.substr(0, cut));
        bh_decoder *d = nullptr;
        if (bhh_decoder_open(p.c_str(), &d) == BH_OK) {
            std::vector<float> seg(300);
            size_t start = 0;
            int n = 0;
            while (bhh_decoder_next_segment(d, 300, 100, seg.data(), &start) == 1 && n < 100) n++;
            bhh_decoder_close(d);
        }
    }
    for (uint32_t evil : {0u, 1u, 15u, 17u, 0x7fffffffu, 0xfffffff0u, 0xffffffffu}) {
        std::string w = "RIFF"; put32(w, 100); w += "WAVEfmt "; put32(w, evil); put16(w, 1); put16(w, 1); put32(w, 48000); put32(w, 96000); put16(w, 2); put16(w, 16);
        w += "data"; put32(w, evil); w += std::string(64, '\x11');
        const std::string p = dir + "/evil.wav";
        write_file(p, w);
        bh_decoder *d = nullptr;
        if (bhh_decoder_open(p.c_str(), &d) == BH_OK) {
            float seg[64]; size_t st;
            (void)bhh_decoder_next_segment(d, 64, 0, seg, &st);
            bhh_decoder_close(d);
        }
    }
    // 1b. random mutations of well-formed WAV headers (PCM16 / PCM24 / PCM32 / float32, 1-3 channels), deterministic xorshift
    {
        uint64_t rs = 0x9e3779b97f4a7c15ull;
        auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; };
        for (int it = 0; it < 3000; it++) {
            const int fmt_kind = it % 4, ch = 1 + (it / 4) % 3;
            const uint16_t tag = fmt_kind == 3 ? 3 : 1, bits = fmt_kind == 0 ? 16 : fmt_kind == 1 ? 24 : 32;
            const uint32_t nbytes = 600 * ch * (bits / 8);
            std::string w = "RIFF"; put32(w, 36 + nbytes); w += "WAVEfmt "; put32(w, 16); put16(w, tag); put16(w, (uint16_t)ch); put32(w, 22050);
            put32(w, 22050u * ch * (bits / 8)); put16(w, (uint16_t)(ch * (bits / 8))); put16(w, bits); w += "data"; put32(w, nbytes);
            for (uint32_t i = 0; i < nbytes; i++) w += (char)rnd();
            const int nmut = 1 + (int)(rnd() % 4);
            for (int k = 0; k < nmut; k++) w[rnd() % 44] = (char)rnd();          // header bytes only: the parser is what is fuzzed
            if (rnd() % 8 == 0) w.resize(rnd() % w.size());
            const std::string p = dir + "/fuzz.wav";
            write_file(p, w);
            bh_decoder *d = nullptr;
            if (bhh_decoder_open(p.c_str(), &d) == BH_OK) {
                std::vector<float> seg(257);
                size_t start = 0;
                int n = 0;
                while (bhh_decoder_next_segment(d, 257, (size_t)(rnd() % 257), seg.data(), &start) == 1 && n < 20) n++;
                bhh_decoder_close(d);
            }
        }
    }
    // 1c. the model containers (BHM1 / BHC1) are files too: mutated headers and tables must be refused or load consistently
    if (argc > 3) {
        uint64_t rs = 0x2545f4914f6cdd1dull;
        auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; };
        for (int which = 0; which < 2; which++) {
            std::string good;
            {
                FILE *f = fopen(argv[2 + which], "rb");
                CHECK(f != nullptr);
                if (!f) continue;
                char b[65536]; size_t n;
                while ((n = fread(b, 1, sizeof b, f)) > 0) good.append(b, n);
                fclose(f);
            }
            const size_t table = which == 0 ? 256 + 64 * 4 + 128 * 64 : 64 + 32 * 8;     // header + the first records
            int loaded = 0;
            for (int it = 0; it < 1500; it++) {
                std::string w = good;
                const int nmut = 1 + (int)(rnd() % 3);
                for (int k = 0; k < nmut; k++) w[rnd() % std::min(table, w.size())] = (char)rnd();
                if (rnd() % 10 == 0) w.resize(rnd() % w.size());
                const std::string p = dir + "/fuzz.model";
                write_file(p, w);
                std::string err;
                if (which == 0) { bh::Model m; if (bh::load_model(p.c_str(), m, err)) { loaded++; (void)m.macs_per_segment(); } }
                else { bh::CustomModel m; if (bh::load_custom_model(p.c_str(), m, err)) loaded++; }
            }
            printf("model fuzz %d: %d of 1500 mutants still load\n", which, loaded);
        }
    }
    // 1d. the geomodel's .onnx file goes through the library's own protobuf walk (onnx_dense.hpp): every prefix of a
    //     well-formed file and 3 000 random mutants must be refused or load into a consistent dense stack
    if (argc > 4) {
        uint64_t rs = 0xda942042e4dd58b5ull;
        auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; };
        for (int which = 4; which < argc; which++) {
            std::string good;
            FILE *f = fopen(argv[which], "rb");
            CHECK(f != nullptr);
            if (!f) continue;
            char b[65536]; size_t n;
            while ((n = fread(b, 1, sizeof b, f)) > 0) good.append(b, n);
            fclose(f);
            const std::string p = dir + "/fuzz.onnx";
            {
                bh::CustomModel m; std::string err;
                CHECK(bh::onnxd::load_dense_onnx(argv[which], m, err));
                CHECK(m.h.input_dim == 3 && m.layers.size() >= 1 && m.h.n_classes == m.layers.back().out_dim);
            }
            int loaded = 0;
            for (size_t cut = 0; cut < good.size(); cut++) {
                write_file(p, good.substr(0, cut));
                bh::CustomModel m; std::string err;
                if (bh::onnxd::load_dense_onnx(p.c_str(), m, err)) loaded++;
            }
            for (int it = 0; it < 3000; it++) {
                std::string w = good;
                const int nmut = 1 + (int)(rnd() % 4);
                for (int k = 0; k < nmut; k++) w[rnd() % w.size()] = (char)rnd();
                if (rnd() % 10 == 0) w.resize(rnd() % w.size());
                write_file(p, w);
                bh::CustomModel m; std::string err;
                if (bh::onnxd::load_dense_onnx(p.c_str(), m, err)) {
                    loaded++;
                    uint64_t dim = m.h.input_dim;
                    for (const auto &L : m.layers) {
                        CHECK(L.in_dim == dim && L.w_off + (uint64_t)L.in_dim * L.out_dim <= m.blob.size() && L.b_off + L.out_dim <= m.blob.size());
                        dim = L.out_dim;
                    }
                    CHECK(dim == m.h.n_classes);
                }
            }
            printf("onnx fuzz %s: %d prefixes / mutants still load\n", argv[which], loaded);
        }
    }
    // 1e. the classifier's own .onnx file goes through onnx_conv.hpp (protobuf walk, conv-stack walk, BN folding, the family
    //     table): prefixes and random mutants of well-formed conv-stack graphs (BIRDA_FUZZ_CONV_ONNX = paths separated by ':') must
    //     be refused or come out as a model that passes the container's own validation
    if (const char *lst = getenv("BIRDA_FUZZ_CONV_ONNX")) {
        uint64_t rs = 0x9e3779b97f4a7c15ull;
        auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; };
        std::string all(lst);
        size_t a = 0;
        while (a <= all.size()) {
            const size_t b = all.find(':', a) == std::string::npos ? all.size() : all.find(':', a);
            const std::string path = all.substr(a, b - a);
            a = b + 1;
            if (path.empty()) continue;
            std::string good;
            FILE *f = fopen(path.c_str(), "rb");
            CHECK(f != nullptr);
            if (!f) continue;
            char buf[65536]; size_t n;
            while ((n = fread(buf, 1, sizeof buf, f)) > 0) good.append(buf, n);
            fclose(f);
            // (a graph that starts at the audio input has its front-end READ by probing on every load that gets that far --
            //  onnx_frontend.hpp, ~0.1 s for the quarter-second test models, ten times that under the sanitizers: the number of
            //  prefixes and mutants follows the measured cost of one good load, about a minute per file)
            double t_good = 0;
            {
                bh::Model m; std::string err;
                const auto t0 = std::chrono::steady_clock::now();
                CHECK(bh::onnxc::load_onnx_model(path.c_str(), m, err));
                t_good = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                CHECK(m.layers.size() >= 3 && m.h.n_classes == m.layers.back().cout && m.macs_per_segment() > 0);
            }
            const int n_mut = (int)std::min(2500.0, std::max(150.0, 40.0 / std::max(t_good, 1e-3)));
            const std::string p = dir + "/fuzz_conv.onnx";
            int loaded = 0;
            const size_t stride = std::max<size_t>(1, good.size() / (size_t)(n_mut * 3 / 5));
            for (size_t cut = 0; cut < good.size(); cut += stride) {
                write_file(p, good.substr(0, cut));
                bh::Model m; std::string err;
                if (bh::onnxc::load_onnx_model(p.c_str(), m, err)) loaded++;
            }
            // (most of such a file is weight payload: half of the mutants aim at the structural bytes between the tensors -- the
            //  first kilobytes and the tail, where the nodes and value infos of this writer's files sit)
            for (int it = 0; it < n_mut; it++) {
                std::string w = good;
                const int nmut = 1 + (int)(rnd() % 4);
                for (int k = 0; k < nmut; k++) {
                    const size_t span = (it & 1) ? w.size() : std::min<size_t>(w.size(), 6000);
                    const size_t at = (it & 2) ? w.size() - 1 - rnd() % span : rnd() % span;
                    w[at] = (char)rnd();
                }
                if (rnd() % 10 == 0) w.resize(rnd() % w.size());
                write_file(p, w);
                bh::Model m; std::string err;
                if (bh::onnxc::load_onnx_model(p.c_str(), m, err)) {
                    loaded++;
                    std::string e2;
                    CHECK(bh::validate_model(m, e2));
                    (void)m.macs_per_segment();
                }
            }
            printf("conv onnx fuzz %s: %d prefixes / mutants still load\n", path.c_str(), loaded);
        }
    }
    // 2. every writer, odd labels and paths, NaN / inf confidences
    const char *labels[] = {"Passer domesticus_House Sparrow", "NoUnderscore", "_", "a_b_c d_e", "", "\xc3\x84\xc3\xa4kk\xc3\xb6nen laji_\xc3\x96ljy \"quoted\", name\n", "x_\xf0\x9f\x90\xa6 bird"};
    const char *paths[] = {"/a/b/c.wav", "c.wav", "/", "", "a//b///", "../x/..", "/only"};
    for (uint32_t fmt = 1; fmt <= 32; fmt <<= 1) {
        char out[512];
        size_t n = bhh_output_path_for("/in/some.file.flac", dir.c_str(), fmt, out, sizeof out);
        CHECK(n > 0 && n < sizeof out);
        bhh_writer_options o{};
        o.csv_bom = 1; o.csv_columns = "lat,lon,week,model,overlap,sensitivity,min_conf,species_list,unknown"; o.source_file = "s\"f\\.wav"; o.model = "m";
        o.min_confidence = 0.1f; o.overlap = 1.5f; o.audio_duration = 1e9f; o.has_lat = 1; o.lat = -89.999999; o.has_lon = 1; o.lon = 1e-9; o.week = 48;
        bhh_writer *w = nullptr;
        CHECK(bhh_writer_open(fmt, out, &o, &w) == BH_OK);
        CHECK(bhh_writer_write_header(w) == BH_OK);
        int k = 0;
        for (const char *l : labels)
            for (const char *pth : paths) {
                const float conf = k % 11 == 0 ? NAN : k % 13 == 0 ? INFINITY : 0.01f * (float)(k % 100);
                CHECK(bhh_writer_write_detection(w, l, conf, (float)k * 1.5f, (float)k * 1.5f + 3.0f, pth) == BH_OK);
                k++;
            }
        CHECK(bhh_writer_finalize(w) == BH_OK);
    }
    char buf[512];
    for (double v : {0.0, -0.0, 1e-45, 3.4e38, 1e300, 123456789.125, 0.1, (double)NAN, (double)INFINITY})
        for (int kind = 0; kind < 4; kind++) CHECK(bhh_format_float(kind, v, buf, sizeof buf) < sizeof buf);
    char tiny[2] = {'#', '#'};
    CHECK(bhh_format_float(3, 0.1, tiny, 2) == 3 && tiny[0] == '#');            // too small a buffer: length returned, nothing written past it
    CHECK(bhh_species_code("\xc3\x84", buf, sizeof buf) > 0);
    // 3. reporter: every event, hostile strings
    bhh_reporter *r = nullptr;
    CHECK(bhh_reporter_open(BHH_REPORT_JSON, (dir + "/events.json").c_str(), &r) == BH_OK);
    bhh_range_filter_info rf{"3.0.2", 1, 2, 3, 4, "keep", 0.03f};
    bhh_reporter_pipeline_started(r, 3, "m\"\\\n\t\x01", 0.1f, "gpu", "HIP", nullptr, &rf);
    bhh_reporter_file_started(r, paths[0], 0, 10, 1, 30.0);
    for (int i = 0; i <= 100; i += 7) (void)bhh_reporter_file_progress(r, paths[0], (size_t)i, 100, (float)i);
    (void)bhh_reporter_file_progress(r, paths[0], 1, 0, NAN);
    bhh_reporter_batch_progress(r, 1, 3, 33.3f);
    const float c[2] = {0.5f, NAN}, st[2] = {0.f, 3.f}, en[2] = {3.f, 6.f};
    bhh_reporter_detections(r, paths[1], labels, c, st, en, 2);
    bhh_reporter_file_completed(r, paths[0], BHH_FILE_PROCESSED, 2, 12, nullptr, nullptr);
    bhh_reporter_file_completed(r, paths[1], BHH_FILE_FAILED, 0, 0, "code", "msg\n");
    bhh_reporter_file_completed(r, paths[2], 7, 0, 0, nullptr, nullptr);   // unknown status: ignored
    bhh_reporter_error(r, "e", 0, "m", nullptr);
    bhh_reporter_pipeline_completed(r, 1, 1, 0, 2, 10, 99, 123.456);
    bhh_reporter_close(r);
    // 4. small host helpers at their edges
    CHECK(bhh_estimate_segment_count(1, 10.0, 3.0f, 3.0f) == -1);
    CHECK(bhh_effective_batch_size(8, 0) == 8 && bhh_effective_batch_size(8, 3) == 3);
    CHECK(bhh_source_samples(144000, 44100, 48000) == 132300);
    CHECK(bhh_is_audio_file(".wav") == 0 && bhh_is_audio_file("a.WAV") == 1);
    CHECK(bhh_date_to_week(0, 0) == 1 && bhh_date_to_week(13, 40) == 48 && bhh_date_to_week(6, 15) == 22 && bhh_week_to_start_day(0) == 1);
    { uint32_t mo = 0, dd = 0; bhh_day_of_year_to_date(0, &mo, &dd); CHECK(mo == 1 && dd == 0); bhh_day_of_year_to_date(0xffffffffu, &mo, nullptr); CHECK(mo == 12); }
    void *g = bhh_watchdog_start(60000, 8);
    bhh_watchdog_cancel(g);
    bhh_watchdog_cancel(nullptr);
    // 5. process_file on a file that does not decode: an error code, no crash
    bhh_processing_config cfg{};
    cfg.input_path = (dir + "/cut.wav").c_str();
    bhh_process_result res;
    CHECK(bhh_process_file(reinterpret_cast<bh_classifier *>(&cfg), &cfg, &res) != BH_OK);
    // 6. process_files: bad arguments are refused, undecodable files get their own status, nothing crashes without a device
    {
        const char *many[3] = {cfg.input_path, nullptr, "/nonexistent/x.wav"};
        bhh_process_result rs[3];
        int stt[3] = {0, 0, 0};
        CHECK(bhh_process_files(nullptr, &cfg, many, 3, 0, rs, stt) != BH_OK);
        CHECK(bhh_process_files(reinterpret_cast<bh_classifier *>(&cfg), &cfg, many, 3, 0, rs, stt) != BH_OK);   // (stub classifier: no model info)
    }
    if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
    printf("host sanitizer driver: ok\n");
    return 0;
}

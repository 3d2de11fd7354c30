// Output side of the per-file pipeline: the six result writers behind write_output
// (reference src/pipeline/processor.rs:819-873) and the NDJSON / JSON progress reporter
// (reference src/output/reporter.rs:22-420, json_envelope.rs).
//
//   CsvWriter          src/output/csv.rs:17-132       <stem>.BirdNET.results.csv
//   RavenWriter        src/output/raven.rs:11-86      <stem>.BirdNET.selection.table.txt
//   AudacityWriter     src/output/audacity.rs:11-47   <stem>.BirdNET.results.txt
//   KaleidoscopeWriter src/output/kaleidoscope.rs     <stem>.BirdNET.results.kaleidoscope.csv
//   JsonResultWriter   src/output/json.rs:75-215      <stem>.BirdNET.json
//   ParquetWriter      src/output/parquet.rs          <stem>.BirdNET.results.parquet (host_parquet.cpp)
//
// Text formats are byte-for-byte what the Rust formatters produce: `{:.1}` / `{:.4}` are exact decimal
// expansions rounded half-to-even (glibc printf on the widened value does the same), `{}` on a float is the shortest
// digit string that round-trips, without an exponent (core::fmt::float), and serde_json prints floats through ryu's
// "pretty" layout.  Host logic only: nothing here touches a logit.
#include <sys/stat.h>
#include <algorithm>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include <unistd.h>

#include "host_internal.hpp"

namespace bhh {

// ---- shortest round-trip digits ----------------------------------------------------------------------
// digits d1 d2 ... dn and decimal exponent e such that v = 0.d1d2...dn x 10^e  (n >= 1, d1 != 0 unless v == 0)
template <class F> static void shortest_digits(F v, std::string &digits, int &point) {
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);   // shortest round-trip
    std::string s(buf, r.ptr);                                                          // d[.ddd]e[+-]XX
    const size_t e = s.find('e');
    const int ex = atoi(s.c_str() + e + 1);
    digits.clear();
    for (size_t i = 0; i < e; i++)
        if (s[i] != '.') digits.push_back(s[i]);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    point = ex + 1;   // position of the decimal point relative to the first digit
}

// core::fmt Display for f32 / f64: shortest digits, never an exponent, no trailing ".0"
template <class F> static std::string display_float(F v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    if (v == 0) return std::signbit(v) ? "-0" : "0";
    std::string d; int pt;
    shortest_digits(std::fabs(v), d, pt);
    std::string out = v < 0 ? "-" : "";
    const int n = (int)d.size();
    if (pt <= 0) { out += "0."; out.append((size_t)-pt, '0'); out += d; }
    else if (pt >= n) { out += d; out.append((size_t)(pt - n), '0'); }
    else { out += d.substr(0, (size_t)pt); out += '.'; out += d.substr((size_t)pt); }
    return out;
}
std::string rust_display_f32(float v) { return display_float(v); }
std::string rust_display_f64(double v) { return display_float(v); }

// serde_json float: ryu::Buffer::format_finite ("pretty" layout, ryu/src/pretty/mod.rs); non-finite -> null.
// With kk = position of the decimal point: integers up to kk <= LIM print as "ddd000.0", 0 < kk <= LIM as "dd.ddd",
// LOW < kk <= 0 as "0.00ddd", everything else in exponent form "d.ddde-7" / "de16".  (LIM, LOW) = (13, -6) for f32,
// (16, -5) for f64.
template <class F> static std::string json_float(F v, int lim, int low) {
    if (!std::isfinite(v)) return "null";
    if (v == 0) return std::signbit(v) ? "-0.0" : "0.0";
    std::string d; int kk;
    shortest_digits(std::fabs(v), d, kk);
    const int n = (int)d.size(), k = kk - n;   // v = digits x 10^k
    std::string out = v < 0 ? "-" : "";
    if (0 <= k && kk <= lim) { out += d; out.append((size_t)k, '0'); out += ".0"; }
    else if (0 < kk && kk <= lim) { out += d.substr(0, (size_t)kk); out += '.'; out += d.substr((size_t)kk); }
    else if (low < kk && kk <= 0) { out += "0."; out.append((size_t)-kk, '0'); out += d; }
    else {
        out += d[0];
        if (n > 1) { out += '.'; out += d.substr(1); }
        out += 'e';
        out += std::to_string(kk - 1);
    }
    return out;
}
std::string json_f32(float v) { return json_float(v, 13, -6); }
std::string json_f64(double v) { return json_float(v, 16, -5); }

// serde_json string escaping (ser.rs ESCAPE table): " \ and the C0 controls; everything else verbatim (UTF-8 passes through)
std::string json_string(const std::string &s) {
    std::string o = "\"";
    for (unsigned char c : s) {
        switch (c) {
        case '"': o += "\\\""; break;
        case '\\': o += "\\\\"; break;
        case '\b': o += "\\b"; break;
        case '\f': o += "\\f"; break;
        case '\n': o += "\\n"; break;
        case '\r': o += "\\r"; break;
        case '\t': o += "\\t"; break;
        default:
            if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
            else o += (char)c;
        }
    }
    return o + "\"";
}

// chrono DateTime<Utc> through serde: RFC 3339 with SecondsFormat::AutoSi and a 'Z' suffix
std::string rfc3339_now() {
    using namespace std::chrono;
    const auto now = system_clock::now();
    const auto ns = duration_cast<nanoseconds>(now.time_since_epoch()).count();
    const time_t secs = (time_t)(ns / 1000000000LL);
    const long frac = (long)(ns % 1000000000LL);
    struct tm tmv;
    gmtime_r(&secs, &tmv);
    char b[64];
    size_t n = strftime(b, sizeof b, "%Y-%m-%dT%H:%M:%S", &tmv);
    std::string out(b, n);
    if (frac != 0) {
        char f[16];
        if (frac % 1000000 == 0) snprintf(f, sizeof f, ".%03ld", frac / 1000000);
        else if (frac % 1000 == 0) snprintf(f, sizeof f, ".%06ld", frac / 1000);
        else snprintf(f, sizeof f, ".%09ld", frac);
        out += f;
    }
    return out + "Z";
}

// escape_csv -- csv.rs:126-132
std::string escape_csv(const std::string &v) {
    if (v.find(',') != std::string::npos || v.find('"') != std::string::npos || v.find('\n') != std::string::npos) {
        std::string o = "\"";
        for (char c : v) { if (c == '"') o += '"'; o += c; }
        return o + "\"";
    }
    return v;
}

// Detection::from_label -- types.rs:58-79
Detection detection_from_label(const std::string &label, float conf, float start, float end, const std::string &file_path) {
    const size_t us = label.find('_');
    if (us == std::string::npos) return Detection{start, end, conf, label, label, file_path};
    return Detection{start, end, conf, label.substr(0, us), label.substr(us + 1), file_path};
}

// format!("{:.N}") of a value: printf's %.Nf (both round the exact binary value half to even).  Fast path: scale, round to an
// integer and print its digits -- valid away from a rounding tie, where the double product and the exact one round alike (the
// product's error is below 1e-7 for scaled values under 1e9); at or near a tie, and for anything unusual, printf decides on the
// exact value.  A 1 000-segment file formats 15 000 numbers; through snprintf that was 2 ms of a 15 ms file.
void append_fixed(std::string &o, double v, int places) {
    static const double P10[] = {1.0, 10.0, 100.0, 1e3, 1e4, 1e5, 1e6};
    if (places >= 0 && places <= 6 && std::isfinite(v)) {
        const double scaled = std::fabs(v) * P10[places];
        if (scaled < 1e9) {
            const double r = std::nearbyint(scaled);
            if (std::fabs(std::fabs(scaled - r) - 0.5) > 1e-6) {
                uint64_t q = (uint64_t)r;
                char b[32];
                int n = 0;
                for (int i = 0; i < places; i++) { b[n++] = (char)('0' + q % 10); q /= 10; }
                if (places) b[n++] = '.';
                do { b[n++] = (char)('0' + q % 10); q /= 10; } while (q);
                if (std::signbit(v)) b[n++] = '-';
                while (n) o.push_back(b[--n]);
                return;
            }
        }
    }
    char b[400];   // (DBL_MAX has 309 integer digits)
    const int n = snprintf(b, sizeof b, "%.*f", places, v);
    o.append(b, (size_t)std::min<int>(std::max(n, 0), (int)sizeof b - 1));
}
static std::string fixed(double v, int places) {
    std::string s;
    append_fixed(s, v, places);
    return s;
}
// escape_csv appended in place (the common case, nothing to quote, copies the field once)
static void append_csv(std::string &o, const std::string &v) {
    if (v.find_first_of(",\"\n") == std::string::npos) { o += v; return; }
    o += '"';
    for (char c : v) { if (c == '"') o += '"'; o += c; }
    o += '"';
}

static std::string replace_all(std::string s, const std::string &from, const std::string &to) {
    for (size_t p = 0; (p = s.find(from, p)) != std::string::npos; p += to.size()) s.replace(p, from.size(), to);
    return s;
}

// generate_species_code -- raven.rs:72-84 (chars() are Unicode scalar values: count UTF-8 lead bytes, not bytes)
static std::string take_chars(const std::string &w, size_t n) {
    size_t i = 0, taken = 0;
    while (i < w.size() && taken < n) {
        i++;
        while (i < w.size() && ((unsigned char)w[i] & 0xC0) == 0x80) i++;
        taken++;
    }
    return w.substr(0, i);
}
// str::to_lowercase for the scripts BirdNET's label files use in Latin, Greek and Cyrillic text: ASCII, Latin-1
// Supplement, Latin Extended-A, Greek and Cyrillic capitals (simple one-to-one mappings; no other script in those files
// has case).  Not covered: context rules (final sigma) and the rarer blocks -- such characters pass through unchanged.
static uint32_t lower_cp(uint32_t c) {
    if (c >= 'A' && c <= 'Z') return c + 32;
    if ((c >= 0xC0 && c <= 0xDE && c != 0xD7)) return c + 32;
    if (c >= 0x100 && c <= 0x137) return c | 1;                 // pairs (even capital, odd small); U+0130 handled below
    if (c >= 0x139 && c <= 0x148) return (c & 1) ? c + 1 : c;   // pairs (odd capital, even small)
    if (c >= 0x14A && c <= 0x177) return c | 1;
    if (c == 0x178) return 0xFF;
    if (c >= 0x179 && c <= 0x17E) return (c & 1) ? c + 1 : c;
    if (c >= 0x391 && c <= 0x3A9 && c != 0x3A2) return c + 32;
    if (c >= 0x410 && c <= 0x42F) return c + 32;
    if (c >= 0x400 && c <= 0x40F) return c + 80;
    return c;
}
static std::string ascii_lower(std::string s) {   // (name kept: the ASCII case is the common one)
    std::string o;
    for (size_t i = 0; i < s.size();) {
        const unsigned char b = (unsigned char)s[i];
        if (b < 0x80) { o.push_back((char)lower_cp(b)); i++; continue; }
        if ((b & 0xE0) == 0xC0 && i + 1 < s.size()) {
            uint32_t c = lower_cp(((uint32_t)(b & 0x1F) << 6) | ((unsigned char)s[i + 1] & 0x3F));
            if (c == 0x130) c = 0x130;   // I-with-dot lowercases to two code points in Rust; left as is
            if (c < 0x80) o.push_back((char)c);
            else { o.push_back((char)(0xC0 | (c >> 6))); o.push_back((char)(0x80 | (c & 0x3F))); }
            i += 2;
            continue;
        }
        size_t len = (b & 0xF0) == 0xE0 ? 3 : (b & 0xF8) == 0xF0 ? 4 : 1;
        o.append(s, i, std::min(len, s.size() - i));
        i += len;
    }
    return o;
}
static bool is_rust_whitespace(unsigned char c) { return c == ' ' || (c >= 0x09 && c <= 0x0D); }   // ASCII White_Space
std::string species_code(const std::string &common_name) {
    std::vector<std::string> words;
    std::string cur;
    for (char c : common_name) {
        if (is_rust_whitespace((unsigned char)c)) { if (!cur.empty()) { words.push_back(cur); cur.clear(); } }
        else cur.push_back(c);
    }
    if (!cur.empty()) words.push_back(cur);
    if (words.empty()) return "unkn";
    if (words.size() == 1) return ascii_lower(take_chars(words[0], 4));
    return ascii_lower(take_chars(words.front(), 3)) + ascii_lower(take_chars(words.back(), 3));
}

// ---- std::path::Path pieces the writers use ---------------------------------------------------------
static std::string strip_trailing_slashes(std::string p) {
    while (p.size() > 1 && p.back() == '/') p.pop_back();
    return p;
}
// Path::parent: None for "" and "/"; Some("") for a bare file name
bool path_parent(const std::string &path, std::string &parent) {
    std::string p = strip_trailing_slashes(path);
    if (p.empty() || p == "/") return false;
    const size_t s = p.find_last_of('/');
    if (s == std::string::npos) { parent.clear(); return true; }
    parent = s == 0 ? "/" : strip_trailing_slashes(p.substr(0, s));
    return true;
}
// Path::file_name: None for "", "/" and a path ending in ".."
bool path_file_name(const std::string &path, std::string &name) {
    std::string p = strip_trailing_slashes(path);
    if (p.empty() || p == "/") return false;
    const size_t s = p.find_last_of('/');
    name = s == std::string::npos ? p : p.substr(s + 1);
    return name != "..";
}
// Path::file_stem: the file name up to its LAST dot (a leading dot alone does not start an extension)
static std::string path_file_stem(const std::string &path) {
    std::string name;
    if (!path_file_name(path, name)) return "output";   // output_path_for's fallback, coordinator.rs:66-69
    const size_t dot = name.find_last_of('.');
    if (dot == std::string::npos || dot == 0) return name;
    return name.substr(0, dot);
}

static const char *format_extension(uint32_t format) {   // constants.rs:263-276
    switch (format) {
    case BHH_FORMAT_CSV: return ".BirdNET.results.csv";
    case BHH_FORMAT_RAVEN: return ".BirdNET.selection.table.txt";
    case BHH_FORMAT_AUDACITY: return ".BirdNET.results.txt";
    case BHH_FORMAT_KALEIDOSCOPE: return ".BirdNET.results.kaleidoscope.csv";
    case BHH_FORMAT_JSON: return ".BirdNET.json";
    case BHH_FORMAT_PARQUET: return ".BirdNET.results.parquet";
    default: return nullptr;
    }
}

// output_dir_for + output_path_for -- coordinator.rs:41-94
std::string output_path_for(const std::string &input, const std::string &out_dir, uint32_t format) {
    const char *ext = format_extension(format);
    if (!ext) return "";
    std::string stem = path_file_stem(input);
    for (char &c : stem) if (c == '/' || c == '\\') c = '_';   // sanitize_filename, :55-57
    std::string dir = out_dir;
    if (dir.empty()) {                                          // output_dir_for, :41-50
        std::string parent;
        dir = (path_parent(input, parent) && !parent.empty()) ? parent : (path_parent(input, parent) ? "" : ".");
    }
    // PathBuf::join: an empty directory joins to the bare file name
    if (dir.empty()) return stem + ext;
    return (dir.back() == '/' ? dir : dir + "/") + stem + ext;
}

// ---- writers ---------------------------------------------------------------------------------------
struct Writer {
    uint32_t format = 0;
    std::string path;
    FILE *f = nullptr;
    WriterOptions opt;
    uint32_t selection_id = 0;              // Raven
    std::vector<Detection> kept;            // JSON / Parquet collect, then write at finalize
    bool failed = false;
    size_t extra_columns = 0;               // CSV: additional (empty) cells per row
    std::string row;                        // CSV: the row being built (reused)
    ~Writer() { if (f) { (void)flush_pending(); fclose(f); } }
    std::string pending;                    // rows not yet handed to stdio (one fwrite per 256 KB, not one per row)
    bool flush_pending() {
        if (pending.empty()) return !failed;
        if (!f || fwrite(pending.data(), 1, pending.size(), f) != pending.size()) failed = true;
        pending.clear();
        return !failed;
    }
    bool put(const std::string &s) {
        if (!f) { failed = true; return false; }
        pending += s;
        return pending.size() < (256u << 10) ? !failed : flush_pending();
    }
};

static std::vector<std::string> split_columns(const std::string &cols) {
    std::vector<std::string> out;
    std::string cur;
    for (char c : cols) {
        if (c == ',') { if (!cur.empty()) out.push_back(cur); cur.clear(); }
        else if (c != ' ') cur.push_back(c);
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}

int writer_open(uint32_t format, const std::string &path, const WriterOptions &opt, std::unique_ptr<Writer> &out, std::string &err) {
    if (!format_extension(format)) { err = "InvalidOutputFormat: unknown format bit"; return BH_ERR_INVALID; }
    auto w = std::make_unique<Writer>();
    w->format = format; w->path = path; w->opt = opt;
    w->extra_columns = split_columns(opt.csv_columns).size();
    if (format != BHH_FORMAT_JSON && format != BHH_FORMAT_PARQUET) {   // those two create their file at finalize
        w->f = fopen(path.c_str(), "wb");
        if (!w->f) { err = "cannot create " + path; return BH_ERR_IO; }
        if (format == BHH_FORMAT_CSV && opt.csv_bom && !w->put("\xEF\xBB\xBF")) { err = "write failed: " + path; return BH_ERR_IO; }   // csv.rs:29-32
    }
    out = std::move(w);
    return BH_OK;
}

int writer_header(Writer &w) {
    switch (w.format) {
    case BHH_FORMAT_CSV: {   // csv.rs:41-52
        std::string h = "Start (s),End (s),Scientific name,Common name,Confidence,File";
        for (const auto &c : split_columns(w.opt.csv_columns)) { h += ','; h += c; }
        return w.put(h + "\n") ? BH_OK : BH_ERR_IO;
    }
    case BHH_FORMAT_RAVEN:   // raven.rs:29-35
        return w.put("Selection\tView\tChannel\tBegin Time (s)\tEnd Time (s)\tLow Freq (Hz)\tHigh Freq (Hz)\tCommon Name\t"
                     "Species Code\tConfidence\tBegin Path\tFile Offset (s)\n") ? BH_OK : BH_ERR_IO;
    case BHH_FORMAT_KALEIDOSCOPE:   // kaleidoscope.rs:27-33
        return w.put("INDIR,FOLDER,IN FILE,OFFSET,DURATION,TOP1MATCH,TOP1DIST\n") ? BH_OK : BH_ERR_IO;
    default: return BH_OK;   // Audacity, JSON, Parquet: OutputWriter::write_header's default (writer.rs:9-11)
    }
}

int writer_detection(Writer &w, const Detection &d) {
    switch (w.format) {
    case BHH_FORMAT_CSV: {   // csv.rs:54-118; DetectionMetadata is all-None on this path (types.rs:69-78), so the extra
        // columns are empty cells
        std::string &row = w.row;
        row.clear();
        append_fixed(row, d.start_time, 1); row += ',';
        append_fixed(row, d.end_time, 1); row += ',';
        append_csv(row, d.scientific_name); row += ',';
        append_csv(row, d.common_name); row += ',';
        append_fixed(row, d.confidence, 4); row += ',';
        append_csv(row, d.file_path);
        row.append(w.extra_columns, ',');
        row += '\n';
        return w.put(row) ? BH_OK : BH_ERR_IO;
    }
    case BHH_FORMAT_RAVEN: {   // raven.rs:38-63; VIEW "Spectrogram 1", CHANNEL 1, 150 / 15000 Hz (constants.rs:300-309)
        w.selection_id++;
        std::string row = std::to_string(w.selection_id) + "\tSpectrogram 1\t1\t" + fixed(d.start_time, 1) + "\t" + fixed(d.end_time, 1) +
                          "\t150\t15000\t" + replace_all(d.common_name, " ", "_") + "\t" + species_code(d.common_name) + "\t" +
                          fixed(d.confidence, 4) + "\t" + d.file_path + "\t" + fixed(d.start_time, 1) + "\n";
        return w.put(row) ? BH_OK : BH_ERR_IO;
    }
    case BHH_FORMAT_AUDACITY:   // audacity.rs:27-40
        return w.put(fixed(d.start_time, 1) + "\t" + fixed(d.end_time, 1) + "\t" + replace_all(d.common_name, "_", ", ") + "\t" +
                     fixed(d.confidence, 4) + "\n") ? BH_OK : BH_ERR_IO;
    case BHH_FORMAT_KALEIDOSCOPE: {   // kaleidoscope.rs:36-71
        std::string parent, folder, indir, grand, filename;
        const bool has_parent = path_parent(d.file_path, parent);
        if (has_parent) (void)path_file_name(parent, folder);
        if (has_parent && path_parent(parent, grand)) indir = grand;
        (void)path_file_name(d.file_path, filename);
        const float duration = d.end_time - d.start_time;   // f32 subtraction, as in the reference
        return w.put(indir + "," + folder + "," + filename + "," + fixed(d.start_time, 1) + "," + fixed(duration, 1) + "," +
                     replace_all(d.common_name, " ", "_") + "," + fixed(d.confidence, 4) + "\n") ? BH_OK : BH_ERR_IO;
    }
    case BHH_FORMAT_JSON:
    case BHH_FORMAT_PARQUET:
        w.kept.push_back(d);
        return BH_OK;
    default: return BH_ERR_INVALID;
    }
}

// serde_json::to_writer_pretty of JsonResultFile (json.rs:13-72, 159-215): two-space indent, "key": value
static std::string json_result_file(const Writer &w) {
    const WriterOptions &o = w.opt;
    std::set<std::string> species;
    for (const auto &d : w.kept) species.insert(d.scientific_name);
    std::string s = "{\n";
    s += "  \"source_file\": " + json_string(o.source_file) + ",\n";
    s += "  \"analysis_date\": " + json_string(rfc3339_now()) + ",\n";
    s += "  \"model\": " + json_string(o.model) + ",\n";
    s += "  \"settings\": {\n";
    s += "    \"min_confidence\": " + json_f32(o.min_confidence) + ",\n";
    s += "    \"overlap\": " + json_f32(o.overlap);
    if (o.has_lat) s += ",\n    \"lat\": " + json_f64(o.lat);            // skip_serializing_if = "Option::is_none"
    if (o.has_lon) s += ",\n    \"lon\": " + json_f64(o.lon);
    if (o.week >= 0) s += ",\n    \"week\": " + std::to_string(o.week);
    s += "\n  },\n";
    if (w.kept.empty()) s += "  \"detections\": [],\n";
    else {
        s += "  \"detections\": [\n";
        for (size_t i = 0; i < w.kept.size(); i++) {
            const auto &d = w.kept[i];
            s += "    {\n";
            s += "      \"start_time\": " + json_f32(d.start_time) + ",\n";
            s += "      \"end_time\": " + json_f32(d.end_time) + ",\n";
            s += "      \"scientific_name\": " + json_string(d.scientific_name) + ",\n";
            s += "      \"common_name\": " + json_string(d.common_name) + ",\n";
            s += "      \"confidence\": " + json_f32(d.confidence) + "\n";
            s += i + 1 < w.kept.size() ? "    },\n" : "    }\n";
        }
        s += "  ],\n";
    }
    s += "  \"summary\": {\n";
    s += "    \"total_detections\": " + std::to_string(w.kept.size()) + ",\n";
    s += "    \"unique_species\": " + std::to_string(species.size()) + ",\n";
    s += "    \"audio_duration_seconds\": " + json_f32(o.audio_duration) + "\n";
    s += "  }\n}";
    return s;
}

int writer_finalize(Writer &w, std::string &err) {
    int rc = BH_OK;
    if (w.format == BHH_FORMAT_JSON) {
        w.f = fopen(w.path.c_str(), "wb");
        if (!w.f) { err = "cannot create " + w.path; return BH_ERR_IO; }
        if (!w.put(json_result_file(w))) rc = BH_ERR_IO;
    } else if (w.format == BHH_FORMAT_PARQUET) {
        rc = write_parquet_file(w.path, w.kept, split_columns(w.opt.csv_columns), err);
        if (rc != BH_OK) return rc;
    }
    if (w.f) {
        (void)w.flush_pending();
        if (fflush(w.f) != 0 || w.failed) rc = BH_ERR_IO;   // every sibling writer flushes explicitly (json.rs:196-209)
        if (fclose(w.f) != 0) rc = BH_ERR_IO;
        w.f = nullptr;
    }
    if (rc != BH_OK) err = "write failed: " + w.path;
    return rc;
}

// write_output -- processor.rs:819-873
int write_output(const std::string &input_path, const std::string &out_dir, uint32_t format, const std::vector<Detection> &detections,
                 const WriterOptions &opt, std::string &out_path, std::string &err) {
    out_path = output_path_for(input_path, out_dir, format);
    WriterOptions o = opt;
    if (format == BHH_FORMAT_JSON) {   // source_file = input file name, "unknown" without one (:838-841)
        std::string name;
        o.source_file = path_file_name(input_path, name) ? name : "unknown";
    }
    std::unique_ptr<Writer> w;
    int rc = writer_open(format, out_path, o, w, err);
    if (rc != BH_OK) return rc;
    if ((rc = writer_header(*w)) != BH_OK) { err = "write failed: " + out_path; return rc; }
    for (const auto &d : detections)
        if ((rc = writer_detection(*w, d)) != BH_OK) { err = "write failed: " + out_path; return rc; }
    return writer_finalize(*w, err);
}

// ---- progress reporter (reporter.rs:22-420) -----------------------------------------------------------
struct Reporter {
    int mode = BHH_REPORT_NDJSON;
    FILE *out = nullptr;
    bool owns = false;
    std::mutex mu;
    std::vector<std::string> buffer;        // JSON mode: events held until pipeline_completed / cancelled
    // ProgressThrottler (reporter.rs:93-168): 10 % or 500 ms
    int last_percent = 0;
    std::chrono::steady_clock::time_point last_update = std::chrono::steady_clock::now();
    bool write_error_logged = false;

    bool should_emit(float percent) {
        float cl = std::floor(percent);
        cl = cl < 0.f ? 0.f : (cl > 100.f ? 100.f : cl);
        if (std::isnan(cl)) cl = 0.f;       // `NaN as u8` is 0
        const int cur = (int)cl;
        const auto now = std::chrono::steady_clock::now();
        if (cur == 0 || cur >= 100) { last_percent = cur; last_update = now; return true; }
        const bool pct = (cur > last_percent ? cur - last_percent : 0) >= 10;     // saturating_sub
        const bool time = std::chrono::duration_cast<std::chrono::milliseconds>(now - last_update).count() >= 500;
        if (pct || time) { last_percent = cur; last_update = now; return true; }
        return false;
    }
    void reset() { last_percent = 0; last_update = std::chrono::steady_clock::now(); }

    void emit(const char *event, const std::string &payload) {   // JsonEnvelope (json_envelope.rs:13-37), compact form
        const std::string json = std::string("{\"spec_version\":\"1.1\",\"timestamp\":") + json_string(rfc3339_now()) + ",\"event\":\"" + event +
                                 "\",\"payload\":" + payload + "}";
        if (mode == BHH_REPORT_NDJSON) {
            if (fputs(json.c_str(), out) < 0 || fputc('\n', out) == EOF) {
                if (!write_error_logged) {
                    fprintf(stderr, "birda: warning: failed to write to stdout (subsequent errors suppressed)\n");
                    write_error_logged = true;
                }
            }
            fflush(out);
        } else buffer.push_back(json);
    }
    void flush_json() {   // reporter.rs:224-243
        if (mode != BHH_REPORT_JSON) return;
        fputs("[\n", out);
        for (size_t i = 0; i < buffer.size(); i++) {
            if (i) fputs(",\n", out);
            fputs("  ", out);
            fputs(buffer[i].c_str(), out);
        }
        fputs("\n]\n", out);
        fflush(out);
    }
};

}  // namespace bhh

using namespace bhh;

struct bhh_writer { std::unique_ptr<Writer> w; };
struct bhh_reporter { Reporter r; };

namespace {
thread_local std::string o_err;
int ofail(int code, const std::string &m) { o_err = m; bhh::set_host_error(m); return code; }
int o_on_exception() noexcept {
    try {
        try { throw; }
        catch (const std::bad_alloc &) { return ofail(BH_ERR_INTERNAL, "out of host memory"); }
        catch (const std::exception &e) { return ofail(BH_ERR_INTERNAL, std::string("internal error: ") + e.what()); }
        catch (...) { return ofail(BH_ERR_INTERNAL, "internal error (unknown exception)"); }
    } catch (...) { return BH_ERR_INTERNAL; }
}
size_t copy_out(const std::string &s, char *out, size_t cap) {
    if (out && cap > s.size()) memcpy(out, s.c_str(), s.size() + 1);
    return s.size();
}
WriterOptions options_from(const bhh_writer_options *o) {
    WriterOptions w;
    if (!o) return w;
    w.csv_bom = o->csv_bom != 0;
    if (o->csv_columns) w.csv_columns = o->csv_columns;
    if (o->source_file) w.source_file = o->source_file;
    if (o->model) w.model = o->model;
    w.min_confidence = o->min_confidence; w.overlap = o->overlap; w.audio_duration = o->audio_duration;
    w.has_lat = o->has_lat != 0; w.has_lon = o->has_lon != 0; w.lat = o->lat; w.lon = o->lon; w.week = o->week;
    return w;
}
}  // namespace

extern "C" {

size_t bhh_output_path_for(const char *input_path, const char *output_dir, uint32_t format, char *out, size_t cap) try {
    if (!input_path) return 0;
    return copy_out(output_path_for(input_path, output_dir ? output_dir : "", format), out, cap);
} catch (...) { return (o_on_exception(), (size_t)0); }

int bhh_should_process(const char *input_path, const char *output_dir, uint32_t format_mask, int force) try {
    if (!input_path || force || (format_mask & BHH_FORMAT_ALL) == 0) return 1;
    for (uint32_t f = 1; f <= BHH_FORMAT_PARQUET; f <<= 1) {
        if (!(format_mask & f)) continue;
        const std::string p = output_path_for(input_path, output_dir ? output_dir : "", f);
        struct stat st;
        if (p.empty() || stat(p.c_str(), &st) != 0) return 1;   // (a path that cannot be formed counts as missing, :128-134)
    }
    return 0;
} catch (...) { return (o_on_exception(), 1); }

size_t bhh_species_code(const char *common_name, char *out, size_t cap) try {
    return copy_out(species_code(common_name ? common_name : ""), out, cap);
} catch (...) { return (o_on_exception(), (size_t)0); }

size_t bhh_format_float(int kind, double value, char *out, size_t cap) try {
    std::string s;
    switch (kind) {
    case BHH_FLOAT_DISPLAY_F32: s = rust_display_f32((float)value); break;
    case BHH_FLOAT_DISPLAY_F64: s = rust_display_f64(value); break;
    case BHH_FLOAT_JSON_F32: s = json_f32((float)value); break;
    case BHH_FLOAT_JSON_F64: s = json_f64(value); break;
    default: return 0;
    }
    return copy_out(s, out, cap);
} catch (...) { return (o_on_exception(), (size_t)0); }

int bhh_writer_open(uint32_t format, const char *path, const bhh_writer_options *opt, bhh_writer **out) try {
    if (!path || !out) return ofail(BH_ERR_INVALID, "writer_open: null argument");
    *out = nullptr;
    auto h = std::make_unique<bhh_writer>();
    std::string err;
    int rc = writer_open(format, path, options_from(opt), h->w, err);
    if (rc != BH_OK) return ofail(rc, err);
    *out = h.release();
    return BH_OK;
} catch (...) { return o_on_exception(); }

int bhh_writer_write_header(bhh_writer *w) try {
    if (!w || !w->w) return ofail(BH_ERR_INVALID, "writer: null handle");
    int rc = writer_header(*w->w);
    return rc == BH_OK ? rc : ofail(rc, "write failed: " + w->w->path);
} catch (...) { return o_on_exception(); }

int bhh_writer_write_detection(bhh_writer *w, const char *label, float confidence, float start_time, float end_time,
                               const char *file_path) try {
    if (!w || !w->w || !label) return ofail(BH_ERR_INVALID, "writer: null argument");
    int rc = writer_detection(*w->w, detection_from_label(label, confidence, start_time, end_time, file_path ? file_path : ""));
    return rc == BH_OK ? rc : ofail(rc, "write failed: " + w->w->path);
} catch (...) { return o_on_exception(); }

int bhh_writer_finalize(bhh_writer *w) try {
    if (!w) return BH_OK;
    std::unique_ptr<bhh_writer> own(w);
    if (!w->w) return BH_OK;
    std::string err;
    int rc = writer_finalize(*w->w, err);
    return rc == BH_OK ? rc : ofail(rc, err);
} catch (...) { return o_on_exception(); }

// ---- reporter ----------------------------------------------------------------------------------------
int bhh_reporter_open(int mode, const char *path, bhh_reporter **out) try {
    if (!out || (mode != BHH_REPORT_NDJSON && mode != BHH_REPORT_JSON)) return ofail(BH_ERR_INVALID, "reporter_open: bad arguments");
    *out = nullptr;
    auto h = std::make_unique<bhh_reporter>();
    h->r.mode = mode;
    if (path && *path) {
        h->r.out = fopen(path, "wb");
        if (!h->r.out) return ofail(BH_ERR_IO, std::string("cannot create ") + path);
        h->r.owns = true;
    } else h->r.out = stdout;   // stdout is reserved for JSON (lib.rs:1121-1126)
    *out = h.release();
    return BH_OK;
} catch (...) { return o_on_exception(); }

void bhh_reporter_close(bhh_reporter *r) {
    if (!r) return;
    if (r->r.owns && r->r.out) fclose(r->r.out);
    delete r;
}

#define REPORTER_GUARD(rp) if (!(rp)) return; std::lock_guard<std::mutex> lock_((rp)->r.mu)

void bhh_reporter_pipeline_started(bhh_reporter *r, size_t total_files, const char *model, float min_confidence, const char *requested,
                                   const char *actual, const char *fallback_reason, const bhh_range_filter_info *rf) try {
    REPORTER_GUARD(r);
    std::string p = "{\"total_files\":" + std::to_string(total_files) + ",\"model\":" + json_string(model ? model : "") +
                    ",\"min_confidence\":" + json_f32(min_confidence) + ",\"execution_provider\":{\"requested\":" +
                    json_string(requested ? requested : "") + ",\"actual\":" + json_string(actual ? actual : "");
    if (fallback_reason && *fallback_reason) p += ",\"fallback_reason\":" + json_string(fallback_reason);
    p += "}";
    if (rf) {   // RangeFilterInfo (json_envelope.rs:176-193)
        p += ",\"range_filter\":{\"geomodel_version\":" + json_string(rf->geomodel_version ? rf->geomodel_version : "") +
             ",\"species_in_range\":" + std::to_string(rf->species_in_range) + ",\"total_species\":" + std::to_string(rf->total_species) +
             ",\"mapped_species\":" + std::to_string(rf->mapped_species) + ",\"unmatched_species\":" + std::to_string(rf->unmatched_species) +
             ",\"unmatched_policy\":" + json_string(rf->unmatched_policy ? rf->unmatched_policy : "") + ",\"threshold\":" + json_f32(rf->threshold) + "}";
    }
    r->r.emit("pipeline_started", p + "}");
} catch (...) { (void)o_on_exception(); }

void bhh_reporter_file_started(bhh_reporter *r, const char *file, size_t index, size_t estimated_segments, int has_duration,
                               double duration_seconds) try {
    REPORTER_GUARD(r);
    r->r.reset();
    std::string p = "{\"file\":" + json_string(file ? file : "") + ",\"index\":" + std::to_string(index) +
                    ",\"estimated_segments\":" + std::to_string(estimated_segments);
    if (has_duration) p += ",\"duration_seconds\":" + json_f64(duration_seconds);
    r->r.emit("file_started", p + "}");
} catch (...) { (void)o_on_exception(); }

int bhh_reporter_file_progress(bhh_reporter *r, const char *path, size_t segments_done, size_t segments_total, float percent) try {
    if (!r) return 0;
    std::lock_guard<std::mutex> lock_(r->r.mu);
    if (!r->r.should_emit(percent)) return 0;   // throttled on the file's percentage (reporter.rs:306-318)
    r->r.emit("progress", "{\"file\":{\"path\":" + json_string(path ? path : "") + ",\"segments_done\":" + std::to_string(segments_done) +
                              ",\"segments_total\":" + std::to_string(segments_total) + ",\"percent\":" + json_f32(percent) + "}}");
    return 1;
} catch (...) { return (o_on_exception(), 0); }

void bhh_reporter_batch_progress(bhh_reporter *r, size_t current, size_t total, float percent) try {
    REPORTER_GUARD(r);   // no file part: never throttled (`file.is_none_or(..)`)
    r->r.emit("progress", "{\"batch\":{\"current\":" + std::to_string(current) + ",\"total\":" + std::to_string(total) + ",\"percent\":" +
                              json_f32(percent) + "}}");
} catch (...) { (void)o_on_exception(); }

void bhh_reporter_file_completed(bhh_reporter *r, const char *file, int status, size_t detections, uint64_t duration_ms,
                                 const char *error_code, const char *error_message) try {
    REPORTER_GUARD(r);
    static const char *names[] = {"processed", "failed", "skipped", "locked"};   // FileStatus, snake_case
    if (status < 0 || status > 3) return;
    std::string p = "{\"file\":" + json_string(file ? file : "") + ",\"status\":\"" + names[status] + "\"";
    if (status == BHH_FILE_PROCESSED) p += ",\"detections\":" + std::to_string(detections) + ",\"duration_ms\":" + std::to_string(duration_ms);
    if (status == BHH_FILE_FAILED)
        p += ",\"error\":{\"code\":" + json_string(error_code ? error_code : "") + ",\"message\":" + json_string(error_message ? error_message : "") + "}";
    r->r.emit("file_completed", p + "}");
} catch (...) { (void)o_on_exception(); }

static std::string bsg_json(const bhh_bsg_metadata *b) {   // ,"bsg":{...} or nothing (skip_serializing_if = "Option::is_none")
    if (!b) return "";
    std::string p = std::string(",\"bsg\":{\"calibration_applied\":") + (b->calibration_applied ? "true" : "false") + ",\"sdm_applied\":" +
                    (b->sdm_applied ? "true" : "false");
    if (b->has_location) p += ",\"latitude\":" + json_f32(b->latitude) + ",\"longitude\":" + json_f32(b->longitude);
    if (b->has_day) p += ",\"day_of_year\":" + std::to_string(b->day_of_year);
    return p + "}";
}

void bhh_reporter_detections_bsg(bhh_reporter *r, const char *file, const char *const *labels, const float *confidence,
                                 const float *start_time, const float *end_time, size_t n, const bhh_bsg_metadata *bsg) try {
    REPORTER_GUARD(r);
    std::string p = "{\"file\":" + json_string(file ? file : "") + ",\"detections\":[";
    for (size_t i = 0; i < n; i++) {   // DetectionInfo (json_envelope.rs:379-395; reporter.rs:404-418)
        const Detection d = detection_from_label(labels[i], confidence[i], start_time[i], end_time[i], "");
        if (i) p += ",";
        p += "{\"species\":" + json_string(d.scientific_name + "_" + d.common_name) + ",\"common_name\":" + json_string(d.common_name) +
             ",\"scientific_name\":" + json_string(d.scientific_name) + ",\"confidence\":" + json_f32(d.confidence) + ",\"start_time\":" +
             json_f32(d.start_time) + ",\"end_time\":" + json_f32(d.end_time) + "}";
    }
    r->r.emit("detections", p + "]" + bsg_json(bsg) + "}");
} catch (...) { (void)o_on_exception(); }

void bhh_reporter_detections(bhh_reporter *r, const char *file, const char *const *labels, const float *confidence,
                             const float *start_time, const float *end_time, size_t n) {
    bhh_reporter_detections_bsg(r, file, labels, confidence, start_time, end_time, n, nullptr);
}

void bhh_reporter_pipeline_completed(bhh_reporter *r, size_t files_processed, size_t files_failed, size_t files_skipped,
                                     size_t total_detections, size_t total_segments, uint64_t duration_ms, double realtime_factor) try {
    REPORTER_GUARD(r);
    const char *status = files_failed == 0 ? "success" : files_processed > 0 ? "partial_success" : "failed";   // reporter.rs:363-369
    r->r.emit("pipeline_completed", std::string("{\"status\":\"") + status + "\",\"files_processed\":" + std::to_string(files_processed) +
                                        ",\"files_failed\":" + std::to_string(files_failed) + ",\"files_skipped\":" + std::to_string(files_skipped) +
                                        ",\"total_detections\":" + std::to_string(total_detections) + ",\"total_segments\":" +
                                        std::to_string(total_segments) + ",\"duration_ms\":" + std::to_string(duration_ms) +
                                        ",\"realtime_factor\":" + json_f64(realtime_factor) + "}");
    r->r.flush_json();
} catch (...) { (void)o_on_exception(); }

void bhh_reporter_error(bhh_reporter *r, const char *code, int fatal, const char *message, const char *suggestion) try {
    REPORTER_GUARD(r);
    std::string p = "{\"code\":" + json_string(code ? code : "") + ",\"severity\":\"" + (fatal ? "fatal" : "warning") + "\",\"message\":" +
                    json_string(message ? message : "");
    if (suggestion) p += ",\"suggestion\":" + json_string(suggestion);
    r->r.emit("error", p + "}");
} catch (...) { (void)o_on_exception(); }

}  // extern "C"

namespace bhh {
void reporter_detections(bhh_reporter *r, const std::string &file, const std::vector<Detection> &dets, const bhh_bsg_metadata *bsg) {
    if (!r) return;
    std::lock_guard<std::mutex> lock_(r->r.mu);
    std::string p = "{\"file\":" + json_string(file) + ",\"detections\":[";
    for (size_t i = 0; i < dets.size(); i++) {
        const Detection &d = dets[i];
        if (i) p += ",";
        p += "{\"species\":" + json_string(d.scientific_name + "_" + d.common_name) + ",\"common_name\":" + json_string(d.common_name) +
             ",\"scientific_name\":" + json_string(d.scientific_name) + ",\"confidence\":" + json_f32(d.confidence) + ",\"start_time\":" +
             json_f32(d.start_time) + ",\"end_time\":" + json_f32(d.end_time) + "}";
    }
    r->r.emit("detections", p + "]" + bsg_json(bsg) + "}");
}
}  // namespace bhh

// Shared between the host-side translation units (host_pipeline.cpp, host_output.cpp, host_parquet.cpp).  Not installed.
#pragma once
// (see kernels.hpp: A/B switches exist only in the EXPERIMENTS build)
#ifndef BH_XENV
#ifdef BIRDA_HIP_EXPERIMENTS
#define BH_XENV(name) getenv(name)
#else
#define BH_XENV(name) (static_cast<const char *>(nullptr))
#endif
#endif
#include <memory>
#include <string>
#include <vector>

#include "../../include/birda_hip.h"
#include "../../include/birda_host.h"

namespace bhh {

struct Detection {   // output/types.rs:8-23 (metadata is all-None on this path, types.rs:69-78)
    float start_time, end_time, confidence;
    std::string scientific_name, common_name, file_path;
};

struct WriterOptions {
    bool csv_bom = true;                 // lib.rs:1078
    std::string csv_columns;             // comma-separated additional columns (csv.rs:68-113; parquet.rs:151-168)
    std::string source_file, model;      // JsonResultWriter::new (json.rs:107-131)
    float min_confidence = 0.1f, overlap = 0.0f, audio_duration = 0.0f;
    bool has_lat = false, has_lon = false;
    double lat = 0.0, lon = 0.0;
    int week = -1;
};

void set_host_error(const std::string &msg);   // host_pipeline.cpp: the text bhh_last_error() returns

Detection detection_from_label(const std::string &label, float conf, float start, float end, const std::string &file_path);
std::string escape_csv(const std::string &v);
void append_fixed(std::string &o, double v, int places);   // "{:.N}" of v, appended (host_output.cpp)
std::string rust_display_f32(float v);
std::string rust_display_f64(double v);
std::string json_f32(float v);
std::string json_f64(double v);
std::string json_string(const std::string &s);
std::string rfc3339_now();
std::string species_code(const std::string &common_name);
bool path_parent(const std::string &path, std::string &parent);
bool path_file_name(const std::string &path, std::string &name);
std::string output_path_for(const std::string &input, const std::string &out_dir, uint32_t format);
int write_output(const std::string &input_path, const std::string &out_dir, uint32_t format, const std::vector<Detection> &detections,
                 const WriterOptions &opt, std::string &out_path, std::string &err);
int write_parquet_file(const std::string &path, const std::vector<Detection> &detections, const std::vector<std::string> &extra_columns,
                       std::string &err);
void reporter_detections(bhh_reporter *r, const std::string &file, const std::vector<Detection> &dets, const bhh_bsg_metadata *bsg);

}  // namespace bhh

/*
 * birda_host.h -- C exports of the host-side pipeline that sits above birda_hip.h.
 *
 * Mirrors the reference's per-file pipeline (src/pipeline/processor.rs:418-796) and the
 * pieces it is made of, so that tests can pin each piece against the oracle and against the
 * reference's own unit-test expectations.  Host logic only: every logit comes from
 * libbirda_hip's kernels through bh_predict* (birda_hip.h).
 */
#ifndef BIRDA_HOST_H
#define BIRDA_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "birda_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bh_decoder bh_decoder;

BH_API const char *bhh_last_error(void);

/* StreamingDecoder (reference src/audio/decode.rs:34-245); WAV PCM16/24/32/f32 only */
BH_API int bhh_decoder_open(const char *path, bh_decoder **out);                 /* open :54-128 */
BH_API void bhh_decoder_close(bh_decoder *d);
BH_API uint32_t bhh_decoder_sample_rate(const bh_decoder *d);                     /* :138-140 */
BH_API int bhh_decoder_duration_hint(const bh_decoder *d, double *secs);          /* :133-135; 0 = None */
BH_API int bhh_decoder_next_segment(bh_decoder *d, size_t segment_samples, size_t overlap_samples,
                                    float *out, size_t *start_sample);            /* :150-202 */

BH_API int64_t bhh_estimate_segment_count(int has_duration, double duration_secs, float segment_duration,
                                          float overlap);                         /* progress.rs:80-92 */
BH_API size_t bhh_effective_batch_size(size_t batch_size, int64_t estimated);     /* processor.rs:531-545 */
BH_API size_t bhh_source_samples(size_t target_samples, uint32_t source_rate, uint32_t target_rate); /* processor.rs:67-82 */
BH_API size_t bhh_duration_to_samples(float seconds, uint32_t rate);              /* processor.rs:514,520 */

/* inference watchdog (reference src/gpu/watchdog.rs:22-66; timeout processor.rs:194-211) */
BH_API uint64_t bhh_watchdog_timeout_secs(void);
BH_API void *bhh_watchdog_start(uint64_t timeout_ms, size_t batch_size);
BH_API void bhh_watchdog_cancel(void *guard);

/* CsvWriter (reference src/output/csv.rs:41-132) + Detection::from_label (types.rs:58-79) */
BH_API size_t bhh_csv_header(int bom, char *out, size_t cap);
BH_API size_t bhh_csv_row(const char *label, float start, float end, float conf, const char *path,
                          char *out, size_t cap);

/* ---- result writers (reference src/output/{csv,raven,...}.rs behind write_output, processor.rs:819-873) ------------ */
/* OutputFormat (config/types.rs:326-339) as a bit mask */
#define BHH_FORMAT_CSV 1u          /* <stem>.BirdNET.results.csv               csv.rs */
#define BHH_FORMAT_RAVEN 2u        /* <stem>.BirdNET.selection.table.txt       raven.rs */
#define BHH_FORMAT_AUDACITY 4u     /* <stem>.BirdNET.results.txt               audacity.rs */
#define BHH_FORMAT_KALEIDOSCOPE 8u /* <stem>.BirdNET.results.kaleidoscope.csv  kaleidoscope.rs */
#define BHH_FORMAT_JSON 16u        /* <stem>.BirdNET.json                      json.rs */
#define BHH_FORMAT_PARQUET 32u     /* <stem>.BirdNET.results.parquet           parquet.rs */
#define BHH_FORMAT_ALL 63u

typedef struct {
    int csv_bom;              /* CsvWriter::new include_bom */
    const char *csv_columns;  /* comma-separated additional columns (CSV header + empty cells; Parquet nullable columns) or NULL */
    const char *source_file;  /* JsonResultWriter::new(output_path, source_file, audio_duration, model, min_confidence, */
    const char *model;        /*                       overlap, lat, lon, week), json.rs:107-131 */
    float min_confidence, overlap, audio_duration;
    int has_lat, has_lon;
    double lat, lon;
    int week;                 /* -1 = None */
} bhh_writer_options;

typedef struct bhh_writer bhh_writer;
/* OutputWriter (writer.rs:7-18): new / write_header / write_detection / finalize; finalize also frees the handle.
 * write_detection takes what Detection::from_label takes (types.rs:58-79). */
BH_API int bhh_writer_open(uint32_t format, const char *path, const bhh_writer_options *opt, bhh_writer **out);
BH_API int bhh_writer_write_header(bhh_writer *w);
BH_API int bhh_writer_write_detection(bhh_writer *w, const char *label, float confidence, float start_time, float end_time,
                                      const char *file_path);
BH_API int bhh_writer_finalize(bhh_writer *w);
/* output_dir_for + output_path_for (coordinator.rs:41-94): stem sanitised, extension by format; output_dir NULL / "" = next
 * to the input.  Returns the length (0 = unknown format); out receives the text when cap allows. */
BH_API size_t bhh_output_path_for(const char *input_path, const char *output_dir, uint32_t format, char *out, size_t cap);
/* should_process (pipeline/coordinator.rs:96-143) without the lock-file arm: 0 = SkipExists (every requested format's
 * output file already exists and force is 0), 1 = Process (force, an output missing, or an empty format mask: "no output
 * can be found to already exist when none was asked for", :113-127).  The resume rule of directory mode. */
BH_API int bhh_should_process(const char *input_path, const char *output_dir, uint32_t format_mask, int force);
/* generate_species_code (raven.rs:72-84) */
BH_API size_t bhh_species_code(const char *common_name, char *out, size_t cap);
/* float formatting the writers rely on: Rust `{}` (core::fmt, shortest digits, no exponent) and serde_json (ryu) */
#define BHH_FLOAT_DISPLAY_F32 0
#define BHH_FLOAT_DISPLAY_F64 1
#define BHH_FLOAT_JSON_F32 2
#define BHH_FLOAT_JSON_F64 3
BH_API size_t bhh_format_float(int kind, double value, char *out, size_t cap);

/* ---- progress reporter (reference src/output/reporter.rs:22-420: JsonProgressReporter) ------------------
 * One JSON envelope {spec_version "1.1", timestamp, event, payload} per event (json_envelope.rs:13-61).  NDJSON mode
 * writes and flushes a line per event; JSON mode buffers and prints one array at pipeline_completed. */
#define BHH_REPORT_NDJSON 1
#define BHH_REPORT_JSON 2
typedef struct bhh_reporter bhh_reporter;
typedef struct {   /* RangeFilterInfo (json_envelope.rs:176-193) */
    const char *geomodel_version;
    size_t species_in_range, total_species, mapped_species, unmatched_species;
    const char *unmatched_policy;   /* "keep" | "drop" */
    float threshold;
} bhh_range_filter_info;
BH_API int bhh_reporter_open(int mode, const char *path /* NULL = stdout */, bhh_reporter **out);
BH_API void bhh_reporter_close(bhh_reporter *r);
BH_API void bhh_reporter_pipeline_started(bhh_reporter *r, size_t total_files, const char *model, float min_confidence,
                                          const char *requested, const char *actual, const char *fallback_reason /* NULL = None */,
                                          const bhh_range_filter_info *range_filter /* NULL = None */);
BH_API void bhh_reporter_file_started(bhh_reporter *r, const char *file, size_t index, size_t estimated_segments, int has_duration,
                                      double duration_seconds);
/* progress(None, Some(file)) is throttled (10 % or 500 ms; always at 0 % and 100 %): returns 1 when the event was written */
BH_API int bhh_reporter_file_progress(bhh_reporter *r, const char *path, size_t segments_done, size_t segments_total, float percent);
BH_API void bhh_reporter_batch_progress(bhh_reporter *r, size_t current, size_t total, float percent);
#define BHH_FILE_PROCESSED 0
#define BHH_FILE_FAILED 1
#define BHH_FILE_SKIPPED 2
#define BHH_FILE_LOCKED 3
BH_API void bhh_reporter_file_completed(bhh_reporter *r, const char *file, int status, size_t detections, uint64_t duration_ms,
                                        const char *error_code, const char *error_message);
BH_API void bhh_reporter_detections(bhh_reporter *r, const char *file, const char *const *labels, const float *confidence,
                                    const float *start_time, const float *end_time, size_t n);
/* BsgMetadata (output/json_envelope.rs:362-378), attached to the detections event whenever the classifier has a BSG processor
 * (processor.rs:741-768): calibration is always applied; the SDM only when latitude / longitude AND a day of year are known.
 * latitude / longitude / day_of_year are Option<>s: has_location / has_day = 0 leaves them out of the payload. */
typedef struct {
    int calibration_applied, sdm_applied;
    int has_location;
    float latitude, longitude;
    int has_day;
    uint32_t day_of_year;
} bhh_bsg_metadata;
BH_API void bhh_reporter_detections_bsg(bhh_reporter *r, const char *file, const char *const *labels, const float *confidence,
                                        const float *start_time, const float *end_time, size_t n, const bhh_bsg_metadata *bsg /* NULL = None */);
BH_API void bhh_reporter_pipeline_completed(bhh_reporter *r, size_t files_processed, size_t files_failed, size_t files_skipped,
                                            size_t total_detections, size_t total_segments, uint64_t duration_ms, double realtime_factor);
BH_API void bhh_reporter_error(bhh_reporter *r, const char *code, int fatal, const char *message, const char *suggestion /* NULL = None */);

/* ---- per-file pipeline --------------------------------------------------------------------------- */
/* Where decode + segmentation run:
 *   HOST    the reference's structure: a decode thread (decode_and_stream, processor.rs:49-108) feeds AudioChunks through a
 *           bounded channel to the batcher (run_streaming_inference, :114-190), which zero-pads the last batch (:240-258).
 *   DEVICE   the WAV file's frames (PCM16 / PCM24 / PCM32 / float32) are mapped, uploaded in the file's own sample
 *           layout and scaled / mixed / windowed / resampled on
 *           the GPU (bh_predict_pcm16); the same detections, no padding rows, no per-segment host work.
 *   AUTO    DEVICE when the file is a WAV of one of those four sample formats, else HOST (8-bit PCM, compressed audio). */
#define BHH_FRONT_END_AUTO 0u
#define BHH_FRONT_END_HOST 1u
#define BHH_FRONT_END_DEVICE 2u

/* ProcessingConfig (reference src/pipeline/config.rs:32-66), the fields this path uses */
typedef struct {
    const char *input_path;
    const char *output_dir;   /* NULL/"" = next to the input */
    const char *display_path; /* path text written in the File column; NULL = input_path */
    float min_confidence;     /* 0.1 */
    float overlap;            /* seconds, 0.0 */
    size_t batch_size;        /* CLI -b; 0 = bh_classifier_default_batch_size() (determine_default_batch_size, lib.rs:1035) */
    int csv_bom;              /* default on, lib.rs:1078 */
    uint32_t formats;         /* BHH_FORMAT_* mask; 0 = CSV (config default [csv], config/types.rs:185-205) */
    uint32_t front_end;       /* BHH_FRONT_END_* */
    const char *csv_columns;  /* config.csv_columns (config.rs:46) or NULL */
    const char *model_name;   /* JsonOutputConfig.model (processor.rs:800-815) */
    int has_lat, has_lon;     /* JsonOutputConfig lat / lon / week */
    double lat, lon;
    int week;                 /* -1 = None */
    bhh_reporter *reporter;   /* config.reporter (processor.rs:438): progress + detections events; NULL = none */
    int dual_output;          /* dual_output_mode (:721): with a reporter, also write the result files */
    /* config.custom_classifier / bat_mode (config.rs:63, lib.rs:773-774): when set, no resampling (the source samples go to
     * the model as they are: the "slow-down trick", processor.rs:464-475), segments of bat::CHUNK_SAMPLES = 144 000 samples
     * overlapping by a quarter (:502-508, constants.rs:525-542), and the custom classifier's predictions on the backbone's
     * embeddings replace the backbone's (:319-360, :369-372).  Runs on the HOST front end. */
    bh_custom_classifier *custom_classifier;
    /* BSG models (bh_classifier_set_bsg installed on the classifier): the metadata the stdout reporter attaches to its detections
     * event (processor.rs:741-768); NULL = the classifier has no BSG processor */
    const bhh_bsg_metadata *bsg;
} bhh_processing_config;

/* ProcessResult (processor.rs:877-886) + batching counters */
typedef struct {
    size_t detections, segments;
    double duration_secs, audio_duration_secs, segments_per_sec;
    size_t effective_batch, batches, padded_rows;
    char output_path[1024];   /* the first format written (lowest bit) */
    uint32_t front_end;       /* BHH_FRONT_END_HOST or BHH_FRONT_END_DEVICE: the one that ran */
    uint32_t formats_written;
} bhh_process_result;

BH_API int bhh_process_file(bh_classifier *clf, const bhh_processing_config *cfg, bhh_process_result *res);

/* process_files_sequential (lib.rs:1003-1100) for MANY SHORT recordings.  The reference runs the files one after another, each
 * in batches of its own segments; a one-minute recording is 20 segments -- 2 % of what this GPU wants per launch, and a forward
 * has a ~1.3 ms floor however few segments it holds.  Here consecutive WAV files of one sample format, rate and channel count are
 * PACKED: their streams are copied into one pinned buffer (each followed by a segment's length of silence, so that trailing
 * segments pad with zeros exactly as next_segment does), cut into up to `pack_segments` segments and run as ONE upload and ONE
 * forward (bh_predict_pcm16_at); the rows are scattered back and every file gets the detections, sort and outputs
 * bhh_process_file would have given it (tests/test_parity_gpu.py::test_packed_short_files_match_the_per_file_pipeline).
 * cfg is a template: input_path / display_path are ignored, the rest applies to every file.  Files the packer does not take --
 * not a WAV the device front end takes, a stream longer than a context's staging buffer (about 2 x pack_segments segments of mono
 * PCM16), bat mode, a reporter in the template -- go through
 * bhh_process_file where they stand.  results[i] belongs to paths[i]; status[i] (nullable) receives each file's BH_OK /
 * BH_ERR_*; the call itself fails only for bad arguments.  pack_segments 0 = 1024. */
BH_API int bhh_process_files(bh_classifier *clf, const bhh_processing_config *cfg, const char *const *paths, size_t n_files,
                             size_t pack_segments, bhh_process_result *results, int *status);

/* ---- directory mode (coordinator.rs:146-190) ------------------------------------------ */
/* is_audio_file (:179-190): extension in {wav, flac, mp3, m4a, aac}, ASCII case-insensitive. */
BH_API int bhh_is_audio_file(const char *path);
/* collect_input_files (:146-176): files are taken when they are audio files, directories are walked recursively, paths
 * that do not exist are skipped.  Writes the '\n'-separated list into out (nul-terminated) and returns the byte count
 * needed (call with cap = 0 to size the buffer), (size_t)-1 on an I/O error.  The reference keeps read_dir order, which
 * is unspecified; entries of one directory come back sorted by name here so that runs are repeatable. */
BH_API size_t bhh_collect_input_files(const char *const *paths, size_t n_paths, char *out, size_t cap, size_t *n_files);

/* ---- range filter tables (geomodel.rs; pure host logic, once per run) ------------------ */
/* scientific_name (geomodel.rs:28-33): length of the scientific-name prefix of a label. */
BH_API size_t bhh_scientific_name_len(const char *label);
/* SpeciesMapping::build (geomodel.rs:58-93) + GeomodelScores::project (:140-162) flattened onto class indices for
 * bh_classifier_set_range_filter: out_scores[c] = NaN when classifier label c has no geomodel entry, else the reported
 * score of its species (0 when the report omits it).  Matching is on the lower-cased scientific name (ASCII case
 * folding); of two classifier labels sharing one the first is mapped.  score_species / score_values are the geomodel's
 * LocationScores (birdnet_onnx RangeFilter::predict output, queried with threshold 0: classifier.rs:117-145).
 * mapped / in_range (nullable) receive MappingSummary's counts (mapped_count, in_range_count(threshold)). */
BH_API int bhh_project_scores(const char *const *geomodel_labels, size_t n_geomodel, const char *const *score_species,
                              const float *score_values, size_t n_scores, const char *const *classifier_labels,
                              size_t n_classes, float threshold, float *out_scores, size_t *mapped, size_t *in_range);

/* ---- the range filter's date arithmetic (reference src/utils/date.rs; config/range_filter.rs:106-123) ------------------
 * date_to_week (:21-33): floor((day_of_year - 1) / 7.6) + 1, capped at 48, on a non-leap calendar, in f32 as the reference
 * computes it; week_to_start_day (:57-70): (week - 1) * 7.6 + 1 truncated (one fused multiply-add in f32, `mul_add`);
 * day_of_year_to_date (:42-55): (month, day), saturating to December 31. */
BH_API uint32_t bhh_date_to_week(uint32_t month, uint32_t day);
BH_API uint32_t bhh_week_to_start_day(uint32_t week);
BH_API void bhh_day_of_year_to_date(uint32_t day_of_year, uint32_t *month, uint32_t *day);

#ifdef __cplusplus
}
#endif
#endif

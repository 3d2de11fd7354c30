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

/* ProcessingConfig (reference src/pipeline/config.rs:32-66), the fields this path uses */
typedef struct {
    const char *input_path;
    const char *output_dir;   /* NULL/"" = next to the input */
    const char *display_path; /* path text written in the File column; NULL = input_path */
    float min_confidence;     /* 0.1 */
    float overlap;            /* seconds, 0.0 */
    size_t batch_size;        /* CLI -b; 0 = 8 */
    int csv_bom;              /* default on, lib.rs:1078 */
} bhh_processing_config;

/* ProcessResult (processor.rs:877-886) + batching counters */
typedef struct {
    size_t detections, segments;
    double duration_secs, audio_duration_secs, segments_per_sec;
    size_t effective_batch, batches, padded_rows;
    char output_path[1024];
} bhh_process_result;

BH_API int bhh_process_file(bh_classifier *clf, const bhh_processing_config *cfg, bhh_process_result *res);

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

#ifdef __cplusplus
}
#endif
#endif

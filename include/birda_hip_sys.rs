// birda_hip_sys.rs -- GENERATED from include/birda_hip.h by tools/gen_rust_ffi.py; do not edit.
// The raw `extern "C"` surface of libbirda_hip.so for src/inference/hip_backend.rs.
#![allow(non_camel_case_types, dead_code)]
use std::ffi::{c_char, c_int, c_void};

pub const BH_MAX_TOP_K: usize = 32;
pub const BH_FLAG_LOW_LATENCY: u32 = 0x10;
pub const BH_FLAG_PRECISION_MASK: u32 = 0x3;
pub const BH_FLAG_AUTO: u32 = 0x0;
pub const BH_FLAG_F16X3: u32 = 0x1;
pub const BH_FLAG_F16: u32 = 0x2;
pub const BH_FLAG_F32: u32 = 0x3;
pub const BH_TOPK_NONFINITE: i32 = -2;
pub const BH_MODEL_BIRDNET_V24: u32 = 0;
pub const BH_MODEL_PERCH_V2: u32 = 1;
pub const BH_MODEL_BIRDNET_V30: u32 = 2;
pub const BH_MODEL_BSG_FINLAND: u32 = 3;
pub const BH_MIN_BATCH_SIZE: usize = 1;
pub const BH_MAX_BATCH_SIZE: usize = 512;
pub const BH_N_STAGES: usize = 9;
pub const BH_PCM_S16: u32 = 1;
pub const BH_PCM_S24: u32 = 2;
pub const BH_PCM_S32: u32 = 3;
pub const BH_PCM_F32: u32 = 4;
pub const BH_GATHER_AUTO: u32 = 0;
pub const BH_GATHER_HOST: u32 = 1;
pub const BH_GATHER_RCCL: u32 = 2;
pub const BH_OK: c_int = 0;
pub const BH_ERR_INVALID: c_int = -1;
pub const BH_ERR_IO: c_int = -2;
pub const BH_ERR_NO_DEVICE: c_int = -3;
pub const BH_ERR_HIP: c_int = -4;
pub const BH_ERR_LABELS: c_int = -5;
pub const BH_ERR_UNSUPPORTED: c_int = -6;
pub const BH_ERR_INTERNAL: c_int = -7;
pub const BH_ERR_NONFINITE: c_int = -8;

pub enum BhClassifier {}   // opaque handle `bh_classifier`
pub enum BhBatchContext {}   // opaque handle `bh_batch_context`
pub enum BhCustomClassifier {}   // opaque handle `bh_custom_classifier`
pub enum BhRangeFilter {}   // opaque handle `bh_range_filter`
pub enum BhMulti {}   // opaque handle `bh_multi`

#[repr(C)]
pub struct BhConfig {
    pub model_path: *const c_char,
    pub labels_path: *const c_char,
    pub top_k: u32,
    pub min_confidence: f32,
    pub device: i32,
    pub flags: u32,
}

#[repr(C)]
pub struct BhModelInfo {
    pub sample_rate: u32,
    pub segment_duration: f32,
    pub sample_count: u32,
    pub n_classes: u32,
    pub embedding_dim: u32,
    pub output_activation: u32,
    pub spec_channels: u32,
    pub spec_h: u32,
    pub spec_w: u32,
    pub n_layers: u32,
    pub macs_per_segment: u64,
    pub mel_flops_per_segment: u64,
    pub model_type: u32,
    pub precision: u32,
}

#[repr(C)]
pub struct BhResult {
    pub n_pred: u32,
    pub index: [i32; 32],
    pub confidence: [f32; 32],
}

#[repr(C)]
pub struct BhProviderStatus {
    pub requested: [c_char; 32],
    pub actual: [c_char; 32],
    pub fallback_reason: [c_char; 256],
    pub device: i32,
    pub device_count: u32,
    pub device_name: [c_char; 128],
    pub arch: [c_char; 32],
    pub compute_units: u32,
    pub hbm_bytes: u64,
}

#[repr(C)]
pub struct BhMultiConfig {
    pub model_path: *const c_char,
    pub labels_path: *const c_char,
    pub top_k: u32,
    pub min_confidence: f32,
    pub flags: u32,
    pub devices: *const i32,
    pub n_devices: u32,
    pub max_batch: u32,
    pub gather: u32,
}

pub type bh_rows_fn = Option<unsafe extern "C" fn(user: *mut c_void, first_segment: usize, n_segments: usize, rows: *const BhResult, start_samples: *const u64)>;

#[link(name = "birda_hip")]
extern "C" {
    pub fn bh_device_count() -> c_int;
    pub fn bh_backend_name() -> *const c_char;
    pub fn bh_last_error() -> *const c_char;
    pub fn bh_select_provider(requested: *const c_char, device_ordinal: i32, out: *mut BhProviderStatus) -> c_int;
    pub fn bh_classifier_provider_status(c: *const BhClassifier, out: *mut BhProviderStatus) -> c_int;
    pub fn bh_classifier_fallback_segments(c: *const BhClassifier) -> u64;
    pub fn bh_default_batch_size(model_type: u32, provider_actual: *const c_char) -> usize;
    pub fn bh_classifier_default_batch_size(c: *const BhClassifier) -> usize;
    pub fn bh_classifier_create(cfg: *const BhConfig, out: *mut *mut BhClassifier) -> c_int;
    pub fn bh_onnx_to_bhm(onnx_path: *const c_char, bhm_path: *const c_char) -> c_int;
    pub fn bh_onnx_eval(onnx_path: *const c_char, feed_name: *const c_char, feed: *const f64, feed_dims: *const i64, feed_rank: u32, target: *const c_char, out: *mut f64, out_cap: usize, out_dims: *mut i64, out_rank: *mut u32) -> c_int;
    pub fn bh_classifier_destroy(c: *mut BhClassifier);
    pub fn bh_classifier_info(c: *const BhClassifier, info: *mut BhModelInfo) -> c_int;
    pub fn bh_classifier_label(c: *const BhClassifier, index: u32) -> *const c_char;
    pub fn bh_classifier_ensure_warm(c: *mut BhClassifier, batch_size: usize) -> c_int;
    pub fn bh_classifier_is_warm(c: *const BhClassifier, batch_size: usize) -> c_int;
    pub fn bh_batch_context_create(c: *mut BhClassifier, max_batch: usize, out: *mut *mut BhBatchContext) -> c_int;
    pub fn bh_batch_context_destroy(ctx: *mut BhBatchContext);
    pub fn bh_batch_context_bytes(ctx: *const BhBatchContext) -> usize;
    pub fn bh_batch_context_host_buffer(ctx: *mut BhBatchContext, bytes: *mut usize) -> *mut c_void;
    pub fn bh_batch_context_set_sub_slices(ctx: *mut BhBatchContext, n: u32) -> c_int;
    pub fn bh_batch_context_device_bytes(ctx: *const BhBatchContext) -> usize;
    pub fn bh_batch_context_lane_fallbacks(ctx: *const BhBatchContext) -> u64;
    pub fn bh_classifier_trim(c: *mut BhClassifier) -> usize;
    pub fn bh_predict(c: *mut BhClassifier, segment: *const f32, n_samples: usize, out: *mut BhResult) -> c_int;
    pub fn bh_predict_batch(c: *mut BhClassifier, segments: *const *const f32, n: usize, n_samples: usize, out: *mut BhResult) -> c_int;
    pub fn bh_predict_batch_with_context(c: *mut BhClassifier, ctx: *mut BhBatchContext, segments: *const *const f32, n: usize, n_samples: usize, out: *mut BhResult) -> c_int;
    pub fn bh_predict_batch_contig(c: *mut BhClassifier, ctx: *mut BhBatchContext, base: *const f32, n: usize, out: *mut BhResult) -> c_int;
    pub fn bh_host_alloc(bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn bh_host_free(p: *mut c_void);
    pub fn bh_host_register(p: *mut c_void, bytes: usize) -> c_int;
    pub fn bh_host_unregister(p: *mut c_void) -> c_int;
    pub fn bh_predict_batch_logits(c: *mut BhClassifier, ctx: *mut BhBatchContext, base: *const f32, n: usize, logits: *mut f32, embeddings: *mut f32) -> c_int;
    pub fn bh_classifier_set_range_filter(c: *mut BhClassifier, scores: *const f32, n_classes: usize, threshold: f32, keep_unmatched: c_int, rerank: c_int) -> c_int;
    pub fn bh_classifier_set_species_list(c: *mut BhClassifier, keep: *const u8, n_classes: usize) -> c_int;
    pub fn bh_classifier_clear_filters(c: *mut BhClassifier) -> c_int;
    pub fn bh_topk_from_logits(c: *mut BhClassifier, logits: *const f32, n: usize, out: *mut BhResult) -> c_int;
    pub fn bh_forward_device(c: *mut BhClassifier, ctx: *mut BhBatchContext, d_segments: *const f32, n: usize, d_logits: *mut f32, d_topk_index: *mut i32, d_topk_conf: *mut f32) -> c_int;
    pub fn bh_batch_context_synchronize(ctx: *mut BhBatchContext) -> c_int;
    pub fn bh_batch_context_stream(ctx: *mut BhBatchContext) -> *mut c_void;
    pub fn bh_tensor_floats(c: *const BhClassifier, tensor: u32) -> u64;
    pub fn bh_batch_context_set_profiling(ctx: *mut BhBatchContext, enabled: c_int) -> c_int;
    pub fn bh_batch_context_stage_ms(ctx: *mut BhBatchContext, ms: *mut f32, launches: *mut u32) -> c_int;
    pub fn bh_batch_context_layer_ms(ctx: *mut BhBatchContext, ms: *mut f32, launches: *mut u32, n_layers: usize) -> c_int;
    pub fn bh_classifier_fused_blocks(c: *const BhClassifier, cfgs: *mut i32, cap: usize) -> c_int;
    pub fn bh_plan_fused_blocks(model_path: *const c_char, flags: u32, cfgs: *mut i32, layers: *mut i32, cap: usize) -> c_int;
    pub fn bh_classifier_frontend_kernel(c: *const BhClassifier, out: *mut c_char, cap: usize) -> c_int;
    pub fn bh_mb_config_name(cfg: i32, out: *mut c_char, cap: usize) -> c_int;
    pub fn bh_predict_batch_source_rate(c: *mut BhClassifier, ctx: *mut BhBatchContext, segments: *const *const f32, n: usize, n_src_samples: usize, source_rate: u32, out: *mut BhResult) -> c_int;
    pub fn bh_segment_starts(n_frames: usize, segment_samples: usize, overlap_samples: usize, starts: *mut u64, cap: usize) -> usize;
    pub fn bh_predict_pcm16(c: *mut BhClassifier, ctx: *mut BhBatchContext, pcm: *const i16, n_frames: usize, channels: u32, source_rate: u32, overlap_samples: usize, out: *mut BhResult, out_cap: usize, n_segments: *mut usize, start_samples: *mut u64) -> c_int;
    pub fn bh_predict_pcm(c: *mut BhClassifier, ctx: *mut BhBatchContext, pcm: *const c_void, sample_format: u32, n_frames: usize, channels: u32, source_rate: u32, overlap_samples: usize, out: *mut BhResult, out_cap: usize, n_segments: *mut usize, start_samples: *mut u64) -> c_int;
    pub fn bh_predict_pcm_at(c: *mut BhClassifier, ctx: *mut BhBatchContext, pcm: *const c_void, sample_format: u32, n_frames: usize, channels: u32, source_rate: u32, start_samples: *const u64, n_segments: usize, out: *mut BhResult) -> c_int;
    pub fn bh_predict_pcm_rows(c: *mut BhClassifier, ctx: *mut BhBatchContext, pcm: *const c_void, sample_format: u32, n_frames: usize, channels: u32, source_rate: u32, overlap_samples: usize, out: *mut BhResult, out_cap: usize, n_segments: *mut usize, start_samples: *mut u64, on_rows: bh_rows_fn, user: *mut c_void) -> c_int;
    pub fn bh_predict_pcm_fd_rows(c: *mut BhClassifier, ctx: *mut BhBatchContext, fd: c_int, file_offset: u64, sample_format: u32, n_frames: usize, channels: u32, source_rate: u32, overlap_samples: usize, out: *mut BhResult, out_cap: usize, n_segments: *mut usize, start_samples: *mut u64, on_rows: bh_rows_fn, user: *mut c_void) -> c_int;
    pub fn bh_predict_pcm16_at(c: *mut BhClassifier, ctx: *mut BhBatchContext, pcm: *const i16, n_frames: usize, channels: u32, source_rate: u32, start_samples: *const u64, n_segments: usize, out: *mut BhResult) -> c_int;
    pub fn bh_resample(c: *mut BhClassifier, in_: *const f32, n_in: usize, from_rate: u32, to_rate: u32, out: *mut f32, out_cap: usize, n_out: *mut usize) -> c_int;
    pub fn bh_resample_output_len(n_in: usize, from_rate: u32, to_rate: u32, n_out: *mut usize) -> c_int;
    pub fn bh_resample_supported(c: *mut BhClassifier, from_rate: u32, to_rate: u32) -> c_int;
    pub fn bh_resample_device(c: *mut BhClassifier, ctx: *mut BhBatchContext, d_in: *const f32, in_stride: usize, src_len: usize, from_rate: u32, to_rate: u32, d_out: *mut f32, out_stride: usize, out_len: usize, n_seg: usize) -> c_int;
    pub fn bh_custom_classifier_create(model_path: *const c_char, labels_path: *const c_char, device: i32, top_k: u32, out: *mut *mut BhCustomClassifier) -> c_int;
    pub fn bh_custom_classifier_destroy(cc: *mut BhCustomClassifier);
    pub fn bh_custom_classifier_num_classes(cc: *const BhCustomClassifier) -> u32;
    pub fn bh_custom_classifier_input_dim(cc: *const BhCustomClassifier) -> u32;
    pub fn bh_custom_classifier_label(cc: *const BhCustomClassifier, index: u32) -> *const c_char;
    pub fn bh_custom_classifier_predict_batch(cc: *mut BhCustomClassifier, embeddings: *const f32, n: usize, out: *mut BhResult) -> c_int;
    pub fn bh_predict_batch_two_stage(c: *mut BhClassifier, ctx: *mut BhBatchContext, cc: *mut BhCustomClassifier, segments: *const *const f32, n: usize, n_samples: usize, out: *mut BhResult, logits_out: *mut f32) -> c_int;
    pub fn bh_classifier_set_bsg(c: *mut BhClassifier, intercept: *const f32, slope: *const f32, prior: *const f32, n_classes: usize) -> c_int;
    pub fn bh_classifier_clear_bsg(c: *mut BhClassifier) -> c_int;
    pub fn bh_range_filter_create(model_path: *const c_char, labels_path: *const c_char, device: i32, threshold: f32, out: *mut *mut BhRangeFilter) -> c_int;
    pub fn bh_range_filter_destroy(rf: *mut BhRangeFilter);
    pub fn bh_range_filter_num_species(rf: *const BhRangeFilter) -> u32;
    pub fn bh_range_filter_label(rf: *const BhRangeFilter, index: u32) -> *const c_char;
    pub fn bh_range_filter_predict(rf: *mut BhRangeFilter, latitude: f64, longitude: f64, month: u32, day: u32, scores: *mut f32, cap: usize, indices: *mut u32, n_kept: *mut usize) -> c_int;
    pub fn bh_range_filter_predict_week(rf: *mut BhRangeFilter, latitude: f32, longitude: f32, week: f32, scores: *mut f32, cap: usize, indices: *mut u32, n_kept: *mut usize) -> c_int;
    pub fn bh_birdnet_week(month: u32, day: u32) -> u32;
    pub fn bh_multi_create(cfg: *const BhMultiConfig, out: *mut *mut BhMulti) -> c_int;
    pub fn bh_multi_destroy(m: *mut BhMulti);
    pub fn bh_multi_last_error() -> *const c_char;
    pub fn bh_multi_shards(m: *const BhMulti) -> u32;
    pub fn bh_multi_shard_device(m: *const BhMulti, shard: u32) -> c_int;
    pub fn bh_multi_gather_backend(m: *const BhMulti) -> *const c_char;
    pub fn bh_multi_classifier(m: *mut BhMulti, shard: u32) -> *mut BhClassifier;
    pub fn bh_multi_context(m: *mut BhMulti, shard: u32) -> *mut BhBatchContext;
    pub fn bh_shard_range(n_total: usize, shard: u32, n_shards: u32, lo: *mut usize, hi: *mut usize);
    pub fn bh_shard_ranges_weighted(weights: *const u64, n: usize, n_shards: u32, bounds: *mut usize) -> c_int;
    pub fn bh_multi_predict_batch_contig(m: *mut BhMulti, base: *const f32, n: usize, out: *mut BhResult) -> c_int;
    pub fn bh_multi_predict_batch_source_rate(m: *mut BhMulti, segments: *const *const f32, source_rates: *const u32, n_src_samples: *const usize, n: usize, out: *mut BhResult, bounds_out: *mut usize) -> c_int;
    pub fn bh_multi_forward_device(m: *mut BhMulti, d_segments: *const *const f32, n_per_shard: *const usize, out: *mut BhResult) -> c_int;
}

"""ctypes loader for libbirda_hip.so (C ABI: include/birda_hip.h).

Fails loudly when the HIP library is missing: there is no CPU fallback anywhere in the
product path.

Sharing a process with PyTorch (bench.py, the tests): the PyTorch ROCm wheel carries its own copy of the HIP / HSA
runtime, and the copy that is mapped first serves both.  PyTorch finds no GPU when that is not its own, so import
torch BEFORE the first classifier is created; the library itself is indifferent to which copy it runs on.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbirda_hip.so")
BH_MAX_TOP_K = 32
BH_N_STAGES = 9
STAGE_NAMES = ["minmax", "mel", "stem", "depthwise", "pointwise", "pool", "dense", "topk", "mbconv"]


class BhConfig(C.Structure):
    _fields_ = [("model_path", C.c_char_p), ("labels_path", C.c_char_p), ("top_k", C.c_uint32),
                ("min_confidence", C.c_float), ("device", C.c_int32), ("flags", C.c_uint32)]


class BhModelInfo(C.Structure):
    _fields_ = [("sample_rate", C.c_uint32), ("segment_duration", C.c_float), ("sample_count", C.c_uint32),
                ("n_classes", C.c_uint32), ("embedding_dim", C.c_uint32), ("output_activation", C.c_uint32),
                ("spec_channels", C.c_uint32), ("spec_h", C.c_uint32), ("spec_w", C.c_uint32),
                ("n_layers", C.c_uint32), ("macs_per_segment", C.c_uint64), ("mel_flops_per_segment", C.c_uint64),
                ("model_type", C.c_uint32), ("precision", C.c_uint32)]


class BhProviderStatus(C.Structure):
    """ExecutionProviderStatus (classifier.rs:23-30) + device facts."""
    _fields_ = [("requested", C.c_char * 32), ("actual", C.c_char * 32), ("fallback_reason", C.c_char * 256),
                ("device", C.c_int32), ("device_count", C.c_uint32), ("device_name", C.c_char * 128),
                ("arch", C.c_char * 32), ("compute_units", C.c_uint32), ("hbm_bytes", C.c_uint64)]


class BhMultiConfig(C.Structure):
    _fields_ = [("model_path", C.c_char_p), ("labels_path", C.c_char_p), ("top_k", C.c_uint32), ("min_confidence", C.c_float),
                ("flags", C.c_uint32), ("devices", C.POINTER(C.c_int32)), ("n_devices", C.c_uint32), ("max_batch", C.c_uint32),
                ("gather", C.c_uint32)]


class BhResult(C.Structure):
    _fields_ = [("n_pred", C.c_uint32), ("index", C.c_int32 * BH_MAX_TOP_K), ("confidence", C.c_float * BH_MAX_TOP_K)]


# every symbol include/birda_hip.h declares: (name, restype, argtypes)
_VP, _SZ = C.c_void_p, C.c_size_t
# bh_rows_fn: (user, first_segment, n_segments, rows, start_samples) -- rows of a host-fed stream as they complete
BhRowsFn = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(BhResult), C.POINTER(C.c_uint64))

SYMBOLS = [
    ("bh_device_count", C.c_int, []),
    ("bh_backend_name", C.c_char_p, []),
    ("bh_last_error", C.c_char_p, []),
    ("bh_select_provider", C.c_int, [C.c_char_p, C.c_int32, C.POINTER(BhProviderStatus)]),
    ("bh_classifier_provider_status", C.c_int, [_VP, C.POINTER(BhProviderStatus)]),
    ("bh_classifier_fallback_segments", C.c_uint64, [_VP]),
    ("bh_onnx_to_bhm", C.c_int, [C.c_char_p, C.c_char_p]),
    ("bh_onnx_eval", C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_uint32, C.c_char_p,
                               C.POINTER(C.c_double), _SZ, C.POINTER(C.c_int64), C.POINTER(C.c_uint32)]),
    ("bh_default_batch_size", _SZ, [C.c_uint32, C.c_char_p]),
    ("bh_classifier_default_batch_size", _SZ, [_VP]),
    ("bh_classifier_create", C.c_int, [C.POINTER(BhConfig), C.POINTER(_VP)]),
    ("bh_classifier_destroy", None, [_VP]),
    ("bh_classifier_info", C.c_int, [_VP, C.POINTER(BhModelInfo)]),
    ("bh_classifier_label", C.c_char_p, [_VP, C.c_uint32]),
    ("bh_classifier_ensure_warm", C.c_int, [_VP, _SZ]),
    ("bh_classifier_is_warm", C.c_int, [_VP, _SZ]),
    ("bh_batch_context_create", C.c_int, [_VP, _SZ, C.POINTER(_VP)]),
    ("bh_batch_context_destroy", None, [_VP]),
    ("bh_batch_context_bytes", _SZ, [_VP]),
    ("bh_batch_context_host_buffer", C.c_void_p, [_VP, C.POINTER(_SZ)]),
    ("bh_batch_context_set_sub_slices", C.c_int, [_VP, C.c_uint32]),
    ("bh_batch_context_device_bytes", _SZ, [_VP]),
    ("bh_classifier_trim", _SZ, [_VP]),
    ("bh_predict", C.c_int, [_VP, _VP, _SZ, C.POINTER(BhResult)]),
    ("bh_predict_batch", C.c_int, [_VP, C.POINTER(_VP), _SZ, _SZ, C.POINTER(BhResult)]),
    ("bh_predict_batch_with_context", C.c_int, [_VP, _VP, C.POINTER(_VP), _SZ, _SZ, C.POINTER(BhResult)]),
    ("bh_predict_batch_contig", C.c_int, [_VP, _VP, _VP, _SZ, C.POINTER(BhResult)]),
    ("bh_host_alloc", C.c_int, [_SZ, C.POINTER(C.c_void_p)]),
    ("bh_host_free", None, [_VP]),
    ("bh_host_register", C.c_int, [_VP, _SZ]),
    ("bh_host_unregister", C.c_int, [_VP]),
    ("bh_predict_batch_logits", C.c_int, [_VP, _VP, _VP, _SZ, _VP, _VP]),
    ("bh_forward_device", C.c_int, [_VP, _VP, _VP, _SZ, _VP, _VP, _VP]),
    ("bh_batch_context_synchronize", C.c_int, [_VP]),
    ("bh_batch_context_stream", _VP, [_VP]),
    ("bh_debug_read_tensor", C.c_int, [_VP, _VP, C.c_uint32, _VP, _SZ]),
    ("bh_tensor_floats", C.c_uint64, [_VP, C.c_uint32]),
    ("bh_batch_context_lane_fallbacks", C.c_uint64, [_VP]),
    ("bh_batch_context_set_profiling", C.c_int, [_VP, C.c_int]),
    ("bh_batch_context_stage_ms", C.c_int, [_VP, _VP, _VP]),
    ("bh_batch_context_layer_ms", C.c_int, [_VP, _VP, _VP, _SZ]),
    ("bh_classifier_fused_blocks", C.c_int, [_VP, _VP, _SZ]),
    ("bh_plan_fused_blocks", C.c_int, [C.c_char_p, C.c_uint32, _VP, _VP, _SZ]),
    ("bh_mb_config_name", C.c_int, [C.c_int32, C.c_char_p, _SZ]),
    ("bh_classifier_frontend_kernel", C.c_int, [_VP, C.c_char_p, _SZ]),
    ("bh_debug_mb_stamps", C.c_int, [_VP, _VP, _SZ]),
    ("bh_debug_gated_gemm", C.c_int, [C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _SZ, _SZ, _SZ, _SZ, C.c_int, C.c_int]),
    ("bh_predict_batch_source_rate", C.c_int, [_VP, _VP, C.POINTER(_VP), _SZ, _SZ, C.c_uint32, C.POINTER(BhResult)]),
    ("bh_segment_starts", _SZ, [_SZ, _SZ, _SZ, _VP, _SZ]),
    ("bh_predict_pcm16", C.c_int, [_VP, _VP, _VP, _SZ, C.c_uint32, C.c_uint32, _SZ, C.POINTER(BhResult), _SZ,
                                   C.POINTER(_SZ), _VP]),
    ("bh_predict_pcm", C.c_int, [_VP, _VP, _VP, C.c_uint32, _SZ, C.c_uint32, C.c_uint32, _SZ, C.POINTER(BhResult), _SZ, C.POINTER(_SZ), _VP]),
    ("bh_predict_pcm_rows", C.c_int, [_VP, _VP, _VP, C.c_uint32, _SZ, C.c_uint32, C.c_uint32, _SZ, C.POINTER(BhResult), _SZ, C.POINTER(_SZ), _VP,
                                      BhRowsFn, _VP]),
    ("bh_predict_pcm_fd_rows", C.c_int, [_VP, _VP, C.c_int, C.c_uint64, C.c_uint32, _SZ, C.c_uint32, C.c_uint32, _SZ, C.POINTER(BhResult), _SZ,
                                         C.POINTER(_SZ), _VP, BhRowsFn, _VP]),
    ("bh_predict_pcm_at", C.c_int, [_VP, _VP, _VP, C.c_uint32, _SZ, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), _SZ, C.POINTER(BhResult)]),
    ("bh_predict_pcm16_at", C.c_int, [_VP, _VP, _VP, _SZ, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), _SZ, C.POINTER(BhResult)]),
    ("bh_resample", C.c_int, [_VP, _VP, _SZ, C.c_uint32, C.c_uint32, _VP, _SZ, C.POINTER(_SZ)]),
    ("bh_resample_output_len", C.c_int, [_SZ, C.c_uint32, C.c_uint32, C.POINTER(_SZ)]),
    ("bh_resample_supported", C.c_int, [_VP, C.c_uint32, C.c_uint32]),
    ("bh_resample_device", C.c_int, [_VP, _VP, _VP, _SZ, _SZ, C.c_uint32, C.c_uint32, _VP, _SZ, _SZ, _SZ]),
    ("bh_classifier_set_range_filter", C.c_int, [_VP, _VP, _SZ, C.c_float, C.c_int, C.c_int]),
    ("bh_classifier_set_species_list", C.c_int, [_VP, _VP, _SZ]),
    ("bh_classifier_clear_filters", C.c_int, [_VP]),
    ("bh_topk_from_logits", C.c_int, [_VP, _VP, _SZ, C.POINTER(BhResult)]),
    ("bh_custom_classifier_create", C.c_int, [C.c_char_p, C.c_char_p, C.c_int32, C.c_uint32, C.POINTER(_VP)]),
    ("bh_custom_classifier_destroy", None, [_VP]),
    ("bh_custom_classifier_num_classes", C.c_uint32, [_VP]),
    ("bh_custom_classifier_input_dim", C.c_uint32, [_VP]),
    ("bh_custom_classifier_label", C.c_char_p, [_VP, C.c_uint32]),
    ("bh_custom_classifier_predict_batch", C.c_int, [_VP, _VP, _SZ, C.POINTER(BhResult)]),
    ("bh_predict_batch_two_stage", C.c_int, [_VP, _VP, _VP, C.POINTER(_VP), _SZ, _SZ, C.POINTER(BhResult), _VP]),
    ("bh_classifier_set_bsg", C.c_int, [_VP, _VP, _VP, _VP, _SZ]),
    ("bh_classifier_clear_bsg", C.c_int, [_VP]),
    ("bh_range_filter_create", C.c_int, [C.c_char_p, C.c_char_p, C.c_int32, C.c_float, C.POINTER(_VP)]),
    ("bh_range_filter_destroy", None, [_VP]),
    ("bh_range_filter_num_species", C.c_uint32, [_VP]),
    ("bh_range_filter_label", C.c_char_p, [_VP, C.c_uint32]),
    ("bh_range_filter_predict", C.c_int, [_VP, C.c_double, C.c_double, C.c_uint32, C.c_uint32, _VP, _SZ, _VP, C.POINTER(_SZ)]),
    ("bh_range_filter_predict_week", C.c_int, [_VP, C.c_float, C.c_float, C.c_float, _VP, _SZ, _VP, C.POINTER(_SZ)]),
    ("bh_birdnet_week", C.c_uint32, [C.c_uint32, C.c_uint32]),
    ("bh_multi_create", C.c_int, [C.POINTER(BhMultiConfig), C.POINTER(_VP)]),
    ("bh_multi_destroy", None, [_VP]),
    ("bh_multi_last_error", C.c_char_p, []),
    ("bh_multi_shards", C.c_uint32, [_VP]),
    ("bh_multi_shard_device", C.c_int, [_VP, C.c_uint32]),
    ("bh_multi_gather_backend", C.c_char_p, [_VP]),
    ("bh_multi_classifier", _VP, [_VP, C.c_uint32]),
    ("bh_multi_context", _VP, [_VP, C.c_uint32]),
    ("bh_shard_range", None, [_SZ, C.c_uint32, C.c_uint32, C.POINTER(_SZ), C.POINTER(_SZ)]),
    ("bh_shard_ranges_weighted", C.c_int, [_VP, _SZ, C.c_uint32, _VP]),
    ("bh_multi_predict_batch_contig", C.c_int, [_VP, _VP, _SZ, C.POINTER(BhResult)]),
    ("bh_multi_predict_batch_source_rate", C.c_int, [_VP, C.POINTER(_VP), _VP, _VP, _SZ, C.POINTER(BhResult), _VP]),
    ("bh_multi_forward_device", C.c_int, [_VP, C.POINTER(_VP), _VP, C.POINTER(BhResult)]),
]


FORMATS = {"csv": 1, "raven": 2, "table": 2, "audacity": 4, "kaleidoscope": 8, "json": 16, "parquet": 32}   # OutputFormat::from_str
FRONT_ENDS = {"auto": 0, "host": 1, "device": 2}
REPORT_NDJSON, REPORT_JSON = 1, 2
FLOAT_DISPLAY_F32, FLOAT_DISPLAY_F64, FLOAT_JSON_F32, FLOAT_JSON_F64 = 0, 1, 2, 3


class BhhWriterOptions(C.Structure):
    _fields_ = [("csv_bom", C.c_int), ("csv_columns", C.c_char_p), ("source_file", C.c_char_p), ("model", C.c_char_p),
                ("min_confidence", C.c_float), ("overlap", C.c_float), ("audio_duration", C.c_float),
                ("has_lat", C.c_int), ("has_lon", C.c_int), ("lat", C.c_double), ("lon", C.c_double), ("week", C.c_int)]


class BhhRangeFilterInfo(C.Structure):
    _fields_ = [("geomodel_version", C.c_char_p), ("species_in_range", C.c_size_t), ("total_species", C.c_size_t),
                ("mapped_species", C.c_size_t), ("unmatched_species", C.c_size_t), ("unmatched_policy", C.c_char_p),
                ("threshold", C.c_float)]


class BhhBsgMetadata(C.Structure):
    _fields_ = [("calibration_applied", C.c_int), ("sdm_applied", C.c_int), ("has_location", C.c_int), ("latitude", C.c_float),
                ("longitude", C.c_float), ("has_day", C.c_int), ("day_of_year", C.c_uint32)]


class BhhProcessingConfig(C.Structure):
    _fields_ = [("input_path", C.c_char_p), ("output_dir", C.c_char_p), ("display_path", C.c_char_p),
                ("min_confidence", C.c_float), ("overlap", C.c_float), ("batch_size", C.c_size_t),
                ("csv_bom", C.c_int), ("formats", C.c_uint32), ("front_end", C.c_uint32), ("csv_columns", C.c_char_p),
                ("model_name", C.c_char_p), ("has_lat", C.c_int), ("has_lon", C.c_int), ("lat", C.c_double),
                ("lon", C.c_double), ("week", C.c_int), ("reporter", C.c_void_p), ("dual_output", C.c_int),
                ("custom_classifier", C.c_void_p), ("bsg", C.POINTER(BhhBsgMetadata))]


class BhhProcessResult(C.Structure):
    _fields_ = [("detections", C.c_size_t), ("segments", C.c_size_t), ("duration_secs", C.c_double),
                ("audio_duration_secs", C.c_double), ("segments_per_sec", C.c_double),
                ("effective_batch", C.c_size_t), ("batches", C.c_size_t), ("padded_rows", C.c_size_t),
                ("output_path", C.c_char * 1024), ("front_end", C.c_uint32), ("formats_written", C.c_uint32)]


# every symbol include/birda_host.h declares
HOST_SYMBOLS = [
    ("bhh_last_error", C.c_char_p, []),
    ("bhh_decoder_open", C.c_int, [C.c_char_p, C.POINTER(_VP)]),
    ("bhh_decoder_close", None, [_VP]),
    ("bhh_decoder_sample_rate", C.c_uint32, [_VP]),
    ("bhh_decoder_duration_hint", C.c_int, [_VP, C.POINTER(C.c_double)]),
    ("bhh_decoder_next_segment", C.c_int, [_VP, _SZ, _SZ, _VP, C.POINTER(_SZ)]),
    ("bhh_estimate_segment_count", C.c_int64, [C.c_int, C.c_double, C.c_float, C.c_float]),
    ("bhh_effective_batch_size", _SZ, [_SZ, C.c_int64]),
    ("bhh_source_samples", _SZ, [_SZ, C.c_uint32, C.c_uint32]),
    ("bhh_duration_to_samples", _SZ, [C.c_float, C.c_uint32]),
    ("bhh_watchdog_timeout_secs", C.c_uint64, []),
    ("bhh_watchdog_start", _VP, [C.c_uint64, _SZ]),
    ("bhh_watchdog_cancel", None, [_VP]),
    ("bhh_csv_header", _SZ, [C.c_int, C.c_char_p, _SZ]),
    ("bhh_csv_row", _SZ, [C.c_char_p, C.c_float, C.c_float, C.c_float, C.c_char_p, C.c_char_p, _SZ]),
    ("bhh_process_file", C.c_int, [_VP, C.POINTER(BhhProcessingConfig), C.POINTER(BhhProcessResult)]),
    ("bhh_process_files", C.c_int, [_VP, C.POINTER(BhhProcessingConfig), C.POINTER(C.c_char_p), _SZ, _SZ, C.POINTER(BhhProcessResult), C.POINTER(C.c_int)]),
    ("bhh_writer_open", C.c_int, [C.c_uint32, C.c_char_p, C.POINTER(BhhWriterOptions), C.POINTER(_VP)]),
    ("bhh_writer_write_header", C.c_int, [_VP]),
    ("bhh_writer_write_detection", C.c_int, [_VP, C.c_char_p, C.c_float, C.c_float, C.c_float, C.c_char_p]),
    ("bhh_writer_finalize", C.c_int, [_VP]),
    ("bhh_output_path_for", _SZ, [C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p, _SZ]),
    ("bhh_should_process", C.c_int, [C.c_char_p, C.c_char_p, C.c_uint32, C.c_int]),
    ("bhh_species_code", _SZ, [C.c_char_p, C.c_char_p, _SZ]),
    ("bhh_format_float", _SZ, [C.c_int, C.c_double, C.c_char_p, _SZ]),
    ("bhh_reporter_open", C.c_int, [C.c_int, C.c_char_p, C.POINTER(_VP)]),
    ("bhh_reporter_close", None, [_VP]),
    ("bhh_reporter_pipeline_started", None, [_VP, _SZ, C.c_char_p, C.c_float, C.c_char_p, C.c_char_p, C.c_char_p,
                                             C.POINTER(BhhRangeFilterInfo)]),
    ("bhh_reporter_file_started", None, [_VP, C.c_char_p, _SZ, _SZ, C.c_int, C.c_double]),
    ("bhh_reporter_file_progress", C.c_int, [_VP, C.c_char_p, _SZ, _SZ, C.c_float]),
    ("bhh_reporter_batch_progress", None, [_VP, _SZ, _SZ, C.c_float]),
    ("bhh_reporter_file_completed", None, [_VP, C.c_char_p, C.c_int, _SZ, C.c_uint64, C.c_char_p, C.c_char_p]),
    ("bhh_reporter_detections", None, [_VP, C.c_char_p, _VP, _VP, _VP, _VP, _SZ]),
    ("bhh_reporter_detections_bsg", None, [_VP, C.c_char_p, _VP, _VP, _VP, _VP, _SZ, C.POINTER(BhhBsgMetadata)]),
    ("bhh_reporter_pipeline_completed", None, [_VP, _SZ, _SZ, _SZ, _SZ, _SZ, C.c_uint64, C.c_double]),
    ("bhh_reporter_error", None, [_VP, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p]),
    ("bhh_is_audio_file", C.c_int, [C.c_char_p]),
    ("bhh_collect_input_files", _SZ, [_VP, _SZ, C.c_char_p, _SZ, C.POINTER(_SZ)]),
    ("bhh_scientific_name_len", _SZ, [C.c_char_p]),
    ("bhh_date_to_week", C.c_uint32, [C.c_uint32, C.c_uint32]),
    ("bhh_week_to_start_day", C.c_uint32, [C.c_uint32]),
    ("bhh_day_of_year_to_date", None, [C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("bhh_project_scores", C.c_int, [_VP, _SZ, _VP, _VP, _SZ, _VP, _SZ, C.c_float, _VP, C.POINTER(_SZ), C.POINTER(_SZ)]),
]

_lib = None


def load():
    """Load libbirda_hip.so; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C birda_amd/csrc` "
                           "(the HIP hot path has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS + HOST_SYMBOLS:
        fn = getattr(L, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


class BirdaHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libbirda_hip error {code}: {msg}")
        self.code = code


def check(rc: int):
    if rc != 0:
        raise BirdaHipError(rc, load().bh_last_error().decode("utf-8", "replace"))

"""Python handles on the C++ host pipeline (include/birda_host.h), named after the
reference's items: `StreamingDecoder` (src/audio/decode.rs:34), `process_file`
(src/pipeline/processor.rs:418), `estimate_segment_count` (src/output/progress.rs:80)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import BhhProcessingConfig, BhhProcessResult, BirdaHipError


def _hcheck(rc: int):
    if rc != 0:
        L = _lib.load()
        raise BirdaHipError(rc, L.bhh_last_error().decode("utf-8", "replace"))


class StreamingDecoder:
    def __init__(self, path: str):
        self._L = _lib.load()
        h = C.c_void_p()
        _hcheck(self._L.bhh_decoder_open(path.encode(), C.byref(h)))
        self._h = h

    def sample_rate(self) -> int:
        return int(self._L.bhh_decoder_sample_rate(self._h))

    def duration_hint(self) -> Optional[float]:
        d = C.c_double()
        return float(d.value) if self._L.bhh_decoder_duration_hint(self._h, C.byref(d)) else None

    def next_segment(self, segment_samples: int, overlap_samples: int) -> Optional[Tuple[np.ndarray, int]]:
        buf = np.empty(max(segment_samples, 1), np.float32)
        start = C.c_size_t()
        rc = self._L.bhh_decoder_next_segment(self._h, segment_samples, overlap_samples, buf.ctypes.data, C.byref(start))
        if rc < 0:
            _hcheck(rc)
        if rc == 0:
            return None
        return buf[:segment_samples], int(start.value)

    def close(self):
        if getattr(self, "_h", None):
            self._L.bhh_decoder_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def estimate_segment_count(duration_secs: Optional[float], segment_duration: float, overlap: float) -> Optional[int]:
    L = _lib.load()
    v = L.bhh_estimate_segment_count(int(duration_secs is not None), duration_secs or 0.0, segment_duration, overlap)
    return None if v < 0 else int(v)


def effective_batch_size(batch_size: int, estimated: Optional[int]) -> int:
    return int(_lib.load().bhh_effective_batch_size(batch_size, -1 if estimated is None else estimated))


def source_samples(target_samples: int, source_rate: int, target_rate: int) -> int:
    return int(_lib.load().bhh_source_samples(target_samples, source_rate, target_rate))


def csv_header(bom: bool = True) -> bytes:
    buf = C.create_string_buffer(256)
    n = _lib.load().bhh_csv_header(int(bom), buf, 256)
    return buf.raw[:n]


def csv_row(label: str, start: float, end: float, conf: float, path: str) -> bytes:
    buf = C.create_string_buffer(8192)
    n = _lib.load().bhh_csv_row(label.encode("utf-8"), start, end, conf, path.encode("utf-8"), buf, 8192)
    return buf.raw[:n]


def scientific_name(label: str) -> str:
    """geomodel.rs:28-33."""
    raw = label.encode("utf-8")
    return raw[:_lib.load().bhh_scientific_name_len(raw)].decode("utf-8")


@dataclass
class MappingSummary:   # classifier.rs:128-131 (MappingSummary::new)
    mapped: int
    total: int
    unmatched: int
    in_range: int


def project_scores(geomodel_labels: List[str], reported: List[Tuple[str, float]], classifier_labels: List[str],
                   threshold: float = 0.0) -> Tuple[np.ndarray, MappingSummary]:
    """SpeciesMapping::build + GeomodelScores::project (geomodel.rs:58-162) onto class indices: the table
    BirdClassifier.set_range_filter takes (NaN = no geomodel entry)."""
    def arr(items):
        raw = [s.encode("utf-8") for s in items]
        return (C.c_char_p * max(1, len(raw)))(*raw), raw
    g, _g = arr(geomodel_labels)
    sp, _s = arr([r[0] for r in reported])
    c, _c = arr(classifier_labels)
    vals = np.asarray([r[1] for r in reported] or [0.0], np.float32)
    out = np.zeros(max(1, len(classifier_labels)), np.float32)
    mapped, in_range = C.c_size_t(), C.c_size_t()
    _hcheck(_lib.load().bhh_project_scores(g, len(geomodel_labels), sp, vals.ctypes.data, len(reported), c,
                                           len(classifier_labels), threshold, out.ctypes.data, C.byref(mapped), C.byref(in_range)))
    n = len(classifier_labels)
    return out[:n].copy(), MappingSummary(mapped.value, n, n - mapped.value, in_range.value)


def bsg_metadata(latitude: Optional[float] = None, longitude: Optional[float] = None, day_of_year: Optional[int] = None):
    """BsgMetadata as process_file builds it (processor.rs:741-768): calibration always applied; with a location the SDM is
    applied only when the day of year is known too."""
    loc = latitude is not None and longitude is not None
    return _lib.BhhBsgMetadata(1, int(loc and day_of_year is not None), int(loc), float(latitude or 0.0), float(longitude or 0.0),
                               int(loc and day_of_year is not None), int(day_of_year or 0))


def format_mask(formats: Sequence[str]) -> int:
    """OutputFormat::from_str (config/types.rs:354-370) onto the BHH_FORMAT_* bits."""
    mask = 0
    for f in formats:
        if f.lower() not in _lib.FORMATS:
            raise ValueError(f"InvalidOutputFormat: {f}")
        mask |= _lib.FORMATS[f.lower()]
    return mask


def output_path_for(input_path: str, output_dir: Optional[str], fmt: str) -> str:
    """coordinator.rs:41-94."""
    buf = C.create_string_buffer(4096)
    _lib.load().bhh_output_path_for(input_path.encode(), output_dir.encode() if output_dir else None,
                                    _lib.FORMATS[fmt], buf, 4096)
    return buf.value.decode()


def species_code(common_name: str) -> str:
    """generate_species_code (raven.rs:72-84)."""
    buf = C.create_string_buffer(64)
    _lib.load().bhh_species_code(common_name.encode("utf-8"), buf, 64)
    return buf.value.decode("utf-8")


def format_float(kind: int, value: float) -> str:
    buf = C.create_string_buffer(512)
    _lib.load().bhh_format_float(kind, value, buf, 512)
    return buf.value.decode()


class OutputWriter:
    """One of the six writers behind write_output (processor.rs:819-873): write_header / write_detection / finalize."""

    def __init__(self, fmt: str, path: str, csv_bom: bool = True, csv_columns: Optional[Sequence[str]] = None,
                 source_file: str = "", model: str = "", min_confidence: float = 0.1, overlap: float = 0.0,
                 audio_duration: float = 0.0, lat: Optional[float] = None, lon: Optional[float] = None,
                 week: Optional[int] = None):
        self._L = _lib.load()
        self._keep = [",".join(csv_columns).encode() if csv_columns else None, source_file.encode(), model.encode()]
        opt = _lib.BhhWriterOptions(int(csv_bom), self._keep[0], self._keep[1], self._keep[2], min_confidence, overlap,
                                    audio_duration, int(lat is not None), int(lon is not None), lat or 0.0, lon or 0.0,
                                    -1 if week is None else week)
        h = C.c_void_p()
        _hcheck(self._L.bhh_writer_open(_lib.FORMATS[fmt], path.encode(), C.byref(opt), C.byref(h)))
        self._h = h

    def write_header(self):
        _hcheck(self._L.bhh_writer_write_header(self._h))

    def write_detection(self, label: str, confidence: float, start_time: float, end_time: float, file_path: str):
        _hcheck(self._L.bhh_writer_write_detection(self._h, label.encode("utf-8"), confidence, start_time, end_time,
                                                   file_path.encode("utf-8")))

    def finalize(self):
        h, self._h = self._h, None
        if h:
            _hcheck(self._L.bhh_writer_finalize(h))


class ProgressReporter:
    """JsonProgressReporter (reporter.rs:170-420): NDJSON (a line per event) or JSON (one array at the end)."""

    def __init__(self, mode: str = "ndjson", path: Optional[str] = None):
        self._L = _lib.load()
        h = C.c_void_p()
        _hcheck(self._L.bhh_reporter_open({"ndjson": _lib.REPORT_NDJSON, "json": _lib.REPORT_JSON}[mode],
                                          path.encode() if path else None, C.byref(h)))
        self._h = h

    def pipeline_started(self, total_files, model, min_confidence, requested, actual, fallback_reason=None, range_filter=None):
        rf = None
        if range_filter is not None:
            rf = _lib.BhhRangeFilterInfo(range_filter["geomodel_version"].encode(), range_filter["species_in_range"],
                                         range_filter["total_species"], range_filter["mapped_species"],
                                         range_filter["unmatched_species"], range_filter["unmatched_policy"].encode(),
                                         range_filter["threshold"])
        self._L.bhh_reporter_pipeline_started(self._h, total_files, model.encode(), min_confidence, requested.encode(),
                                              actual.encode(), fallback_reason.encode() if fallback_reason else None,
                                              C.byref(rf) if rf is not None else None)

    def file_started(self, file, index, estimated_segments, duration_seconds=None):
        self._L.bhh_reporter_file_started(self._h, file.encode(), index, estimated_segments,
                                          int(duration_seconds is not None), duration_seconds or 0.0)

    def file_progress(self, path, segments_done, segments_total, percent) -> bool:
        return bool(self._L.bhh_reporter_file_progress(self._h, path.encode(), segments_done, segments_total, percent))

    def batch_progress(self, current, total, percent):
        self._L.bhh_reporter_batch_progress(self._h, current, total, percent)

    def file_completed_success(self, file, detections, duration_ms):
        self._L.bhh_reporter_file_completed(self._h, file.encode(), 0, detections, duration_ms, None, None)

    def file_completed_failure(self, file, error_code, error_message):
        self._L.bhh_reporter_file_completed(self._h, file.encode(), 1, 0, 0, error_code.encode(), error_message.encode())

    def file_skipped(self, file, locked=False):
        self._L.bhh_reporter_file_completed(self._h, file.encode(), 3 if locked else 2, 0, 0, None, None)

    def detections(self, file, dets, bsg=None):
        """dets: (label, confidence, start_time, end_time) tuples; bsg: bsg_metadata(...) for BSG models (BsgMetadata,
        json_envelope.rs:362-378) or None."""
        n = len(dets)
        raw = [d[0].encode("utf-8") for d in dets]
        labels = (C.c_char_p * max(1, n))(*raw)
        conf = np.asarray([d[1] for d in dets] or [0.0], np.float32)
        st = np.asarray([d[2] for d in dets] or [0.0], np.float32)
        en = np.asarray([d[3] for d in dets] or [0.0], np.float32)
        self._L.bhh_reporter_detections_bsg(self._h, file.encode(), labels, conf.ctypes.data, st.ctypes.data, en.ctypes.data, n,
                                            C.byref(bsg) if bsg is not None else None)

    def pipeline_completed(self, files_processed, files_failed, files_skipped, total_detections, total_segments,
                           duration_ms, realtime_factor):
        self._L.bhh_reporter_pipeline_completed(self._h, files_processed, files_failed, files_skipped, total_detections,
                                                total_segments, duration_ms, realtime_factor)

    def error(self, code, fatal, message, suggestion=None):
        self._L.bhh_reporter_error(self._h, code.encode(), int(fatal), message.encode(),
                                   suggestion.encode() if suggestion else None)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.bhh_reporter_close(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class ProcessResult:
    detections: int
    segments: int
    duration_secs: float
    audio_duration_secs: float
    segments_per_sec: float
    effective_batch: int
    batches: int
    padded_rows: int
    output_path: str
    front_end: str = "host"
    formats_written: int = 0


def process_file(classifier, input_path: str, output_dir: Optional[str] = None, min_confidence: float = 0.1,
                 overlap: float = 0.0, batch_size: int = 0, csv_bom: bool = True,
                 display_path: Optional[str] = None, formats: Sequence[str] = ("csv",), front_end: str = "auto",
                 csv_columns: Optional[Sequence[str]] = None, model_name: str = "", lat: Optional[float] = None,
                 lon: Optional[float] = None, week: Optional[int] = None, reporter: Optional[ProgressReporter] = None,
                 dual_output: bool = False, custom_classifier=None, bsg=None) -> ProcessResult:
    """process_file (processor.rs:418-796).  batch_size 0 = the backend's default (determine_default_batch_size);
    front_end "host" keeps the reference's decode-thread + padded-batch structure, "device" / "auto" run decode
    scaling, mono mix, segmentation and resampling on the GPU for WAV input (PCM16 / PCM24 / PCM32 / float32).  custom_classifier: bat mode
    (classifier.CustomClassifier): no resampling, 144 000-sample segments overlapping by a quarter, the custom
    classifier's predictions on the backbone's embeddings."""
    L = _lib.load()
    cfg = BhhProcessingConfig(input_path.encode(), output_dir.encode() if output_dir else None,
                              display_path.encode() if display_path else None, min_confidence, overlap, batch_size,
                              int(csv_bom), format_mask(formats), _lib.FRONT_ENDS[front_end],
                              ",".join(csv_columns).encode() if csv_columns else None, model_name.encode(),
                              int(lat is not None), int(lon is not None), lat or 0.0, lon or 0.0,
                              -1 if week is None else week, reporter._h if reporter is not None else None,
                              int(dual_output), custom_classifier._h if custom_classifier is not None else None,
                              C.pointer(bsg) if bsg is not None else None)
    res = BhhProcessResult()
    _hcheck(L.bhh_process_file(classifier._h, C.byref(cfg), C.byref(res)))
    return ProcessResult(res.detections, res.segments, res.duration_secs, res.audio_duration_secs,
                         res.segments_per_sec, res.effective_batch, res.batches, res.padded_rows,
                         res.output_path.decode(), "device" if res.front_end == 2 else "host", res.formats_written)


def process_files_packed(classifier, files: Sequence[str], output_dir: Optional[str] = None, min_confidence: float = 0.1,
                         overlap: float = 0.0, csv_bom: bool = True, formats: Sequence[str] = ("csv",),
                         csv_columns: Optional[Sequence[str]] = None, model_name: str = "", pack_segments: int = 0):
    """bhh_process_files: many short recordings, packed into shared uploads / forwards (include/birda_host.h); the results and
    outputs of `process_file` per file.  Returns (results, status) in the order of `files`."""
    L = _lib.load()
    cfg = BhhProcessingConfig(None, output_dir.encode() if output_dir else None, None, min_confidence, overlap, 0,
                              int(csv_bom), format_mask(formats), _lib.FRONT_ENDS["auto"],
                              ",".join(csv_columns).encode() if csv_columns else None, model_name.encode(),
                              0, 0, 0.0, 0.0, -1, None, 0, None, None)
    n = len(files)
    raw = [f.encode() for f in files]
    arr = (C.c_char_p * max(1, n))(*raw)
    res = (BhhProcessResult * max(1, n))()
    status = (C.c_int * max(1, n))()
    _hcheck(L.bhh_process_files(classifier._h, C.byref(cfg), arr, n, pack_segments, res, status))
    out = [ProcessResult(r.detections, r.segments, r.duration_secs, r.audio_duration_secs, r.segments_per_sec, r.effective_batch,
                         r.batches, r.padded_rows, r.output_path.decode(), "device" if r.front_end == 2 else "host", r.formats_written)
           for r in res[:n]]
    return out, [int(status[i]) for i in range(n)]


def collect_input_files(paths: List[str]) -> List[str]:
    """coordinator.rs:146-176: audio files given directly or found under the given directories."""
    L = _lib.load()
    raw = [p.encode() for p in paths]
    arr = (C.c_char_p * max(1, len(raw)))(*raw)
    n = C.c_size_t()
    need = L.bhh_collect_input_files(arr, len(raw), None, 0, C.byref(n))
    if need == C.c_size_t(-1).value:
        _hcheck(-2)
    buf = C.create_string_buffer(need)
    L.bhh_collect_input_files(arr, len(raw), buf, need, C.byref(n))
    return buf.value.decode().split("\n") if n.value else []


def should_process(input_path: str, output_dir: Optional[str], formats=None, force: bool = False) -> bool:
    """`should_process` of the reference (pipeline/coordinator.rs:96-143) without the lock-file arm: False when every requested
    format's output already exists and `force` is off (the resume rule of directory mode)."""
    return bool(_lib.load().bhh_should_process(input_path.encode(), output_dir.encode() if output_dir else None,
                                                format_mask(formats if formats is not None else ("csv",)), 1 if force else 0))


def process_files(classifier, files: List[str], rank: int = 0, world: int = 1, durations: Optional[List[float]] = None,
                  force: bool = True, packed: bool = False, **kwargs) -> List[ProcessResult]:
    """Directory mode on `world` processes (one per GPU): rank g takes the files `sharding.assign_by_duration` gives
    it (the reference scales out as N processes over one directory with lock files, file_lock.rs:36-88)
    and runs them through `process_file` one after another on its own classifier, as `process_files_sequential`
    does (lib.rs:1003-1100).  force=False applies the reference's resume rule first (`should_process`, coordinator.rs:96-143):
    files whose requested outputs all exist are left out before the list is cut."""
    from . import sharding
    if not force:
        keep = [i for i, f in enumerate(files) if should_process(f, kwargs.get("output_dir"), kwargs.get("formats"), False)]
        files = [files[i] for i in keep]
        if durations is not None:
            durations = [durations[i] for i in keep]
    if durations is None:
        durations = []
        for f in files:
            d = StreamingDecoder(f)
            durations.append(d.duration_hint() or 0.0)
            d.close()
    mine = sharding.assign_by_duration(durations, world)[rank]
    if packed:     # short recordings share uploads and forwards (bhh_process_files); same outputs
        allowed = {"output_dir", "min_confidence", "overlap", "csv_bom", "formats", "csv_columns", "model_name", "pack_segments"}
        extra = set(kwargs) - allowed - {"batch_size"}
        if extra:
            raise ValueError(f"process_files(packed=True) does not take {sorted(extra)}")
        kw = {k: v for k, v in kwargs.items() if k in allowed}
        res, status = process_files_packed(classifier, [files[i] for i in mine], **kw)
        for f, st in zip([files[i] for i in mine], status):
            if st != 0:
                raise BirdaHipError(st, f"process_files: {f}")
        return res
    return [process_file(classifier, files[i], **kwargs) for i in mine]

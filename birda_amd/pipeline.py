"""Python handles on the C++ host pipeline (include/birda_host.h), named after the
reference's items: `StreamingDecoder` (src/audio/decode.rs:34), `process_file`
(src/pipeline/processor.rs:418), `estimate_segment_count` (src/output/progress.rs:80)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Tuple

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


def process_file(classifier, input_path: str, output_dir: Optional[str] = None, min_confidence: float = 0.1,
                 overlap: float = 0.0, batch_size: int = 8, csv_bom: bool = True,
                 display_path: Optional[str] = None) -> ProcessResult:
    L = _lib.load()
    cfg = BhhProcessingConfig(input_path.encode(), output_dir.encode() if output_dir else None,
                              display_path.encode() if display_path else None, min_confidence, overlap, batch_size,
                              int(csv_bom))
    res = BhhProcessResult()
    _hcheck(L.bhh_process_file(classifier._h, C.byref(cfg), C.byref(res)))
    return ProcessResult(res.detections, res.segments, res.duration_secs, res.audio_duration_secs,
                         res.segments_per_sec, res.effective_batch, res.batches, res.padded_rows,
                         res.output_path.decode())


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


def process_files(classifier, files: List[str], rank: int = 0, world: int = 1, durations: Optional[List[float]] = None,
                  **kwargs) -> List[ProcessResult]:
    """Directory mode on `world` processes (one per GPU): rank g takes the files `sharding.assign_by_duration` gives
    it (the reference scales out as N processes over one directory with lock files, file_lock.rs:36-88)
    and runs them through `process_file` one after another on its own classifier, as `process_files_sequential`
    does (lib.rs:1003-1100)."""
    from . import sharding
    if durations is None:
        durations = []
        for f in files:
            d = StreamingDecoder(f)
            durations.append(d.duration_hint() or 0.0)
            d.close()
    mine = sharding.assign_by_duration(durations, world)[rank]
    return [process_file(classifier, files[i], **kwargs) for i in mine]

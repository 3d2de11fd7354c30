"""Host-side mirror of the reference's `BirdClassifier` over the C ABI.

Same names, argument meaning and error behaviour as reference
`src/inference/classifier.rs:191-646` (`sample_rate`, `segment_duration`, `sample_count`,
`ensure_warm`, `predict`, `predict_batch`, `create_batch_context`,
`predict_batch_with_context`), so parity tests read like the reference's own.  All compute
happens in libbirda_hip.so; nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
import weakref
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import BhConfig, BhModelInfo, BhResult, check

DEFAULT_TOP_K = 5            # reference src/constants.rs:178
DEFAULT_MIN_CONFIDENCE = 0.1  # reference src/constants.rs:25


@dataclass
class Prediction:
    """birdnet_onnx::Prediction{species, confidence, index} (processor.rs:372)."""
    species: str
    confidence: float
    index: int


@dataclass
class PredictionResult:
    predictions: List[Prediction]


class PinnedSegments:
    """[n][sample_count] float32 in pinned host memory (bh_host_alloc): `array` is a numpy view to fill; handing it to
    `predict_batch_contig` uploads straight from it, without the gather copy pageable input needs."""

    def __init__(self, n: int, sample_count: int):
        self._L = _lib.load()
        self._p = C.c_void_p()
        check(self._L.bh_host_alloc(n * sample_count * 4, C.byref(self._p)))
        buf = (C.c_float * (n * sample_count)).from_address(self._p.value)
        self.array = np.frombuffer(buf, np.float32).reshape(n, sample_count)

    def close(self) -> None:
        if self._p:
            self.array = None
            self._L.bh_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass


class BatchInferenceContext:
    """create_batch_context(max_batch_size) -- classifier.rs:559-565."""

    def __init__(self, classifier: "BirdClassifier", max_batch_size: int):
        self._L = classifier._L
        self.classifier = classifier
        self.max_batch_size = max_batch_size
        h = C.c_void_p()
        check(self._L.bh_batch_context_create(classifier._h, max_batch_size, C.byref(h)))
        self._h = h
        classifier._register_context(self)

    def input_buffer_bytes(self) -> int:  # processor.rs:588
        return int(self._L.bh_batch_context_bytes(self._h))

    def device_bytes(self) -> int:
        return int(self._L.bh_batch_context_device_bytes(self._h))

    def synchronize(self):
        check(self._L.bh_batch_context_synchronize(self._h))

    def stream(self) -> int:
        return int(self._L.bh_batch_context_stream(self._h) or 0)

    def set_profiling(self, on: bool):
        check(self._L.bh_batch_context_set_profiling(self._h, int(on)))

    def stage_ms(self):
        ms = (C.c_float * _lib.BH_N_STAGES)()
        n = (C.c_uint32 * _lib.BH_N_STAGES)()
        check(self._L.bh_batch_context_stage_ms(self._h, ms, n))
        return {name: (float(ms[i]), int(n[i])) for i, name in enumerate(_lib.STAGE_NAMES)}

    def layer_ms(self):
        """[(ms, launches)] per layer of the last profiled forward (fused blocks on their expand layer)."""
        n = int(self.classifier.info.n_layers)
        ms = (C.c_float * n)()
        cnt = (C.c_uint32 * n)()
        check(self._L.bh_batch_context_layer_ms(self._h, ms, cnt, n))
        return [(float(ms[i]), int(cnt[i])) for i in range(n)]

    @classmethod
    def borrow(cls, classifier: "BirdClassifier", handle, max_batch_size: int = 0) -> "BatchInferenceContext":
        """A view of a context something else owns (a bh_multi shard's): close() leaves it alone."""
        self = cls.__new__(cls)
        self._L, self.classifier, self.max_batch_size = classifier._L, classifier, max_batch_size
        self._h, self._borrowed = C.c_void_p(handle), True
        return self

    def set_sub_slices(self, n: int):
        """bh_batch_context_set_sub_slices: 0 automatic, 1 whole slices, n equal sub-slices (results do not depend on it)."""
        check(self._L.bh_batch_context_set_sub_slices(self._h, n))

    def close(self):
        if getattr(self, "_h", None):
            # (a context outlives its classifier only by accident -- two module globals collected in the wrong order at interpreter
            #  exit: the classifier's close() has closed it already, see BirdClassifier.close)
            if not getattr(self, "_borrowed", False) and getattr(self.classifier, "_h", None):
                self._L.bh_batch_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# BH_FLAG_* (include/birda_hip.h): "auto" = split-f16 compute with rows that leave the f16 range re-run on the f32 kernels
PRECISION_FLAGS = {"auto": 0, "f16x3": 1, "f16": 2, "f32": 3}


class BirdClassifier:
    def __init__(self, model_path: str, labels_path: Optional[str] = None, top_k: int = DEFAULT_TOP_K,
                 min_confidence: float = DEFAULT_MIN_CONFIDENCE, device: int = 0, precision: str = "auto", low_latency: bool = False):
        self._L = _lib.load()
        self._keep = (model_path.encode(), labels_path.encode() if labels_path else None)
        flags = PRECISION_FLAGS[precision] | (0x10 if low_latency else 0)   # BH_FLAG_* (include/birda_hip.h); 0x10 = BH_FLAG_LOW_LATENCY
        cfg = BhConfig(self._keep[0], self._keep[1], top_k, min_confidence, device, flags)
        h = C.c_void_p()
        check(self._L.bh_classifier_create(C.byref(cfg), C.byref(h)))
        self._h = h
        info = BhModelInfo()
        check(self._L.bh_classifier_info(self._h, C.byref(info)))
        self.info = info
        self.top_k = top_k
        self.min_confidence = min_confidence
        self.device = device

    # ---- ModelConfig accessors (classifier.rs:360-377) ----
    def sample_rate(self) -> int:
        return int(self.info.sample_rate)

    def segment_duration(self) -> float:
        return float(self.info.segment_duration)

    def sample_count(self) -> int:
        return int(self.info.sample_count)

    def n_classes(self) -> int:
        return int(self.info.n_classes)

    def label(self, index: int) -> Optional[str]:
        s = self._L.bh_classifier_label(self._h, index)
        return s.decode("utf-8") if s is not None else None

    def fused_blocks(self) -> List[int]:
        """Tile-configuration index of every expand->depthwise->project block that runs fused."""
        buf = (C.c_int32 * 256)()
        n = int(self._L.bh_classifier_fused_blocks(self._h, buf, 256))
        return [int(buf[i]) for i in range(min(n, 256))]

    def fused_kernel_name(self, cfg: int, se: bool = False) -> str:
        """The block's instantiation as rocprofv3 prints it: the 19 tile arguments of bh_mb_config_name and the twentieth, SE (1 for
        pass A of a squeeze-excite block)."""
        buf = C.create_string_buffer(128)
        self._L.bh_mb_config_name(cfg, buf, 128)
        name = buf.value.decode()
        return "mbconv<" + name + ("," + str(int(se)) if name else "") + ">"

    def mel_kernel_name(self) -> str:
        """The front-end kernel instantiation as a profiler prints it (e.g. "bh::mel_kernel<6, 3>")."""
        buf = C.create_string_buffer(128)
        self._L.bh_classifier_frontend_kernel(self._h, buf, 128)
        return buf.value.decode()

    def provider_status(self):
        """ExecutionProviderStatus{requested, actual, fallback_reason} (classifier.rs:23-30) + device facts."""
        st = _lib.BhProviderStatus()
        check(self._L.bh_classifier_provider_status(self._h, C.byref(st)))
        return st

    def fallback_segments(self) -> int:
        """Segments BH_FLAG_AUTO has re-run on the f32 kernels so far."""
        return int(self._L.bh_classifier_fallback_segments(self._h))

    def default_batch_size(self) -> int:
        """determine_default_batch_size (lib.rs:256-288) for this backend."""
        return int(self._L.bh_classifier_default_batch_size(self._h))

    def trim(self) -> int:
        """Release the parked batch contexts and the internal one; returns the device bytes freed."""
        return int(self._L.bh_classifier_trim(self._h))

    # ---- warm-up (classifier.rs:414-466) ----
    def ensure_warm(self, batch_size: int):
        check(self._L.bh_classifier_ensure_warm(self._h, batch_size))

    def is_warm(self, batch_size: int) -> bool:
        return bool(self._L.bh_classifier_is_warm(self._h, batch_size))

    # ---- inference ----
    def _results(self, arr, n) -> List[PredictionResult]:
        out = []
        for i in range(n):
            r = arr[i]
            preds = []
            for k in range(r.n_pred):
                idx = int(r.index[k])
                preds.append(Prediction(self.label(idx) or str(idx), float(r.confidence[k]), idx))
            out.append(PredictionResult(preds))
        return out

    def predict(self, segment: np.ndarray) -> PredictionResult:
        seg = np.ascontiguousarray(segment, np.float32)
        res = BhResult()
        check(self._L.bh_predict(self._h, seg.ctypes.data, seg.size, C.byref(res)))
        return self._results([res], 1)[0]

    def _ptrs(self, segments: Sequence[np.ndarray]):
        keep = [np.ascontiguousarray(s, np.float32) for s in segments]
        lens = {k.size for k in keep}
        n_samples = lens.pop() if len(lens) == 1 else -1
        if n_samples < 0:
            # mixed lengths: let the library reject the first wrong one with its own message
            n_samples = keep[0].size if keep[0].size != self.sample_count() else next(
                k.size for k in keep if k.size != self.sample_count())
        ptrs = (C.c_void_p * len(keep))(*[k.ctypes.data for k in keep])
        return keep, ptrs, n_samples

    def predict_batch(self, segments: Sequence[np.ndarray]) -> List[PredictionResult]:
        n = len(segments)
        if n == 0:
            return []
        keep, ptrs, ns = self._ptrs(segments)
        res = (BhResult * n)()
        check(self._L.bh_predict_batch(self._h, ptrs, n, ns, res))
        return self._results(res, n)

    def create_batch_context(self, max_batch_size: int) -> BatchInferenceContext:
        return BatchInferenceContext(self, max_batch_size)

    def predict_batch_with_context(self, ctx: BatchInferenceContext, segments: Sequence[np.ndarray]):
        n = len(segments)
        if n == 0:
            return []
        keep, ptrs, ns = self._ptrs(segments)
        res = (BhResult * n)()
        check(self._L.bh_predict_batch_with_context(self._h, ctx._h, ptrs, n, ns, res))
        return self._results(res, n)

    def predict_logits(self, ctx: BatchInferenceContext, segments: np.ndarray, want_embeddings: bool = False):
        segs = np.ascontiguousarray(segments, np.float32).reshape(-1, self.sample_count())
        n = segs.shape[0]
        logits = np.empty((n, self.n_classes()), np.float32)
        emb = np.empty((n, int(self.info.embedding_dim)), np.float32) if want_embeddings else None
        check(self._L.bh_predict_batch_logits(self._h, ctx._h, segs.ctypes.data, n, logits.ctypes.data,
                                              emb.ctypes.data if want_embeddings else None))
        return (logits, emb) if want_embeddings else logits

    def forward_device(self, ctx: BatchInferenceContext, d_segments: int, n: int, d_logits: int,
                       d_topk_index: int = 0, d_topk_conf: int = 0):
        """Device pointers in, device pointers out; enqueued on the context stream."""
        check(self._L.bh_forward_device(self._h, ctx._h, d_segments, n, d_logits, d_topk_index or None,
                                        d_topk_conf or None))

    # ---- decoded PCM16 stream in, per-segment results out (device-side front-end) ----
    def segment_starts(self, n_frames: int, segment_samples: int, overlap_samples: int) -> List[int]:
        n = int(self._L.bh_segment_starts(n_frames, segment_samples, overlap_samples, None, 0))
        buf = (C.c_uint64 * max(n, 1))()
        self._L.bh_segment_starts(n_frames, segment_samples, overlap_samples, buf, n)
        return [int(buf[i]) for i in range(n)]

    def predict_pcm16(self, ctx: BatchInferenceContext, pcm: np.ndarray, source_rate: int, overlap_samples: int = 0, on_rows=None):
        """pcm: int16 [frames] or [frames, channels].  Returns (results, start_samples).
        on_rows(first_segment, results, start_samples): bh_predict_pcm_rows -- called for each finished run of consecutive
        segments, in order, while the device computes the later ones."""
        a = np.ascontiguousarray(pcm, np.int16)
        channels = 1 if a.ndim == 1 else a.shape[1]
        n_frames = a.shape[0]
        cap = n_frames // max(1, self.sample_count() // 4) + 8   # generous: overlap < segment
        cap = max(cap, 8)
        while True:
            res = (BhResult * cap)()
            starts = (C.c_uint64 * cap)()
            n = C.c_size_t()
            if on_rows is None:
                rc = self._L.bh_predict_pcm16(self._h, ctx._h, a.ctypes.data, n_frames, channels, source_rate, overlap_samples,
                                              res, cap, C.byref(n), starts)
            else:
                raised = []   # an exception inside the ctypes trampoline would only be printed: kept, and re-raised after the call

                def _cb(_user, first, count, rows, st):
                    if raised:
                        return
                    try:
                        on_rows(int(first), self._results(rows, int(count)), [int(st[i]) for i in range(int(count))])
                    except BaseException as e:   # noqa: BLE001 -- handed back to the caller below
                        raised.append(e)
                cb = _lib.BhRowsFn(_cb)
                rc = self._L.bh_predict_pcm_rows(self._h, ctx._h, a.ctypes.data, 1, n_frames, channels, source_rate, overlap_samples,
                                                 res, cap, C.byref(n), starts, cb, None)
                if raised and not (rc != 0 and int(n.value) > cap):
                    raise raised[0]
            if rc != 0 and int(n.value) > cap:
                cap = int(n.value)
                continue
            check(rc)
            return self._results(res, int(n.value)), [int(starts[i]) for i in range(int(n.value))]

    # ---- resampler (reference src/audio/resample.rs:10-105) ----
    def resample(self, samples: np.ndarray, from_rate: int, to_rate: int) -> np.ndarray:
        x = np.ascontiguousarray(samples, np.float32)
        n_out = C.c_size_t()
        check(self._L.bh_resample_output_len(x.size, from_rate, to_rate, C.byref(n_out)))
        out = np.empty(max(int(n_out.value), 1), np.float32)
        got = C.c_size_t()
        check(self._L.bh_resample(self._h, x.ctypes.data, x.size, from_rate, to_rate, out.ctypes.data, out.size, C.byref(got)))
        return out[: int(got.value)]

    def resample_device(self, ctx: BatchInferenceContext, d_in: int, in_stride: int, src_len: int, from_rate: int,
                        to_rate: int, d_out: int, out_stride: int, out_len: int, n_seg: int):
        check(self._L.bh_resample_device(self._h, ctx._h, d_in, in_stride, src_len, from_rate, to_rate, d_out,
                                         out_stride, out_len, n_seg))

    def read_tensor(self, ctx: BatchInferenceContext, tensor: int, n: int) -> np.ndarray:
        nfl = int(self._L.bh_tensor_floats(self._h, tensor))
        out = np.empty((n, nfl), np.float32)
        check(self._L.bh_debug_read_tensor(self._h, ctx._h, tensor, out.ctypes.data, out.size))
        return out

    # ---- range filter / species list (classifier.rs:587-645), applied on the device to the kept top-k -------
    def set_range_filter(self, scores: np.ndarray, threshold: float, unmatched: str = "keep", rerank: bool = False):
        """scores[c]: geomodel occurrence score of class c, NaN where the species has no geomodel entry
        (pipeline.project_scores builds it).  unmatched: "keep" | "drop" (config UnmatchedPolicy)."""
        if unmatched not in ("keep", "drop"):
            raise ValueError("unmatched must be 'keep' or 'drop'")
        a = np.ascontiguousarray(scores, np.float32)
        check(self._L.bh_classifier_set_range_filter(self._h, a.ctypes.data, a.size, threshold, int(unmatched == "keep"), int(rerank)))

    def set_species_list(self, keep: np.ndarray):
        a = np.ascontiguousarray(np.asarray(keep) != 0, np.uint8)
        check(self._L.bh_classifier_set_species_list(self._h, a.ctypes.data, a.size))

    def clear_filters(self):
        check(self._L.bh_classifier_clear_filters(self._h))

    # ---- BSG post-processing (classifier.rs:508-545) on the kept top-k, on the device ------------------------
    def set_bsg(self, intercept: np.ndarray, slope: np.ndarray, prior: Optional[np.ndarray] = None):
        """Per-class logistic calibration conf' = sigmoid(intercept + slope * logit(conf)), times the SDM occurrence prior
        when one is given (has_bsg_processor / apply_bsg_postprocessing)."""
        a = np.ascontiguousarray(intercept, np.float32)
        b = np.ascontiguousarray(slope, np.float32)
        p = np.ascontiguousarray(prior, np.float32) if prior is not None else None
        check(self._L.bh_classifier_set_bsg(self._h, a.ctypes.data, b.ctypes.data, p.ctypes.data if p is not None else None, a.size))

    def clear_bsg(self):
        check(self._L.bh_classifier_clear_bsg(self._h))

    # ---- two-stage inference (bat mode, processor.rs:319-360) -------------------------------------------
    def predict_batch_two_stage(self, ctx: BatchInferenceContext, custom: "CustomClassifier", segments: Sequence[np.ndarray],
                                want_logits: bool = False):
        n = len(segments)
        keep, ptrs, ns = self._ptrs(segments)
        res = (BhResult * max(1, n))()
        logits = np.empty((n, custom.num_classes()), np.float32) if want_logits else None
        check(self._L.bh_predict_batch_two_stage(self._h, ctx._h, custom._h, ptrs, n, ns, res,
                                                 logits.ctypes.data if want_logits else None))
        out = custom._results(res, n)
        return (out, logits) if want_logits else out

    def topk_from_logits(self, logits: np.ndarray) -> List[PredictionResult]:
        a = np.ascontiguousarray(logits, np.float32).reshape(-1, self.n_classes())
        arr = (BhResult * max(1, a.shape[0]))()
        check(self._L.bh_topk_from_logits(self._h, a.ctypes.data, a.shape[0], arr))
        return self._results(arr, a.shape[0])

    @classmethod
    def borrow(cls, handle, top_k: int = DEFAULT_TOP_K, min_confidence: float = DEFAULT_MIN_CONFIDENCE, device: int = 0) -> "BirdClassifier":
        """A view of a classifier something else owns (a bh_multi shard's replica): close() leaves it alone."""
        self = cls.__new__(cls)
        self._L = _lib.load()
        self._h, self._borrowed = C.c_void_p(handle), True
        info = BhModelInfo()
        check(self._L.bh_classifier_info(self._h, C.byref(info)))
        self.info, self.top_k, self.min_confidence, self.device = info, top_k, min_confidence, device
        return self

    def _register_context(self, ctx) -> None:
        if not hasattr(self, "_contexts"):
            self._contexts = weakref.WeakSet()
        self._contexts.add(ctx)

    def close(self):
        if getattr(self, "_h", None):
            # batch contexts hold pointers into their classifier (stream sets, parked-context slots): they go first, whichever
            # object the garbage collector happens to finalise first
            for ctx in list(getattr(self, "_contexts", ())):
                ctx.close()
            if not getattr(self, "_borrowed", False):
                self._L.bh_classifier_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CustomClassifier:
    """birdnet_onnx::CustomClassifier (reference src/lib.rs:883-901): a small dense model on the backbone's embeddings."""

    def __init__(self, model_path: str, labels_path: Optional[str] = None, device: int = 0, top_k: int = 0):
        self._L = _lib.load()
        h = C.c_void_p()
        check(self._L.bh_custom_classifier_create(model_path.encode(), labels_path.encode() if labels_path else None, device,
                                                  top_k, C.byref(h)))
        self._h = h

    def num_classes(self) -> int:
        return int(self._L.bh_custom_classifier_num_classes(self._h))

    def input_dim(self) -> int:
        return int(self._L.bh_custom_classifier_input_dim(self._h))

    def label(self, index: int) -> Optional[str]:
        raw = self._L.bh_custom_classifier_label(self._h, index)
        return raw.decode("utf-8") if raw is not None else None

    def _results(self, arr, n) -> List[PredictionResult]:
        out = []
        for i in range(n):
            r = arr[i]
            out.append(PredictionResult([Prediction(self.label(int(r.index[k])) or str(int(r.index[k])), float(r.confidence[k]),
                                                    int(r.index[k])) for k in range(r.n_pred)]))
        return out

    def predict_batch(self, embeddings: np.ndarray) -> List[PredictionResult]:
        e = np.ascontiguousarray(embeddings, np.float32).reshape(-1, self.input_dim())
        res = (BhResult * max(1, e.shape[0]))()
        check(self._L.bh_custom_classifier_predict_batch(self._h, e.ctypes.data, e.shape[0], res))
        return self._results(res, e.shape[0])

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.bh_custom_classifier_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RangeFilter:
    """birda's RangeFilter (reference src/inference/range_filter.rs:8-51): the geomodel query on the device.
    `model_path` is the geomodel's .onnx file (read inside the library) or a BHC1 container; `labels_path` the GEOMODEL's
    labels."""

    def __init__(self, model_path: str, labels_path: str, threshold: float = 0.0, device: int = 0):
        self._L = _lib.load()
        h = C.c_void_p()
        check(self._L.bh_range_filter_create(model_path.encode(), labels_path.encode(), device, threshold, C.byref(h)))
        self._h = h

    def num_species(self) -> int:
        return int(self._L.bh_range_filter_num_species(self._h))

    def labels(self) -> List[str]:
        return [self._L.bh_range_filter_label(self._h, i).decode("utf-8") for i in range(self.num_species())]

    def _run(self, call):
        n = self.num_species()
        scores = np.zeros(n, np.float32)
        idx = np.zeros(n, np.uint32)
        kept = C.c_size_t()
        check(call(scores.ctypes.data, n, idx.ctypes.data, C.byref(kept)))
        return scores, idx[: kept.value].copy()

    def predict(self, latitude: float, longitude: float, month: int, day: int):
        """-> (scores of every species, indices of those at or above the threshold)"""
        return self._run(lambda s, n, i, k: self._L.bh_range_filter_predict(self._h, latitude, longitude, month, day, s, n, i, k))

    def predict_week(self, latitude: float, longitude: float, week: float):
        return self._run(lambda s, n, i, k: self._L.bh_range_filter_predict_week(self._h, latitude, longitude, week, s, n, i, k))

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.bh_range_filter_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""`MultiClassifier`: several shards of one segment list in one process (include/birda_hip.h `bh_multi_*`)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import BhMultiConfig, BhResult, BirdaHipError
from .classifier import Prediction, PredictionResult

GATHER = {"auto": 0, "host": 1, "rccl": 2}
PRECISION_FLAGS = {"auto": 0, "f16x3": 1, "f16": 2, "f32": 3}   # BH_FLAG_* (include/birda_hip.h)


def _mcheck(rc: int):
    if rc != 0:
        raise BirdaHipError(rc, _lib.load().bh_multi_last_error().decode("utf-8", "replace"))


class MultiClassifier:
    def __init__(self, model_path: str, labels_path: Optional[str] = None, devices: Optional[Sequence[int]] = None,
                 top_k: int = 5, min_confidence: float = 0.1, precision: str = "auto", max_batch: int = 0,
                 gather: str = "auto"):
        self._L = _lib.load()
        devs = (C.c_int32 * max(1, len(devices or [])))(*(devices or []))
        cfg = BhMultiConfig(model_path.encode(), labels_path.encode() if labels_path else None, top_k, min_confidence,
                            PRECISION_FLAGS[precision], devs, len(devices or []), max_batch, GATHER[gather])
        h = C.c_void_p()
        _mcheck(self._L.bh_multi_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.top_k = top_k
        info = _lib.BhModelInfo()
        self._L.bh_classifier_info(self._L.bh_multi_classifier(self._h, 0), C.byref(info))
        self.info = info

    @property
    def n_shards(self) -> int:
        return int(self._L.bh_multi_shards(self._h))

    def shard_devices(self) -> List[int]:
        return [int(self._L.bh_multi_shard_device(self._h, g)) for g in range(self.n_shards)]

    def shard_classifier(self, shard: int):
        """Shard `shard`'s classifier replica (borrowed: labels, info, kernel names, filters)."""
        from .classifier import BirdClassifier
        return BirdClassifier.borrow(self._L.bh_multi_classifier(self._h, shard), self.top_k,
                                     device=int(self._L.bh_multi_shard_device(self._h, shard)))

    def shard_context(self, shard: int):
        """Shard `shard`'s batch context (borrowed: profiling events, stream)."""
        from .classifier import BatchInferenceContext
        return BatchInferenceContext.borrow(self.shard_classifier(shard), self._L.bh_multi_context(self._h, shard))

    def gather_backend(self) -> str:
        return self._L.bh_multi_gather_backend(self._h).decode()

    def label(self, index: int) -> Optional[str]:
        raw = self._L.bh_classifier_label(self._L.bh_multi_classifier(self._h, 0), index)
        return raw.decode("utf-8") if raw else None

    def _results(self, res, n) -> List[PredictionResult]:
        out = []
        for i in range(n):
            r = res[i]
            out.append(PredictionResult([Prediction(self.label(r.index[k]) or str(r.index[k]), float(r.confidence[k]), int(r.index[k]))
                                         for k in range(r.n_pred)]))
        return out

    def predict_batch_contig(self, segments: np.ndarray) -> List[PredictionResult]:
        x = np.ascontiguousarray(segments, np.float32)
        assert x.ndim == 2 and x.shape[1] == self.info.sample_count
        res = (BhResult * max(1, x.shape[0]))()
        _mcheck(self._L.bh_multi_predict_batch_contig(self._h, x.ctypes.data, x.shape[0], res))
        return self._results(res, x.shape[0])

    def predict_batch_source_rate(self, segments: Sequence[np.ndarray], source_rates: Sequence[int]):
        """Mixed-rate list (BASELINE config 5): returns (results, cut points of the source-sample-balanced partition)."""
        keep = [np.ascontiguousarray(s, np.float32) for s in segments]
        n = len(keep)
        ptrs = (C.c_void_p * max(1, n))(*[k.ctypes.data for k in keep])
        rates = np.asarray(source_rates, np.uint32)
        lens = np.asarray([k.size for k in keep], np.uint64)      # size_t
        bounds = np.zeros(self.n_shards + 1, np.uint64)
        res = (BhResult * max(1, n))()
        _mcheck(self._L.bh_multi_predict_batch_source_rate(self._h, ptrs, rates.ctypes.data, lens.ctypes.data, n, res, bounds.ctypes.data))
        return self._results(res, n), [int(b) for b in bounds]

    def forward_device(self, device_ptrs: Sequence[int], n_per_shard: Sequence[int]) -> List[PredictionResult]:
        G = self.n_shards
        assert len(device_ptrs) == G and len(n_per_shard) == G
        ptrs = (C.c_void_p * G)(*device_ptrs)
        counts = np.asarray(n_per_shard, np.uint64)
        total = int(counts.sum())
        res = (BhResult * max(1, total))()
        _mcheck(self._L.bh_multi_forward_device(self._h, ptrs, counts.ctypes.data, res))
        return self._results(res, total)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.bh_multi_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

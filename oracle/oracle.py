"""ctypes binding of the CPU oracle (oracle/birda_oracle.c).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from birda_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libbirda_oracle.so")
_lib = None

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "birda_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    L.bo_model_load.restype = C.c_void_p
    L.bo_model_load.argtypes = [C.c_char_p]
    L.bo_model_free.argtypes = [C.c_void_p]
    for name in ("bo_sample_rate", "bo_sample_count", "bo_n_classes", "bo_embedding_dim", "bo_n_layers"):
        getattr(L, name).restype = C.c_uint32
        getattr(L, name).argtypes = [C.c_void_p]
    L.bo_segment_duration.restype = C.c_float
    L.bo_segment_duration.argtypes = [C.c_void_p]
    L.bo_tensor_floats.restype = C.c_uint64
    L.bo_tensor_floats.argtypes = [C.c_void_p, C.c_uint32]
    L.bo_frontend.argtypes = [C.c_void_p, f32p, f32p]
    L.bo_forward.restype = C.c_int
    L.bo_forward.argtypes = [C.c_void_p, f32p, C.c_int, f32p, C.c_void_p, C.c_int, C.c_void_p]
    L.bo_topk.restype = C.c_int
    L.bo_topk.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    L.bo_fft.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int]
    L.bo_scientific_name_len.restype = C.c_size_t
    L.bo_scientific_name_len.argtypes = [C.c_char_p]
    L.bo_project_scores.restype = C.c_size_t
    L.bo_project_scores.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    L.bo_filter_predictions.restype = C.c_int
    L.bo_filter_predictions.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.bo_species_retain.restype = C.c_int
    L.bo_species_retain.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bo_pcm16_to_mono.argtypes = [C.c_void_p, C.c_size_t, C.c_int, f32p]
    L.bo_pcm32_to_mono.argtypes = [C.c_void_p, C.c_size_t, C.c_int, f32p]
    L.bo_f32_to_mono.argtypes = [f32p, C.c_size_t, C.c_int, f32p]
    L.bo_segmenter_new.restype = C.c_void_p
    L.bo_segmenter_new.argtypes = [f32p, C.c_size_t, C.c_size_t]
    L.bo_segmenter_free.argtypes = [C.c_void_p]
    L.bo_segmenter_next.restype = C.c_int
    L.bo_segmenter_next.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, f32p, C.POINTER(C.c_size_t)]
    L.bo_source_samples.restype = C.c_size_t
    L.bo_source_samples.argtypes = [C.c_size_t, C.c_uint32, C.c_uint32]
    L.bo_duration_to_samples.restype = C.c_size_t
    L.bo_duration_to_samples.argtypes = [C.c_float, C.c_uint32]
    L.bo_estimate_segment_count.restype = C.c_int64
    L.bo_estimate_segment_count.argtypes = [C.c_int, C.c_double, C.c_float, C.c_float]
    L.bo_effective_batch_size.restype = C.c_size_t
    L.bo_effective_batch_size.argtypes = [C.c_size_t, C.c_int64]
    L.bo_chunk_times.argtypes = [C.c_size_t, C.c_uint32, C.c_size_t, C.c_uint32, C.POINTER(C.c_float),
                                 C.POINTER(C.c_float)]
    L.bo_chunk_audio_count.restype = C.c_size_t
    L.bo_chunk_audio_count.argtypes = [C.c_size_t, C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_size_t]
    L.bo_resampler_sizes.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.bo_resample_max_len.restype = C.c_size_t
    L.bo_resample_max_len.argtypes = [C.c_size_t, C.c_uint32, C.c_uint32]
    L.bo_resample.restype = C.c_size_t
    L.bo_resample.argtypes = [f32p, C.c_size_t, C.c_uint32, C.c_uint32, f32p]
    L.bo_sort_detections.argtypes = [C.c_void_p, C.c_size_t]
    L.bo_csv_row.restype = C.c_size_t
    L.bo_csv_row.argtypes = [C.c_char_p, C.c_float, C.c_float, C.c_float, C.c_char_p, C.c_char_p]
    L.bo_csv_header.restype = C.c_size_t
    L.bo_csv_header.argtypes = [C.c_int, C.c_char_p]
    L.bo_process_stream.restype = C.c_size_t
    L.bo_process_stream.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), f32p, C.c_size_t, C.c_uint32, C.c_float,
                                    C.c_float, C.c_int, C.c_size_t, C.c_int, C.c_char_p, C.c_char_p, C.c_size_t,
                                    C.c_void_p, C.c_size_t, C.c_void_p]
    _lib = L
    return L


class Detection(C.Structure):
    _fields_ = [("start", C.c_float), ("end", C.c_float), ("conf", C.c_float), ("label", C.c_int)]


class ProcessStats(C.Structure):
    _fields_ = [("n_segments", C.c_size_t), ("n_detections", C.c_size_t), ("effective_batch", C.c_size_t),
                ("n_batches", C.c_size_t), ("n_padded_rows", C.c_size_t)]


class OracleModel:
    """CPU forward of a BHM1 model (restates birdnet_onnx::Classifier for this path)."""

    def __init__(self, path: str):
        self.L = lib()
        self.h = self.L.bo_model_load(path.encode())
        if not self.h:
            raise RuntimeError(f"oracle: cannot load {path}")
        self.sample_rate = self.L.bo_sample_rate(self.h)
        self.sample_count = self.L.bo_sample_count(self.h)
        self.segment_duration = self.L.bo_segment_duration(self.h)
        self.n_classes = self.L.bo_n_classes(self.h)
        self.embedding_dim = self.L.bo_embedding_dim(self.h)
        self.n_layers = self.L.bo_n_layers(self.h)

    def close(self):
        if self.h:
            self.L.bo_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tensor_floats(self, t: int) -> int:
        return int(self.L.bo_tensor_floats(self.h, t))

    def frontend(self, seg: np.ndarray) -> np.ndarray:
        seg = np.ascontiguousarray(seg, np.float32)
        out = np.empty(self.tensor_floats(0), np.float32)
        self.L.bo_frontend(self.h, seg, out)
        return out

    def forward(self, segs: np.ndarray, dump_tensor: int = -1, want_embeddings: bool = False):
        segs = np.ascontiguousarray(segs, np.float32).reshape(-1, self.sample_count)
        n = segs.shape[0]
        logits = np.empty((n, self.n_classes), np.float32)
        emb = np.empty((n, self.embedding_dim), np.float32) if want_embeddings else None
        dump = np.empty((n, self.tensor_floats(dump_tensor)), np.float32) if dump_tensor >= 0 else None
        self.L.bo_forward(self.h, segs, n, logits, emb.ctypes.data if emb is not None else None,
                          dump_tensor, dump.ctypes.data if dump is not None else None)
        res = [logits]
        if want_embeddings:
            res.append(emb)
        if dump_tensor >= 0:
            res.append(dump)
        return res[0] if len(res) == 1 else tuple(res)

    def process_stream(self, labels: List[str], samples: np.ndarray, source_rate: int, overlap: float = 0.0,
                       min_conf: float = 0.1, top_k: int = 5, batch_size: int = 8, bom: bool = True,
                       file_path: str = "audio.wav", want_logits: bool = False):
        samples = np.ascontiguousarray(samples, np.float32)
        arr = (C.c_char_p * len(labels))(*[s.encode("utf-8") for s in labels])
        cap = 1 << 22
        buf = C.create_string_buffer(cap)
        stats = ProcessStats()
        max_rows = int(len(samples) / source_rate / max(self.segment_duration - overlap, 1e-3)) + 8
        logits = np.zeros((max_rows, self.n_classes), np.float32) if want_logits else None
        n = self.L.bo_process_stream(self.h, arr, samples, len(samples), source_rate, overlap, min_conf, top_k,
                                     batch_size, int(bom), file_path.encode(), buf, cap,
                                     logits.ctypes.data if want_logits else None, max_rows, C.byref(stats))
        if n == C.c_size_t(-1).value:
            raise RuntimeError("oracle: process_stream failed")
        out = buf.raw[:n]
        if want_logits:
            return out, stats, logits[:stats.n_segments]
        return out, stats


def topk(logits: np.ndarray, out_act: int, top_k: int, min_conf: float) -> Tuple[np.ndarray, np.ndarray]:
    L = lib()
    logits = np.ascontiguousarray(logits, np.float32)
    idx = np.zeros(top_k, np.int32)
    conf = np.zeros(top_k, np.float32)
    k = L.bo_topk(logits, logits.size, out_act, top_k, min_conf, idx.ctypes.data, conf.ctypes.data)
    return idx[:k].copy(), conf[:k].copy()


def scientific_name(label: str) -> str:
    raw = label.encode("utf-8")
    return raw[:lib().bo_scientific_name_len(raw)].decode("utf-8")


def _cstrs(items):
    raw = [s.encode("utf-8") for s in items]
    return (C.c_char_p * max(1, len(raw)))(*raw), raw


def project_scores(geo_labels, reported, cls_labels) -> Tuple[np.ndarray, int]:
    """reported: [(geomodel species label, score)].  Returns (per-class scores with NaN = no geomodel entry, mapped count)."""
    L = lib()
    g, _g = _cstrs(geo_labels)
    sp, _s = _cstrs([r[0] for r in reported])
    c, _c = _cstrs(cls_labels)
    vals = np.asarray([r[1] for r in reported] or [0.0], np.float32)
    out = np.zeros(max(1, len(cls_labels)), np.float32)
    mapped = L.bo_project_scores(g, len(geo_labels), sp, vals.ctypes.data, len(reported), c, len(cls_labels), out.ctypes.data)
    return out[:len(cls_labels)].copy(), int(mapped)


def filter_predictions(idx, conf, scores, threshold: float, keep_unmatched: bool, rerank: bool):
    L = lib()
    idx = np.ascontiguousarray(idx, np.int32); conf = np.ascontiguousarray(conf, np.float32)
    scores = np.ascontiguousarray(scores, np.float32)
    oi = np.zeros(max(1, idx.size), np.int32); oc = np.zeros(max(1, idx.size), np.float32)
    k = L.bo_filter_predictions(idx.ctypes.data, conf.ctypes.data, idx.size, scores.ctypes.data, threshold,
                                int(keep_unmatched), int(rerank), oi.ctypes.data, oc.ctypes.data)
    return oi[:k].copy(), oc[:k].copy()


def species_retain(idx, conf, keep):
    L = lib()
    idx = np.ascontiguousarray(idx, np.int32); conf = np.ascontiguousarray(conf, np.float32)
    keep = np.ascontiguousarray(keep, np.uint8)
    oi = np.zeros(max(1, idx.size), np.int32); oc = np.zeros(max(1, idx.size), np.float32)
    k = L.bo_species_retain(idx.ctypes.data, conf.ctypes.data, idx.size, keep.ctypes.data, oi.ctypes.data, oc.ctypes.data)
    return oi[:k].copy(), oc[:k].copy()


def fft(x: np.ndarray, sign: int = -1) -> np.ndarray:
    L = lib()
    x = np.asarray(x, np.complex128)
    re, im = np.ascontiguousarray(x.real), np.ascontiguousarray(x.imag)
    ore, oim = np.empty_like(re), np.empty_like(im)
    L.bo_fft(re.ctypes.data, im.ctypes.data, ore.ctypes.data, oim.ctypes.data, x.size, sign)
    return ore + 1j * oim


def resample(x: np.ndarray, from_rate: int, to_rate: int) -> np.ndarray:
    L = lib()
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(max(int(L.bo_resample_max_len(x.size, from_rate, to_rate)), 1), np.float32)
    n = L.bo_resample(x, x.size, from_rate, to_rate, out)
    return out[:n].copy()


def resampler_sizes(from_rate: int, to_rate: int, chunk: int = 1024) -> Tuple[int, int]:
    L = lib()
    a, b = C.c_int(), C.c_int()
    L.bo_resampler_sizes(from_rate, to_rate, chunk, C.byref(a), C.byref(b))
    return a.value, b.value


def segment_stream(samples: np.ndarray, seg: int, ovl: int, packet: int = 1152):
    """[(segment, start_sample)] exactly as StreamingDecoder::next_segment would yield."""
    L = lib()
    samples = np.ascontiguousarray(samples, np.float32)
    keep = samples if samples.size else np.zeros(1, np.float32)
    h = L.bo_segmenter_new(keep, samples.size, packet)
    out = []
    try:
        while True:
            buf = np.empty(max(seg, 1), np.float32)
            start = C.c_size_t()
            rc = L.bo_segmenter_next(h, seg, ovl, buf, C.byref(start))
            if rc < 0:
                raise ValueError("overlap_samples must be less than segment_samples")
            if rc == 0:
                break
            out.append((buf, start.value))
    finally:
        L.bo_segmenter_free(h)
    return out


# --------------------------------------------------------------------------------------------------------
# Two-stage inference and BSG post-processing, numpy restatements.  TEST INFRASTRUCTURE like the rest of
# this package.  [EXT] Both live in birdnet-onnx (CustomClassifier, BsgPostProcessor), not in the reference tree: parity
# unpinned; the forms below are the ones include/birda_hip.h states.
# --------------------------------------------------------------------------------------------------------
def custom_classifier_forward(model, embeddings: np.ndarray) -> np.ndarray:
    """birdnet_onnx::CustomClassifier::predict_batch up to the logits (reference call site src/pipeline/processor.rs:341):
    dense layers x W + b with the layer activation (0 none, 1 ReLU), float32 like ONNX Runtime's Gemm."""
    x = np.asarray(embeddings, np.float32)
    for L in model.layers:
        x = (x.astype(np.float64) @ L.w.astype(np.float64) + L.b.astype(np.float64)).astype(np.float32)
        if L.act == 1:
            x = np.maximum(x, 0.0)
        elif L.act != 0:
            raise ValueError("activation not restated")
    return x


def bsg_postprocess(index: np.ndarray, confidence: np.ndarray, intercept: np.ndarray, slope: np.ndarray,
                    prior: Optional[np.ndarray] = None):
    """BirdClassifier::apply_bsg_postprocessing (reference src/inference/classifier.rs:508-545) on one segment's kept
    predictions: conf' = sigmoid(intercept[c] + slope[c] * logit(conf)) (* prior[c]), stable re-sort descending."""
    idx = [int(i) for i in index if i >= 0]
    out = []
    for c, p in zip(idx, confidence):
        p = min(max(float(p), 1e-7), 1.0 - 1e-7)
        lg = np.log(p / (1.0 - p))
        q = 1.0 / (1.0 + np.exp(-(float(intercept[c]) + float(slope[c]) * lg)))
        if prior is not None:
            q *= float(prior[c])
        out.append((c, q))
    order = sorted(range(len(out)), key=lambda i: -out[i][1])     # sorted() is stable: ties keep their order
    return [out[i][0] for i in order], [out[i][1] for i in order]

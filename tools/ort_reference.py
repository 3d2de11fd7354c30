"""The opportunistic TRUE-reference leg: ONNX Runtime's CPU execution provider running
the published model, bound through the ORT C API with ctypes.

birda reaches ONNX Runtime the same way: `ort` is built with `load-dynamic` and `src/inference/runtime.rs:13-21,89-96`
dlopens `libonnxruntime` from `ORT_DYLIB_PATH` (`src/constants.rs:547`), `LD_LIBRARY_PATH` or the usual directories, then
birdnet-onnx builds a session and calls `Session::run` (reference call sites `src/inference/classifier.rs:269-283,478-488`).

Nothing here can run in the build container or on the GPU box as they are (no libonnxruntime, no model file, no network), so
this module is UNVERIFIED against a live runtime: it is written against the ORT C API's stable, append-only function table
(onnxruntime_c_api.h, `struct OrtApi`, version 1 onward) and refuses to guess when anything looks wrong.  It lives under
tools/ (not in the product package): only bench.py's cpu_baseline leg and a skipped-unless-present test call it.

    run(ort_path, onnx_path, sample_count) -> {"available": True, "value": segments/s, "logits": [n, classes] ...}
"""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import List, Optional

import numpy as np

ORT_API_VERSION = 1          # the v1 table holds every entry used here; ORT 1.24 still serves it
# indices into struct OrtApi (onnxruntime_c_api.h; the table is append-only)
IDX = {"GetErrorMessage": 2, "CreateEnv": 3, "CreateSession": 7, "Run": 9, "CreateSessionOptions": 10,
       "SetSessionGraphOptimizationLevel": 23, "SetIntraOpNumThreads": 24, "SessionGetInputCount": 30,
       "SessionGetOutputCount": 31, "SessionGetInputName": 36, "SessionGetOutputName": 37,
       "CreateTensorWithDataAsOrtValue": 49, "GetTensorMutableData": 51, "GetDimensionsCount": 61, "GetDimensions": 62,
       "GetTensorTypeAndShape": 65, "CreateCpuMemoryInfo": 69, "AllocatorFree": 76, "GetAllocatorWithDefaultOptions": 78,
       "ReleaseEnv": 92, "ReleaseStatus": 93, "ReleaseMemoryInfo": 94, "ReleaseSession": 95, "ReleaseValue": 96,
       "ReleaseTensorTypeAndShapeInfo": 99, "ReleaseSessionOptions": 100}
ONNX_TENSOR_ELEMENT_DATA_TYPE_FLOAT = 1
ORT_LOGGING_LEVEL_WARNING = 2
ORT_ENABLE_ALL = 99


class OrtError(RuntimeError):
    pass


class _Api:
    def __init__(self, lib_path: str):
        self.lib = C.CDLL(lib_path)
        self.lib.OrtGetApiBase.restype = C.c_void_p
        base = self.lib.OrtGetApiBase()
        if not base:
            raise OrtError("OrtGetApiBase returned NULL")
        # struct OrtApiBase { const OrtApi* (*GetApi)(uint32_t); const char* (*GetVersionString)(void); }
        fns = C.cast(base, C.POINTER(C.c_void_p))
        get_api = C.CFUNCTYPE(C.c_void_p, C.c_uint32)(fns[0])
        self.version = C.CFUNCTYPE(C.c_char_p)(fns[1])().decode()
        table = get_api(ORT_API_VERSION)
        if not table:
            raise OrtError(f"ORT {self.version} does not serve API version {ORT_API_VERSION}")
        self.table = C.cast(table, C.POINTER(C.c_void_p))

    def fn(self, name: str, restype, *argtypes):
        return C.CFUNCTYPE(restype, *argtypes)(self.table[IDX[name]])

    def check(self, status):
        if status:
            msg = self.fn("GetErrorMessage", C.c_char_p, C.c_void_p)(status).decode("utf-8", "replace")
            self.fn("ReleaseStatus", None, C.c_void_p)(status)
            raise OrtError(msg)


class OrtSession:
    """One CPU session: intra-op threads = all host cores, full graph optimisation."""

    def __init__(self, lib_path: str, onnx_path: str, threads: Optional[int] = None):
        a = self.api = _Api(lib_path)
        vp = C.c_void_p
        self.env, self.opts, self.sess, self.mem = vp(), vp(), vp(), vp()
        a.check(a.fn("CreateEnv", vp, C.c_int, C.c_char_p, C.POINTER(vp))(ORT_LOGGING_LEVEL_WARNING, b"birda_hip_bench", C.byref(self.env)))
        a.check(a.fn("CreateSessionOptions", vp, C.POINTER(vp))(C.byref(self.opts)))
        a.check(a.fn("SetIntraOpNumThreads", vp, vp, C.c_int)(self.opts, threads or (os.cpu_count() or 1)))
        a.check(a.fn("SetSessionGraphOptimizationLevel", vp, vp, C.c_int)(self.opts, ORT_ENABLE_ALL))
        a.check(a.fn("CreateSession", vp, vp, C.c_char_p, vp, C.POINTER(vp))(self.env, onnx_path.encode(), self.opts, C.byref(self.sess)))
        a.check(a.fn("CreateCpuMemoryInfo", vp, C.c_int, C.c_int, C.POINTER(vp))(0, 0, C.byref(self.mem)))   # OrtArenaAllocator, OrtMemTypeDefault
        alloc = vp()
        a.check(a.fn("GetAllocatorWithDefaultOptions", vp, C.POINTER(vp))(C.byref(alloc)))
        self.inputs, self.outputs = self._names("SessionGetInputCount", "SessionGetInputName", alloc), self._names("SessionGetOutputCount", "SessionGetOutputName", alloc)
        if len(self.inputs) != 1:
            raise OrtError(f"expected one waveform input, the graph has {self.inputs}")

    def _names(self, count_fn: str, name_fn: str, alloc) -> List[bytes]:
        a, vp = self.api, C.c_void_p
        n = C.c_size_t()
        a.check(a.fn(count_fn, vp, vp, C.POINTER(C.c_size_t))(self.sess, C.byref(n)))
        out = []
        for i in range(n.value):
            p = C.c_char_p()
            a.check(a.fn(name_fn, vp, vp, C.c_size_t, vp, C.POINTER(C.c_char_p))(self.sess, i, alloc, C.byref(p)))
            out.append(bytes(p.value))
            a.check(a.fn("AllocatorFree", vp, vp, C.c_char_p)(alloc, p))
        return out

    def run(self, batch: np.ndarray) -> List[np.ndarray]:
        """batch [n, sample_count] f32 -> every graph output as float32 arrays (logits first in the published files)."""
        a, vp = self.api, C.c_void_p
        x = np.ascontiguousarray(batch, np.float32)
        shape = (C.c_int64 * 2)(x.shape[0], x.shape[1])
        val = vp()
        a.check(a.fn("CreateTensorWithDataAsOrtValue", vp, vp, vp, C.c_size_t, C.POINTER(C.c_int64), C.c_size_t, C.c_int, C.POINTER(vp))(
            self.mem, x.ctypes.data, x.nbytes, shape, 2, ONNX_TENSOR_ELEMENT_DATA_TYPE_FLOAT, C.byref(val)))
        in_names = (C.c_char_p * 1)(self.inputs[0])
        out_names = (C.c_char_p * len(self.outputs))(*self.outputs)
        outs = (vp * len(self.outputs))()
        in_vals = (vp * 1)(val)
        try:
            a.check(a.fn("Run", vp, vp, vp, C.POINTER(C.c_char_p), C.POINTER(vp), C.c_size_t, C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(vp))(
                self.sess, None, in_names, in_vals, 1, out_names, len(self.outputs), outs))
            res = []
            for o in outs:
                info = vp()
                a.check(a.fn("GetTensorTypeAndShape", vp, vp, C.POINTER(vp))(o, C.byref(info)))
                nd = C.c_size_t()
                a.check(a.fn("GetDimensionsCount", vp, vp, C.POINTER(C.c_size_t))(info, C.byref(nd)))
                dims = (C.c_int64 * nd.value)()
                a.check(a.fn("GetDimensions", vp, vp, C.POINTER(C.c_int64), C.c_size_t)(info, dims, nd.value))
                a.fn("ReleaseTensorTypeAndShapeInfo", None, vp)(info)
                data = vp()
                a.check(a.fn("GetTensorMutableData", vp, vp, C.POINTER(vp))(o, C.byref(data)))
                n = int(np.prod(list(dims))) if nd.value else 1
                res.append(np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_float)), (n,)).reshape(list(dims)).copy())
            return res
        finally:
            rel = a.fn("ReleaseValue", None, vp)
            for o in outs:
                if o:
                    rel(o)
            rel(val)

    def close(self):
        a, vp = self.api, C.c_void_p
        for name, h in (("ReleaseSession", self.sess), ("ReleaseSessionOptions", self.opts), ("ReleaseMemoryInfo", self.mem), ("ReleaseEnv", self.env)):
            if h:
                a.fn(name, None, vp)(h)
        self.sess = self.opts = self.mem = self.env = C.c_void_p()


def run(ort_path: str, onnx_path: str, sample_count: int, sample_rate: int = 48000, n_segments: int = 256, batch: int = 8,
        repeats: int = 3) -> dict:
    """The reference CPU path as birda configures it: batch 8 (src/constants.rs:36), one warm-up batch (ensure_warm,
    src/inference/classifier.rs:414-427), >= 3 timed repeats; logits of the first 16 SURVEY-8d segments for max |dlogit|."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from birda_amd import synth
    s = OrtSession(ort_path, onnx_path)
    try:
        base = synth.synth_segments(16, sample_count, sample_rate)
        first = np.concatenate([s.run(base[i:i + batch])[0] for i in range(0, 16, batch)])
        segs = np.tile(base, (n_segments // 16 + 1, 1))[:n_segments]
        s.run(segs[:batch])
        times = []
        for _ in range(repeats):
            t = time.perf_counter()
            for i in range(0, n_segments, batch):
                s.run(segs[i:i + batch])
            times.append(time.perf_counter() - t)
        dt = sorted(times)[len(times) // 2]
        return {"available": True, "kind": "reference", "value": round(n_segments / dt, 2), "unit": "segments/s", "cores": os.cpu_count(),
                "runtime": f"ONNX Runtime {s.api.version} CPU EP, intra-op threads = all cores, batch {batch}",
                "model": os.path.basename(onnx_path), "outputs": [o.decode() for o in s.outputs],
                "sample": f"{n_segments} synthetic segments x {repeats} repeats (median), {dt:.1f} s", "logits_first16": first}
    finally:
        s.close()

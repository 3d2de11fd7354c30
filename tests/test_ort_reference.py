"""The opportunistic true-reference leg: ONNX Runtime's CPU provider on the published model, bound through the
ORT C API (tools/ort_reference.py).  Neither a libonnxruntime nor a model file exists in the build container or on the GPU
box, so the live test skips there; what always runs is a consistency check of the binding's function-table indices."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import ort_reference  # noqa: E402


def test_function_table_indices_are_consistent():
    """struct OrtApi is append-only: every entry the binding uses has a distinct slot, in the header's relative order."""
    idx = ort_reference.IDX
    assert len(set(idx.values())) == len(idx)
    order = ["GetErrorMessage", "CreateEnv", "CreateSession", "Run", "CreateSessionOptions", "SetSessionGraphOptimizationLevel",
             "SetIntraOpNumThreads", "SessionGetInputCount", "SessionGetOutputCount", "SessionGetInputName", "SessionGetOutputName",
             "CreateTensorWithDataAsOrtValue", "GetTensorMutableData", "GetDimensionsCount", "GetDimensions", "GetTensorTypeAndShape",
             "CreateCpuMemoryInfo", "AllocatorFree", "GetAllocatorWithDefaultOptions", "ReleaseEnv", "ReleaseStatus",
             "ReleaseMemoryInfo", "ReleaseSession", "ReleaseValue", "ReleaseTensorTypeAndShapeInfo", "ReleaseSessionOptions"]
    assert set(order) == set(idx)
    slots = [idx[n] for n in order]
    assert slots == sorted(slots)


def test_missing_runtime_is_reported_not_guessed(tmp_path):
    with pytest.raises(OSError):
        ort_reference.OrtSession(str(tmp_path / "no_such_libonnxruntime.so"), str(tmp_path / "no.onnx"))


@pytest.mark.skipif(not (os.path.isfile(os.environ.get("ORT_DYLIB_PATH", "")) and os.path.isfile(os.environ.get("BIRDA_REFERENCE_ONNX", ""))),
                    reason="needs ORT_DYLIB_PATH (reference src/constants.rs:547) and BIRDA_REFERENCE_ONNX: neither exists offline")
def test_reference_runtime_runs_the_published_model():
    r = ort_reference.run(os.environ["ORT_DYLIB_PATH"], os.environ["BIRDA_REFERENCE_ONNX"], 144000, n_segments=16, repeats=1)
    assert r["available"] and r["value"] > 0
    logits = r["logits_first16"]
    assert logits.shape[0] == 16 and np.isfinite(logits).all()

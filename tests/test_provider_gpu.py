"""SURVEY 8 row a12 on the hardware: the execution-provider arm of this backend WITH a device present.

Reference: `select_execution_provider` (src/inference/classifier.rs:662-691: the auto / gpu priority list), the CPU fall-back of
that list (:742-754), `configure_explicit_provider` (:924-984: an explicit provider is an error when unavailable) and the status
the run reports, `ExecutionProviderStatus{requested, actual, fallback_reason}` (:23-30).  The no-device arms are asserted on the
CPU box (tests/test_output_writers.py::test_default_batch_size_reference_cases_and_provider_arm)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


def _select(L, requested: bytes, ordinal: int = -1):
    from birda_amd import _lib
    st = _lib.BhProviderStatus()
    rc = L.bh_select_provider(requested, ordinal, C.byref(st))
    return rc, st


def _device_facts_ok(st, n_dev):
    assert st.actual == b"HIP" and st.fallback_reason == b""          # fallback_reason: None
    assert st.device_count == n_dev >= 1 and 0 <= st.device < n_dev
    assert st.arch.startswith(b"gfx950"), st.arch                      # gcnArchName, e.g. gfx950:sramecc+:xnack-
    assert st.compute_units == 256
    assert 280e9 < st.hbm_bytes <= 288 * 2 ** 30                        # 288 GiB of HBM3E (309.2e9 bytes), less whatever is reserved
    assert len(st.device_name) > 0


def test_provider_arm_with_a_device_present():
    from birda_amd import _lib
    L = _lib.load()
    n_dev = L.bh_device_count()
    assert n_dev >= 1
    assert L.bh_backend_name() == b"HIP (gfx950)"
    for req in (b"auto", b"gpu", b"hip", b"rocm", b"AUTO", b"Gpu", b"ROCm"):   # InferenceDevice parses case-insensitively
        rc, st = _select(L, req)
        assert rc == 0, (req, L.bh_last_error())
        assert st.requested == req.lower()
        _device_facts_ok(st, n_dev)
        assert st.device == 0                                          # ordinal < 0 picks device 0
    # an explicit ordinal is honoured; one past the last device is "provider unavailable" for the explicit spellings
    # (configure_explicit_provider: an error) and a CPU fall-back with a reason for auto / gpu (:742-754)
    rc, st = _select(L, b"hip", n_dev - 1)
    assert rc == 0 and st.device == n_dev - 1
    rc, st = _select(L, b"rocm", n_dev)
    assert rc == -3 and b"out of range" in L.bh_last_error()
    rc, st = _select(L, b"auto", n_dev)
    assert rc == 0 and (st.actual, st.fallback_reason, st.device) == (b"CPU", b"No GPU providers available", -1)
    # the CPU arm never touches the device, whatever is installed (:696-705)
    rc, st = _select(L, b"cpu")
    assert rc == 0 and (st.requested, st.actual, st.fallback_reason, st.device) == (b"cpu", b"CPU", b"", -1)
    # other providers' names are not this backend's arm
    for other in (b"cuda", b"tensorrt", b"directml", b"coreml", b"openvino", b""):
        rc, _ = _select(L, other)
        assert rc == -1, other
    assert L.bh_select_provider(None, -1, C.byref(_lib.BhProviderStatus())) == -1


def test_a_built_classifier_reports_the_same_status(model_dir):
    from birda_amd import _lib
    from birda_amd.classifier import BirdClassifier
    L = _lib.load()
    path, labels, m, _ = model_dir["mini"]
    clf = BirdClassifier(path, labels, top_k=5, min_confidence=0.1, device=0)
    try:
        got = clf.provider_status()
        rc, want = _select(L, b"hip", 0)
        assert rc == 0
        _device_facts_ok(got, L.bh_device_count())
        for f in ("actual", "fallback_reason", "device", "device_count", "device_name", "arch", "compute_units", "hbm_bytes"):
            assert getattr(got, f) == getattr(want, f), f
        assert got.requested == b"hip"
        # determine_default_batch_size's arm for the provider the classifier runs under (lib.rs:256-288)
        assert clf.default_batch_size() == L.bh_default_batch_size(0, got.actual) == 512
    finally:
        clf.close()
    assert L.bh_classifier_provider_status(None, C.byref(_lib.BhProviderStatus())) == -1

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The tests use PyTorch for device buffers next to libbirda_hip.so.  The PyTorch wheel bundles its own copy of the HIP /
# HSA runtime; whichever copy is mapped first serves both, and PyTorch only finds the GPU when it is its own
# ("No HIP GPUs are available" otherwise).  Importing torch before the library is loaded fixes the order for any
# selection of test files (a full run got it right by accident: test_sharding_gloo.py imports torch at collection).
try:
    import torch  # noqa: F401
except ImportError:   # CPU-only environments without torch: the tests that need it skip or fail on their own
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def model_dir(tmp_path_factory):
    """Seeded synthetic models + labels written once per session."""
    from birda_amd import modelfile as mf, synth
    d = tmp_path_factory.mktemp("models")
    out = {}
    for kind in ("mini", "birdnet_v24_tiny", "mini_b0", "mini_hg", "mini_se"):
        m = synth.build_model(kind)
        p = str(d / f"{kind}.bhm")
        mf.write_model(p, m)
        lp = str(d / f"{kind}.labels.txt")
        labels = synth.write_labels(lp, m.n_classes)
        out[kind] = (p, lp, m, labels)
    return out


@pytest.fixture(scope="session")
def full_model(tmp_path_factory):
    from birda_amd import modelfile as mf, synth
    d = tmp_path_factory.mktemp("full")
    m = synth.build_model("birdnet_v24")
    p = str(d / "birdnet_v24.bhm")
    mf.write_model(p, m)
    lp = str(d / "labels.txt")
    labels = synth.write_labels(lp, m.n_classes)
    return p, lp, m, labels

"""birda_amd: MI355X-native hot path for tphakala/birda (segments -> logits).

The product is `libbirda_hip.so` (C ABI in `include/birda_hip.h`, HIP kernels in
`birda_amd/csrc/`).  This package holds the host-side Python mirror of the
reference's `BirdClassifier` surface (ctypes over the C ABI), the BHM1 model
container and the seeded synthetic model/audio generators used by tests and
`bench.py`.
"""
__version__ = "0.1.0"

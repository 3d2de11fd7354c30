"""Device resampler throughput: n raw segments at a source rate -> the model's rate (bh_resample_device), ms per launch.
    python tools/gpu_resample_rate.py [n_segments] [precision]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prec = sys.argv[2] if len(sys.argv) > 2 else "auto"
m = synth.build_model("mini"); mf.write_model("/tmp/_rs.bhm", m)
clf = BirdClassifier("/tmp/_rs.bhm", precision=prec)
ctx = clf.create_batch_context(n)
S = 144000
for rate in (22050, 44100, 32000, 96000, 192000, 384000):
    src = int(np.ceil(S * rate / 48000))
    x = torch.randn((n, src), device="cuda") * 0.2
    y = torch.empty((n, S), device="cuda")
    for _ in range(2): clf.resample_device(ctx, x.data_ptr(), src, src, rate, 48000, y.data_ptr(), S, S, n)
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(5): clf.resample_device(ctx, x.data_ptr(), src, src, rate, 48000, y.data_ptr(), S, S, n)
    ctx.synchronize()
    ms = (time.perf_counter() - t) / 5 * 1e3
    print(f"{rate:>7} -> 48000: {ms:8.3f} ms per {n} segments of 3 s ({n / ms * 1e3:9.0f} segments/s)")

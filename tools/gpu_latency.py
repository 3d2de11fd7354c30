"""Latency of small calls (1, 8, 32, 64, 256 segments): device-resident forward and the host entry point bh_predict_batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier
m = synth.build_model("birdnet_v24")
path = "/tmp/v24.bhm"; mf.write_model(path, m)
clf = BirdClassifier(path, precision=os.environ.get("PREC", "f16x3"))
base = synth.synth_segments(16, m.sample_count, m.sample_rate)
for n in (1, 8, 32, 64, 256):
    ctx = clf.create_batch_context(n)
    host = np.ascontiguousarray(np.tile(base, (n // 16 + 1, 1))[:n])
    x = torch.from_numpy(host).cuda()
    logits = torch.empty((n, m.n_classes), device="cuda")
    idx = torch.empty((n, 5), dtype=torch.int32, device="cuda"); conf = torch.empty((n, 5), device="cuda")
    for _ in range(5):
        clf.forward_device(ctx, x.data_ptr(), n, logits.data_ptr(), idx.data_ptr(), conf.data_ptr()); ctx.synchronize()
    reps = 50
    t = time.perf_counter()
    for _ in range(reps):
        clf.forward_device(ctx, x.data_ptr(), n, logits.data_ptr(), idx.data_ptr(), conf.data_ptr()); ctx.synchronize()
    dev = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    for _ in range(reps):
        clf.forward_device(ctx, x.data_ptr(), n, logits.data_ptr(), idx.data_ptr(), conf.data_ptr())
    ctx.synchronize()
    pipe = (time.perf_counter() - t) / reps
    segs = [host[i] for i in range(n)]
    for _ in range(3):
        clf.predict_batch_with_context(ctx, segs)
    t = time.perf_counter()
    for _ in range(reps):
        clf.predict_batch_with_context(ctx, segs)
    hostt = (time.perf_counter() - t) / reps
    print(f"n {n:4d}: forward_device + sync {dev*1e3:7.3f} ms ({n/dev:8.0f} seg/s)   back-to-back {pipe*1e3:7.3f} ms ({n/pipe:8.0f} seg/s)   "
          f"bh_predict_batch (host in/out) {hostt*1e3:7.3f} ms ({n/hostt:8.0f} seg/s)")
    ctx.close()

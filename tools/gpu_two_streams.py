"""Experiment: two batch contexts (= two HIP streams) each running half of the segments, with a time offset so that one
stream's VALU-bound early blocks co-run with the other's latency-bound late blocks.  usage: gpu_two_streams.py [N] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m = synth.build_model("birdnet_v24")
path = "/tmp/birdnet_v24.bhm"; mf.write_model(path, m)
clf = BirdClassifier(path, precision=os.environ.get("BIRDA_HIP_PRECISION", "f16x3"))
base = synth.synth_segments(16, m.sample_count, m.sample_rate)
x = torch.from_numpy(np.tile(base, (N // 16 + 1, 1))[:N]).cuda()
logits = torch.empty((N, m.n_classes), device="cuda")

def run(parts, offset_frac):
    ctxs = [clf.create_batch_context(N // parts) for _ in range(parts)]
    per = N // parts
    def step():
        for p, c in enumerate(ctxs):
            clf.forward_device(c, x.data_ptr() + p * per * m.sample_count * 4, per, logits.data_ptr() + p * per * m.n_classes * 4)
    for _ in range(2): step()
    for c in ctxs: c.synchronize()
    if offset_frac and parts > 1:
        # run an extra partial amount of work on stream 0 only, so that stream 1 trails it by a fraction of a forward
        k = max(1, int(per * offset_frac))
        clf.forward_device(ctxs[0], x.data_ptr(), k, logits.data_ptr())
    t = time.time()
    for _ in range(steps): step()
    for c in ctxs: c.synchronize()
    dt = time.time() - t
    ref = logits.clone()
    for c in ctxs: c.close()
    return N * steps / dt, ref

r1, ref = run(1, 0)
print(f"1 stream  x {N}: {r1:9.0f} seg/s")
for parts, off in ((2, 0.0), (2, 0.5), (4, 0.0), (2, 0.25)):
    r, got = run(parts, off)
    print(f"{parts} streams offset {off}: {r:9.0f} seg/s  ({r / r1:.3f}x)  identical logits: {bool(torch.equal(ref, got))}")

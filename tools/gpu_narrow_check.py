"""Few-segment launches: forward time at n = 1 ... 256 and bit-identity of a small call's logits with the same segments' rows of a
1 000-segment call (the narrow-tile twins of the late blocks, mbconv_cfgs.inc / kernels_mbconv.hip mb_plan_narrow).
EXPERIMENTS build: BIRDA_HIP_MB_NARROW_MAX=0 switches the narrow tiles off.  usage: gpu_narrow_check.py [model]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier
kind = sys.argv[1] if len(sys.argv) > 1 else "birdnet_v24"
m = synth.build_model(kind)
path = f"/tmp/{kind}.bhm"; mf.write_model(path, m)
clf = BirdClassifier(path, precision=os.environ.get("PREC", "auto"))
N = 1000
base = synth.synth_segments(64, m.sample_count, m.sample_rate)
x = torch.from_numpy(np.tile(base, (N // 64 + 1, 1))[:N]).cuda()
big = torch.empty((N, m.n_classes), device="cuda")
ctx = clf.create_batch_context(N)
clf.forward_device(ctx, x.data_ptr(), N, big.data_ptr()); ctx.synchronize(); ctx.close()
line = []
for n in (1, 8, 20, 32, 64, 96, 128, 192, 256):
    c = clf.create_batch_context(n)
    lg = torch.empty((n, m.n_classes), device="cuda")
    for _ in range(5):
        clf.forward_device(c, x.data_ptr(), n, lg.data_ptr()); c.synchronize()
    same = bool(torch.equal(lg, big[:n]))
    t = time.perf_counter()
    for _ in range(40):
        clf.forward_device(c, x.data_ptr(), n, lg.data_ptr()); c.synchronize()
    dt = (time.perf_counter() - t) / 40
    line.append(f"n {n:3d}: {dt*1e3:6.3f} ms {n/dt:8.0f} seg/s {'identical' if same else 'DIFFERENT'}")
    c.close()
print(" | ".join(line))

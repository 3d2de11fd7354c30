"""Small-call latency with and without BH_FLAG_LOW_LATENCY (channel-split late blocks): forward_device + sync for 1 .. 64 segments,
and the logits of both against each other and the oracle's tolerance."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier
m = synth.build_model("birdnet_v24")
path = "/tmp/v24.bhm"; mf.write_model(path, m)
base = synth.synth_segments(16, m.sample_count, m.sample_rate)
out = {}
for ll in (False, True):
    clf = BirdClassifier(path, precision=os.environ.get("PREC", "auto"), low_latency=ll)
    for n in (1, 8, 20, 32, 64):
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
        out[(ll, n)] = (dev, logits.cpu().numpy().copy())
        ctx.close()
    clf.close()
for n in (1, 8, 20, 32, 64):
    a, b = out[(False, n)], out[(True, n)]
    scale = max(1.0, float(np.abs(a[1]).max()))
    print(f"n {n:3d}: plain {a[0]*1e3:6.3f} ms   low-latency {b[0]*1e3:6.3f} ms   ({n/b[0]:7.0f} seg/s)   max |dlogit| between them {float(np.abs(a[1]-b[1]).max())/scale:.2e} of the logit scale")

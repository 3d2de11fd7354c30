"""Logit error of the three GEMM precisions against the oracle, and their throughput (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier
from oracle import oracle as O

kind = sys.argv[1] if len(sys.argv) > 1 else "birdnet_v24"
m = synth.build_model(kind); path = f"/tmp/{kind}.bhm"; mf.write_model(path, m)
segs = synth.synth_segments(4, m.sample_count, m.sample_rate, start=100)
ref = O.OracleModel(path).forward(segs)
scale = max(1.0, float(np.abs(ref).max()))
N = 1000
x = torch.from_numpy(np.tile(synth.synth_segments(8, m.sample_count, m.sample_rate), (N // 8, 1))).cuda()
logits = torch.empty((N, m.n_classes), device="cuda")
for prec in ("f32", "f16x3", "f16"):
    clf = BirdClassifier(path, precision=prec)
    ctx = clf.create_batch_context(N)
    got = clf.predict_logits(ctx, segs)
    err = np.abs(got - ref).max(axis=1)
    for _ in range(2):
        clf.forward_device(ctx, x.data_ptr(), N, logits.data_ptr()); ctx.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        clf.forward_device(ctx, x.data_ptr(), N, logits.data_ptr())
    ctx.synchronize()
    dt = (time.perf_counter() - t) / 3
    ctx.set_profiling(True); clf.forward_device(ctx, x.data_ptr(), N, logits.data_ptr()); st = ctx.stage_ms()
    print(f"{prec:6s} fused {len(clf.fused_blocks()):2d} cfgs {clf.fused_blocks()}  max|dlogit|/scale {err.max()/scale:.3e} per-seg {np.array2string(err/scale, precision=2)}  {N/dt:8.0f} seg/s  mbconv {st['mbconv'][0]:.2f} ms")
    ctx.close(); clf.close()

"""Quick device-resident throughput + per-stage timing (diagnostic; run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier

kind = sys.argv[1] if len(sys.argv) > 1 else "birdnet_v24"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
mb = int(sys.argv[3]) if len(sys.argv) > 3 else 64
m = synth.build_model(kind)
path = f"/tmp/{kind}.bhm"; mf.write_model(path, m)
clf = BirdClassifier(path)
ctx = clf.create_batch_context(mb)
print("ctx device MB", ctx.device_bytes() / 1e6)
base = synth.synth_segments(min(N, 16), m.sample_count, m.sample_rate)
x = torch.from_numpy(np.tile(base, (N // base.shape[0] + 1, 1))[:N]).cuda()
logits = torch.empty((N, m.n_classes), device="cuda")
idx = torch.empty((N, 5), dtype=torch.int32, device="cuda"); conf = torch.empty((N, 5), device="cuda")
torch.cuda.synchronize()
for it in range(3):
    t = time.time()
    clf.forward_device(ctx, x.data_ptr(), N, logits.data_ptr(), idx.data_ptr(), conf.data_ptr())
    ctx.synchronize()
    dt = time.time() - t
    print(f"iter {it}: {dt*1e3:.2f} ms  {N/dt:.0f} seg/s")
ctx.set_profiling(True)
clf.forward_device(ctx, x.data_ptr(), N, logits.data_ptr(), idx.data_ptr(), conf.data_ptr())
st = ctx.stage_ms()
tot = sum(v[0] for v in st.values())
for k, (ms, n) in st.items():
    print(f"  {k:10s} {ms:9.3f} ms  {n:5d} launches  {100*ms/tot:5.1f}%  {ms*1e3/N:8.3f} us/seg")
print("total", tot, "ms ->", N / tot * 1e3, "seg/s (sum of kernel times)")
info = clf.info
print("pointwise TFLOP/s approx:", 2 * info.macs_per_segment * N / (st['pointwise'][0] * 1e-3) / 1e12 if st['pointwise'][0] else 0)
print("mel TFLOP/s:", info.mel_flops_per_segment * N / (st['mel'][0] * 1e-3) / 1e12, " mel GB/s algorithmic:", 968448 * N / (st['mel'][0] * 1e-3) / 1e9)

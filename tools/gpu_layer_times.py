"""Per-launch times of one forward (HIP events per layer; fused blocks are booked on their first layer): python tools/gpu_layer_times.py [kind] [n] [precision]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier
kind = sys.argv[1] if len(sys.argv) > 1 else "perch_v2"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
prec = sys.argv[3] if len(sys.argv) > 3 else "f16x3"
m = synth.build_model(kind)
path = f"/tmp/{kind}.bhm"; mf.write_model(path, m)
clf = BirdClassifier(path, precision=prec, low_latency=os.environ.get("LL") == "1")
ctx = clf.create_batch_context(N)
base = synth.synth_segments(16, m.sample_count, m.sample_rate)
x = torch.from_numpy(np.tile(base, (N // 16 + 1, 1))[:N]).cuda()
logits = torch.empty((N, m.n_classes), device="cuda"); idx = torch.empty((N, 5), dtype=torch.int32, device="cuda"); conf = torch.empty((N, 5), device="cuda")
for _ in range(3):
    clf.forward_device(ctx, x.data_ptr(), N, logits.data_ptr(), idx.data_ptr(), conf.data_ptr())
ctx.synchronize()
ctx.set_profiling(True)
for _ in range(5):
    clf.forward_device(ctx, x.data_ptr(), N, logits.data_ptr(), idx.data_ptr(), conf.data_ptr())
ctx.synchronize()
st = ctx.stage_ms(); ly = ctx.layer_ms()
print({k: round(v[0] / 5 * 1e3 / N, 3) for k, v in st.items()}, "us/segment")
names = {1: "conv", 2: "dw", 3: "pw", 4: "gap", 5: "dense", 6: "scale"}
for i, (ms, n) in enumerate(ly):
    if n:
        L = m.layers[i]
        print(f"layer {i:3d} {names[L.op]:5s} {L.cin:5d}->{L.cout:5d} {L.in_h}x{L.in_w} k{L.kh} s{L.sh}  {ms / 5 * 1e3:9.1f} us per {N}")

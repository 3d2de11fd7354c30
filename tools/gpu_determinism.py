"""Identical inputs anywhere in a batch must give bit-identical logits (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier

kind = sys.argv[1] if len(sys.argv) > 1 else "birdnet_v24"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
m = synth.build_model(kind)
path = f"/tmp/{kind}.bhm"; mf.write_model(path, m)
clf = BirdClassifier(path, precision=os.environ.get("BIRDA_HIP_PRECISION", "f32"))
print("fused:", clf.fused_blocks())
uniq = synth.synth_segments(8, m.sample_count, m.sample_rate)
order = np.arange(n) % 8
x = torch.from_numpy(uniq[order]).cuda()
logits = torch.empty((n, m.n_classes), device="cuda")
ctx = clf.create_batch_context(256)
for rep in range(3):
    clf.forward_device(ctx, x.data_ptr(), n, logits.data_ptr()); ctx.synchronize()
    a = logits.cpu().numpy()
    bad = 0; worst = 0.0
    for k in range(8):
        rows = a[order == k]
        d = np.abs(rows - rows[0]).max(axis=1)
        bad += int((d > 0).sum()); worst = max(worst, float(d.max()))
    print(f"rep {rep}: rows differing from their twin: {bad} of {n}, worst |diff| {worst:.3e}")

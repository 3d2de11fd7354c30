"""Logits must not depend on how a batch is sliced into micro-batches (diagnostic).
usage: gpu_slice_invariance.py [kind] [n] [mb_a] [mb_b] [reps]   (precision: BIRDA_HIP_PRECISION)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier

kind = sys.argv[1] if len(sys.argv) > 1 else "birdnet_v24"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
mba = int(sys.argv[3]) if len(sys.argv) > 3 else 256
mbb = int(sys.argv[4]) if len(sys.argv) > 4 else 96
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 4
m = synth.build_model(kind)
path = f"/tmp/{kind}.bhm"; mf.write_model(path, m)
clf = BirdClassifier(path, precision=os.environ.get("BIRDA_HIP_PRECISION", "f32"))
uniq = synth.synth_segments(8, m.sample_count, m.sample_rate)
x = torch.from_numpy(uniq[np.arange(n) % 8]).cuda()
la = torch.empty((n, m.n_classes), device="cuda"); lb = torch.empty_like(la)
ca, cb = clf.create_batch_context(mba), clf.create_batch_context(mbb)
for r in range(reps):
    clf.forward_device(ca, x.data_ptr(), n, la.data_ptr()); ca.synchronize()
    clf.forward_device(cb, x.data_ptr(), n, lb.data_ptr()); cb.synchronize()
    d = (la - lb).abs().max(dim=1).values.cpu().numpy()
    bad = np.nonzero(d > 0)[0]
    print(f"rep {r}: rows differing between micro-batch {mba} and {mbb}: {len(bad)} of {n}; first {bad[:12].tolist()} max {d.max():.3e}")

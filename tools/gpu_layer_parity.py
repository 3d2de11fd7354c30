"""Layer-by-layer HIP vs oracle comparison (diagnostic; run on the GPU box).

usage: python tools/gpu_layer_parity.py [kind] [n_segments]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BIRDA_HIP_KEEP_TENSORS"] = "1"
import numpy as np

from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier
from oracle import oracle as O

kind = sys.argv[1] if len(sys.argv) > 1 else "mini"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m = synth.build_model(kind)
path = f"/tmp/{kind}.bhm"
mf.write_model(path, m)
segs = synth.synth_segments(n, m.sample_count, m.sample_rate)
om = O.OracleModel(path)
clf = BirdClassifier(path)
ctx = clf.create_batch_context(n)
t = time.time()
logits = clf.predict_logits(ctx, segs)
print(f"hip forward {time.time()-t:.3f}s")
t = time.time()
ref_logits = om.forward(segs)
print(f"oracle forward {time.time()-t:.3f}s")
worst = 0
for ti in range(0, len(m.layers) + 1):
    ref = om.forward(segs, dump_tensor=ti)[1]
    got = clf.read_tensor(ctx, ti, n)
    d = np.abs(ref - got)
    scale = np.abs(ref).max()
    name = "spec" if ti == 0 else f"L{ti-1} op{m.layers[ti-1].op}"
    print(f"tensor {ti:3d} {name:10s} max|d|={d.max():.3e} mean|d|={d.mean():.3e} max|ref|={scale:.3e} "
          f"nan={int(np.isnan(got).sum())}")
print("logits max|d|", np.abs(logits - ref_logits).max(), "max|logit|", np.abs(ref_logits).max())

"""Output tensor of every fused block against the oracle, for one forced tile configuration (diagnostic).
usage: gpu_block_debug.py <model kind> <cfg> <precision>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BIRDA_HIP_KEEP_TENSORS"] = "1"; os.environ["BIRDA_HIP_KEEP_FUSED"] = "1"
import numpy as np
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier
from oracle import oracle as O

kind, cfg, prec = sys.argv[1], int(sys.argv[2]), sys.argv[3]
if cfg >= 0:
    os.environ["BIRDA_HIP_MB_CFG"] = str(cfg)   # < 0: the planner's own choice per block
m = synth.build_model(kind)
path = f"/tmp/{kind}.bhm"; mf.write_model(path, m)
n = 2
segs = synth.synth_segments(n, m.sample_count, m.sample_rate, start=7)
clf = BirdClassifier(path, precision=prec)
print("fused blocks:", clf.fused_blocks())
ctx = clf.create_batch_context(n)
clf.predict_logits(ctx, segs)
om = O.OracleModel(path)
for li, L in enumerate(m.layers):
    t = li + 1
    if L.op not in (mf.OP_PWCONV,) or L.act != mf.ACT_NONE:
        continue   # project layers = block outputs
    ref = om.forward(segs, dump_tensor=t)[1].reshape(n, L.out_h, L.out_w, L.cout)
    got = clf.read_tensor(ctx, t, n).reshape(n, L.out_h, L.out_w, L.cout)
    d = np.abs(got - ref)
    msg = f"tensor {t:2d} {L.out_h}x{L.out_w}x{L.cout}: max err {d.max():.3e} (scale {np.abs(ref).max():.2f})"
    if d.max() > 1e-3 * max(1.0, np.abs(ref).max()):
        bad = np.argwhere(d.max(axis=3) > 1e-3)
        ys = sorted(set(bad[:, 1].tolist())); xs = sorted(set(bad[:, 2].tolist()))
        msg += f"  BAD pixels {len(bad)}: rows {ys[:20]} cols {xs[:40]}"
    print(msg)

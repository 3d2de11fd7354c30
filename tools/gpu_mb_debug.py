"""Per tile-configuration / per-segment error of the fused MBConv path vs the oracle (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier
from oracle import oracle as O

kind = sys.argv[1] if len(sys.argv) > 1 else "mini_b0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 55)
m = synth.build_model(kind)
path = f"/tmp/{kind}.bhm"; mf.write_model(path, m)
segs = synth.synth_segments(n, m.sample_count, m.sample_rate, start=7)
ref = O.OracleModel(path).forward(segs)
for cfg in range(lo, hi):
    prec = "f32" if cfg < 22 or cfg in (63, 64) else ("f16x3" if (cfg % 2 == 0 or cfg >= 48) else "f16")
    os.environ["BIRDA_HIP_MB_CFG"] = str(cfg)
    clf = BirdClassifier(path, precision=prec)
    blocks = clf.fused_blocks()
    if blocks:
        ctx = clf.create_batch_context(8)
        got = clf.predict_logits(ctx, segs)
        err = np.abs(got - ref).max(axis=1)
        print(f"cfg {cfg:2d} {prec:6s} blocks {len(blocks)} per-seg max err {np.array2string(err, precision=2)}")
        ctx.close()
    clf.close()

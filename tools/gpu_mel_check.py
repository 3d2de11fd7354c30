import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BIRDA_HIP_KEEP_TENSORS"] = "1"
import numpy as np
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier
for kind in ("mini", "birdnet_v24_tiny"):
    m = synth.build_model(kind); path = f"/tmp/{kind}.bhm"; mf.write_model(path, m)
    segs = synth.synth_segments(2, m.sample_count, m.sample_rate, start=3)
    out = {}
    for prec in ("f32", "f16x3"):
        clf = BirdClassifier(path, precision=prec); ctx = clf.create_batch_context(2)
        clf.predict_logits(ctx, segs)
        out[prec] = clf.read_tensor(ctx, 0, 2).reshape(2, len(m.branches), m.branches[0].n_mels, m.branches[0].n_frames)
        ctx.close(); clf.close()
    d = np.abs(out["f32"] - out["f16x3"])
    for b in range(len(m.branches)):
        print(kind, "branch", b, "L", m.branches[b].frame_length, "max diff", d[:, b].max(), "mean", d[:, b].mean(), "per-frame-tile max", [float(d[:, b, :, t:t+16].max()) for t in range(0, m.branches[0].n_frames, 16)][:9])
    if kind == "mini":
        a, b = out["f32"][0, 0], out["f16x3"][0, 0]
        np.set_printoptions(precision=4, linewidth=200)
        print("f32   mel 0..3, frames 30..36:\n", a[:4, 30:37]); print("f16x3:\n", b[:4, 30:37])
        print("ratio:\n", (b[:4, 32:40] + 0.4) / (a[:4, 32:40] + 0.4))
        bad = np.argwhere(d[0, 0] > 1e-3); print("bad mel rows:", sorted(set(bad[:, 0]))[:40], "bad frames:", sorted(set(bad[:, 1]))[:60])

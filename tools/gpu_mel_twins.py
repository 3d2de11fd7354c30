"""Spectrogram (tensor 0) of identical segments at different batch positions must be bit-identical (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BIRDA_HIP_KEEP_TENSORS"] = "1"
import numpy as np, torch
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier

kind = sys.argv[1] if len(sys.argv) > 1 else "birdnet_v24"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = synth.build_model(kind)
path = f"/tmp/{kind}.bhm"; mf.write_model(path, m)
clf = BirdClassifier(path, precision=os.environ.get("BIRDA_HIP_PRECISION", "f32"))
uniq = synth.synth_segments(8, m.sample_count, m.sample_rate)
order = np.arange(n) % 8
x = torch.from_numpy(uniq[order]).cuda()
logits = torch.empty((n, m.n_classes), device="cuda")
ctx = clf.create_batch_context(n)
nb = len(m.branches)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for rep in range(reps):
    clf.forward_device(ctx, x.data_ptr(), n, logits.data_ptr()); ctx.synchronize()
    sp = clf.read_tensor(ctx, 0, n).reshape(n, nb, m.spec_h, m.spec_w)
    for k in range(8):
        rows = sp[order == k]
        # majority value per element = median
        med = np.median(rows, axis=0)
        d = rows != med[None]
        if d.any():
            seg, br, mel, t = np.nonzero(d)
            print(f"rep {rep} uniq {k}: {d.sum()} elements differ; segs {sorted(set(seg.tolist()))} branches {sorted(set(br.tolist()))} "
                  f"mels {sorted(set(mel.tolist()))[:12]}.. frames min {t.min()} max {t.max()} n_frames {len(set(t.tolist()))} "
                  f"frame%48 {sorted(set((t % 48).tolist()))[:20]} maxdiff {np.abs(rows - med[None]).max():.3e}")
print("done")

"""Soak: the same batch through the same context N times must give bit-identical logits every time, and identical segments at
different batch positions identical rows (races between waves / DMA pieces show up here first).  python tools/gpu_soak.py [kind] [reps] [micro-batch]
(micro-batch <= 256 runs the late blocks on their one-segment-per-workgroup twins, mbconv_cfgs.inc 133-136)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier

kind = sys.argv[1] if len(sys.argv) > 1 else "birdnet_v24"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = 1000
m = synth.build_model(kind)
path = f"/tmp/{kind}.bhm"; mf.write_model(path, m)
for prec in ("f16x3", "f16"):
    clf = BirdClassifier(path, precision=prec)
    uniq = synth.synth_segments(8, m.sample_count, m.sample_rate)
    order = np.arange(n) % 8
    x = torch.from_numpy(uniq[order]).cuda()
    logits = torch.empty((n, m.n_classes), device="cuda")
    ctx = clf.create_batch_context(int(sys.argv[3]) if len(sys.argv) > 3 else 1000)
    clf.forward_device(ctx, x.data_ptr(), n, logits.data_ptr()); ctx.synchronize()
    ref = logits.clone()
    twins = sum(int((ref[order == k] != ref[k]).any(dim=1).sum()) for k in range(8))
    bad = 0
    for r in range(reps):
        clf.forward_device(ctx, x.data_ptr(), n, logits.data_ptr()); ctx.synchronize()
        bad += int((logits != ref).any(dim=1).sum())
    print(f"{kind} {prec}: fused {clf.fused_blocks()}  rows differing from their twin {twins}, rows differing from run 0 over {reps} runs: {bad}")
    ctx.close(); clf.close()

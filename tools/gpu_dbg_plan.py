import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier
seed = int(sys.argv[1]); prec = sys.argv[2]
P = synth.random_plan(seed, big=True)
print(P)
m = synth.build_model("custom", plan=P)
mf.write_model("/tmp/_d.bhm", m)
clf = BirdClassifier("/tmp/_d.bhm", precision=prec)
print("fused", clf.fused_blocks())
for i, L in enumerate(m.layers): print(i, L.op, L.act, L.cin, L.cout, L.kh, L.sh, L.in_h, L.in_w, L.out_h, L.out_w)
ctx = clf.create_batch_context(8)
segs = synth.synth_segments(8, m.sample_count, m.sample_rate)
try:
    print(clf.predict_logits(ctx, segs)[0, :4])
except Exception as e:
    print("ERR", e)

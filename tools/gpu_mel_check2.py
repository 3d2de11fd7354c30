import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BIRDA_HIP_KEEP_TENSORS"] = "1"
import numpy as np
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier
m = synth.build_model("mini"); path = "/tmp/mini.bhm"; mf.write_model(path, m)
segs = synth.synth_segments(2, m.sample_count, m.sample_rate, start=3)
outs = []
for rep in range(3):
    clf = BirdClassifier(path, precision="f16x3"); ctx = clf.create_batch_context(2)
    clf.predict_logits(ctx, segs)
    outs.append(clf.read_tensor(ctx, 0, 2).copy()); ctx.close(); clf.close()
print("run-to-run max diff:", np.abs(outs[0] - outs[1]).max(), np.abs(outs[0] - outs[2]).max())

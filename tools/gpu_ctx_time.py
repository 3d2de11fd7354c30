import os, sys, time, tempfile
sys.path.insert(0, "/root/repo")
import torch
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier
m = synth.build_model("birdnet_v24")
d = tempfile.mkdtemp(); path = os.path.join(d, "m.bhm"); mf.write_model(path, m)
clf = BirdClassifier(path, None, precision="f16x3")
for n in (1000, 1000, 1000, 256, 256):
    t = time.perf_counter(); ctx = clf.create_batch_context(n); t1 = time.perf_counter(); ctx.close(); t2 = time.perf_counter()
    print(f"n={n}: create {1e3*(t1-t):.2f} ms, destroy {1e3*(t2-t1):.2f} ms")

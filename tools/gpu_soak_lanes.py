"""Soak of the host-fed path: the same PCM16 stream through bh_predict_pcm16 N times (sub-slices on the context's lanes, each in its
own part of the arena) must give the same rows every time, and the rows of one whole-slice forward.  python tools/gpu_soak_lanes.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
m = synth.build_model("birdnet_v24")
path = "/tmp/birdnet_v24.bhm"; mf.write_model(path, m)
clf = BirdClassifier(path, precision="f16x3", top_k=5, min_confidence=0.0)
n = 1000
host = synth.synth_segments(64, m.sample_count, m.sample_rate)
x = np.tile(host, (n // 64 + 1, 1))[:n].reshape(-1)
pcm = np.clip(np.round(x.astype(np.float64) * 32767.0), -32768, 32767).astype(np.int16)
ctx = clf.create_batch_context(n)
key = lambda res: [[(p.index, p.confidence) for p in r.predictions] for r in res]
ctx.set_sub_slices(1)
ref = key(clf.predict_pcm16(ctx, pcm, m.sample_rate, 0)[0])
ctx.set_sub_slices(0)
bad = 0
for r in range(reps):
    got = key(clf.predict_pcm16(ctx, pcm, m.sample_rate, 0)[0])
    bad += sum(a != b for a, b in zip(got, ref))
print(f"{reps} runs of {n} segments on lanes: rows differing from the whole-slice forward: {bad}")

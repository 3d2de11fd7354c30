"""Many short recordings: one file at a time (bhh_process_file) against packed uploads (bhh_process_files), with the phase times
of BIRDA_HOST_TIMING.  usage: gpu_short_files.py [n_files] [segments_per_file]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("TIMING"): os.environ["BIRDA_HOST_TIMING"] = "1"
import numpy as np, torch
from birda_amd import modelfile as mf, pipeline, synth
from birda_amd.classifier import BirdClassifier
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 50
per = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m = synth.build_model("birdnet_v24")
d = tempfile.mkdtemp()
path = os.path.join(d, "m.bhm"); mf.write_model(path, m)
labels = os.path.join(d, "l.txt"); synth.write_labels(labels, m.n_classes)
uniq = synth.synth_segments(16, m.sample_count, m.sample_rate)
files = []
for k in range(nf):
    x = np.tile(uniq, (per // 16 + 1, 1))[:per].reshape(-1)
    f = os.path.join(d, f"r{k:03d}.wav"); synth.write_wav_pcm16(f, x, m.sample_rate); files.append(f)
clf = BirdClassifier(path, labels, precision="auto", low_latency=os.environ.get("LL") == "1")
out = os.path.join(d, "out"); os.makedirs(out)
for rep in range(3):
    t = time.perf_counter(); res, st = pipeline.process_files_packed(clf, files, out); dt = time.perf_counter() - t
    print(f"packed rep {rep}: {nf * per / dt:9.0f} segments/s ({dt*1e3:.1f} ms)", file=sys.stderr)
if os.environ.get("ALSO_SINGLE"):
    for rep in range(2):
        t = time.perf_counter()
        for f in files:
            pipeline.process_file(clf, f, out)
        dt = time.perf_counter() - t
        print(f"one file at a time rep {rep}: {nf * per / dt:9.0f} segments/s ({dt*1e3:.1f} ms)", file=sys.stderr)

"""Long recordings through bhh_process_files (each a pack of its own, two in flight), with BIRDA_HOST_TIMING's phase times.
usage: gpu_long_files.py [n_files] [segments_per_file]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BIRDA_HOST_TIMING"] = "1"
import numpy as np
if os.environ.get("WITH_TORCH"):
    import torch
    _keep = torch.zeros(1000, 144000, device="cuda")
from birda_amd import modelfile as mf, pipeline, synth
from birda_amd.classifier import BirdClassifier
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 6
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
m = synth.build_model("birdnet_v24")
d = tempfile.mkdtemp()
path = os.path.join(d, "m.bhm"); mf.write_model(path, m)
labels = os.path.join(d, "l.txt"); synth.write_labels(labels, m.n_classes)
uniq = synth.synth_segments(int(os.environ.get("UNIQ", "16")), m.sample_count, m.sample_rate)
x = np.tile(uniq, (per // uniq.shape[0] + 1, 1))[:per].reshape(-1)
first = os.path.join(d, "r000.wav"); synth.write_wav_pcm16(first, x, m.sample_rate)
files = [first]
for k in range(1, nf):
    f = os.path.join(d, f"r{k:03d}.wav"); os.link(first, f); files.append(f)
clf = BirdClassifier(path, labels, precision="f16x3")
out = os.path.join(d, "out"); os.makedirs(out)
for rep in range(int(os.environ.get("REPS", "4"))):
    t = time.perf_counter(); res, st = pipeline.process_files_packed(clf, files, out); dt = time.perf_counter() - t
    print(f"rep {rep}: {nf * per / dt:9.0f} segments/s ({dt*1e3:.1f} ms for {nf} files)", file=sys.stderr)

"""End-to-end per-file pipeline timing (bhh_process_file, WAV in -> CSV out) with the phase times of BIRDA_HOST_TIMING."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BIRDA_HOST_TIMING"] = "1"
import numpy as np, torch
from birda_amd import modelfile as mf, pipeline, synth
from birda_amd.classifier import BirdClassifier
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
m = synth.build_model("birdnet_v24")
d = tempfile.mkdtemp()
path = os.path.join(d, "m.bhm"); mf.write_model(path, m)
labels = os.path.join(d, "l.txt"); synth.write_labels(labels, m.n_classes)
uniq = synth.synth_segments(16, m.sample_count, m.sample_rate)
x = np.tile(uniq, (n // 16 + 1, 1))[:n].reshape(-1)
wav = os.path.join(d, "a.wav"); synth.write_wav_pcm16(wav, x, m.sample_rate)
clf = BirdClassifier(path, labels, precision="f16x3")
for bs in (0, 256, 512):
    for rep in range(3):
        print(f"--- batch_size {bs} rep {rep}", file=sys.stderr)
        r = pipeline.process_file(clf, wav, d, batch_size=bs)
        print(f"batch {bs}: {r.segments_per_sec:9.1f} segments/s ({r.front_end}, effective {r.effective_batch})", file=sys.stderr)

"""bhh_process_file's host phases (BIRDA_HOST_TIMING=1) on a 1 000-segment PCM16 WAV: python tools/gpu_e2e_timing.py"""
import os, sys, tempfile
os.environ["BIRDA_HOST_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from birda_amd import modelfile as mf, synth, pipeline
from birda_amd.classifier import BirdClassifier
tmp = tempfile.mkdtemp()
m = synth.build_model("birdnet_v24"); path = os.path.join(tmp, "m.bhm"); mf.write_model(path, m)
labels = os.path.join(tmp, "l.txt"); synth.write_labels(labels, m.n_classes)
uniq = synth.synth_segments(16, m.sample_count, m.sample_rate)
host = np.ascontiguousarray(np.tile(uniq, (1000 // 16 + 1, 1))[:1000])
wav = os.path.join(tmp, "f.wav"); synth.write_wav_pcm16(wav, host.reshape(-1), m.sample_rate)
clf = BirdClassifier(path, labels, precision="auto")
for i in range(5):
    r = pipeline.process_file(clf, wav, tmp, front_end="device")
    print("run", i, round(r.segments_per_sec, 1), "segments/s", round(r.duration_secs * 1e3, 2), "ms", flush=True)

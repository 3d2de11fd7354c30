"""bh_predict_pcm16 (pageable / pinned) and bhh_process_file on a 1 000-segment stream: python tools/gpu_pcm_legs.py"""
import os, sys, time, statistics, tempfile, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from birda_amd import modelfile as mf, synth, pipeline
from birda_amd.classifier import BirdClassifier
from birda_amd._lib import BhResult, check
tmp = tempfile.mkdtemp()
m = synth.build_model("birdnet_v24"); path = os.path.join(tmp, "m.bhm"); mf.write_model(path, m)
labels = os.path.join(tmp, "l.txt"); synth.write_labels(labels, m.n_classes)
n = 1000
uniq = synth.synth_segments(16, m.sample_count, m.sample_rate)
host = np.ascontiguousarray(np.tile(uniq, (n // 16 + 1, 1))[:n])
pcm = np.clip(np.round(host.reshape(-1).astype(np.float64) * 32767.0), -32768, 32767).astype(np.int16)
clf = BirdClassifier(path, labels, precision="auto")
ctx = clf.create_batch_context(n)
arr = (BhResult * n)(); starts = (C.c_uint64 * n)(); n_out = C.c_size_t()
call = lambda: check(clf._L.bh_predict_pcm16(clf._h, ctx._h, pcm.ctypes.data, pcm.shape[0], 1, m.sample_rate, 0, arr, n, C.byref(n_out), starts))
def timed(fn, reps=7):
    fn(); ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return statistics.median(ts)
t = timed(call); print("pcm16 pageable %.1f k segments/s (%.2f ms)" % (n / t / 1e3, t * 1e3))
check(clf._L.bh_host_register(pcm.ctypes.data, pcm.nbytes))
t = timed(call); print("pcm16 pinned   %.1f k segments/s (%.2f ms)" % (n / t / 1e3, t * 1e3))
check(clf._L.bh_host_unregister(pcm.ctypes.data))
ctx.close()
wav = os.path.join(tmp, "f.wav"); synth.write_wav_pcm16(wav, host.reshape(-1), m.sample_rate)
pipeline.process_file(clf, wav, tmp, front_end="device")
rs = sorted(pipeline.process_file(clf, wav, tmp, front_end="device").segments_per_sec for _ in range(5))
print("process_file   %.1f k segments/s (median of 5)" % (rs[2] / 1e3))

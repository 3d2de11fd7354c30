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
# PRE: what else the process holds before the files run (the bench process's last leg runs 20 % below a process of its own:
# DESIGN.md section 6).  PRE=clf: a second classifier with a 1 000-segment context that has run forwards; PRE=pin: 1.2 GB of
# pinned host memory; PRE=ctx: three more 1 000-segment contexts of the SAME classifier, created and parked first.
pre = os.environ.get("PRE", "")
clf = BirdClassifier(path, labels, precision=os.environ.get("PRECISION", "auto"))
_hold = []
if "clf" in pre:
    import torch
    c0 = BirdClassifier(path, labels, precision="auto"); x0 = c0.create_batch_context(1000)
    xs = torch.from_numpy(np.tile(uniq, (63, 1))[:1000]).cuda(); lg = torch.empty((1000, m.n_classes), device="cuda")
    ti = torch.empty((1000, 5), dtype=torch.int32, device="cuda"); tc = torch.empty((1000, 5), device="cuda")
    for _ in range(30):
        c0.forward_device(x0, xs.data_ptr(), 1000, lg.data_ptr(), ti.data_ptr(), tc.data_ptr())
    x0.synchronize(); _hold += [c0, x0, xs, lg]
if "pin" in pre:
    import ctypes as C
    from birda_amd import _lib
    pp = C.c_void_p(); _lib.load().bh_host_alloc(1200 << 20, C.byref(pp)); _hold.append(pp)
if "ctx" in pre:
    cs = [clf.create_batch_context(1000) for _ in range(3)]
    for c_ in cs: c_.close()
out = os.path.join(d, "out"); os.makedirs(out)
for rep in range(int(os.environ.get("REPS", "4"))):
    t = time.perf_counter(); res, st = pipeline.process_files_packed(clf, files, out); dt = time.perf_counter() - t
    print(f"rep {rep}: {nf * per / dt:9.0f} segments/s ({dt*1e3:.1f} ms for {nf} files)", file=sys.stderr)

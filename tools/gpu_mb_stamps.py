"""Where the fused MBConv kernels spend their wave-cycles (diagnostic build of the same kernels: LIBX=1 tools/ab.sh x python
tools/gpu_mb_stamps.py [n] [kind]; for pass A of a squeeze-excite block the "P3" column is its store phase)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BIRDA_HIP_MB_STAMPS"] = "1"
import numpy as np, torch
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
KIND = sys.argv[2] if len(sys.argv) > 2 else "birdnet_v24"
m = synth.build_model(KIND)
path = f"/tmp/{KIND}.bhm"; mf.write_model(path, m)
clf = BirdClassifier(path, precision=os.environ.get("PREC", "f16x3"))
ctx = clf.create_batch_context(N)
base = synth.synth_segments(8, m.sample_count, m.sample_rate)
x = torch.from_numpy(np.tile(base, (N // 8 + 1, 1))[:N]).cuda()
logits = torch.empty((N, m.n_classes), device="cuda")
clf.forward_device(ctx, x.data_ptr(), N, logits.data_ptr()); ctx.synchronize()
buf = (C.c_uint64 * 2048)()
clf._L.bh_debug_mb_stamps(clf._h, buf, 2048)
clf.forward_device(ctx, x.data_ptr(), N, logits.data_ptr()); ctx.synchronize()
nb = clf._L.bh_debug_mb_stamps(clf._h, buf, 2048)
names = ["setup", "wstage", "P1", "bar1", "P2", "bar2", "P3", "epi"]
cfgs = clf.fused_blocks()
print("block cfg   total_Mcyc " + " ".join(f"{n:>7s}" for n in names))
for b in range(nb):
    v = np.array([buf[b * 8 + i] for i in range(8)], dtype=np.float64)
    print(f"{b:5d} {cfgs[b]:3d} {v.sum()/1e6:10.1f}   " + " ".join(f"{100*x/v.sum():6.1f}%" for x in v))

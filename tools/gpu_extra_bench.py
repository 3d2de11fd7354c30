"""C4 / C5 side measurements on one GPU (diagnostic; the headline metric lives in bench.py):
Perch-shaped model throughput and the device resampler's rate for 44.1 k / 22.05 k -> 48 k."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier

def timeit(fn, sync, n=5):
    fn(); sync()
    t = time.perf_counter()
    for _ in range(n): fn()
    sync()
    return (time.perf_counter() - t) / n

N = 1000
PREC = os.environ.get("BIRDA_HIP_BENCH_PRECISION", "f16x3")
m = synth.build_model("birdnet_v24"); p = "/tmp/v24.bhm"; mf.write_model(p, m)
clf = BirdClassifier(p, precision=PREC); ctx = clf.create_batch_context(N)
# PCIe-inclusive: host f32 segments -> pinned staging -> device -> classify -> top-k back (bh_predict_batch_contig)
import ctypes as C
from birda_amd._lib import BhResult
hseg = np.tile(synth.synth_segments(8, m.sample_count, m.sample_rate), (N // 8, 1))
res = (BhResult * N)()
def host_path():
    rc = clf._L.bh_predict_batch_contig(clf._h, ctx._h, hseg.ctypes.data, N, res)
    assert rc == 0
dt = timeit(host_path, lambda: None, 3)
print(f"host f32 segments in, top-k out (PCIe inclusive, bh_predict_batch_contig): {N/dt:.0f} seg/s ({hseg.nbytes/dt/1e9:.1f} GB/s of PCM)")
pcm = np.clip(np.round(hseg.reshape(-1) * 32767), -32768, 32767).astype(np.int16)
cap = N + 8; res2 = (BhResult * cap)(); nseg = C.c_size_t()
def pcm_path():
    rc = clf._L.bh_predict_pcm16(clf._h, ctx._h, pcm.ctypes.data, pcm.size, 1, m.sample_rate, 0, res2, cap, C.byref(nseg), None)
    assert rc == 0 and nseg.value == N
dt = timeit(pcm_path, lambda: None, 3)
print(f"host int16 stream in (bh_predict_pcm16, device-side segmenting): {N/dt:.0f} seg/s ({pcm.nbytes/dt/1e9:.1f} GB/s of PCM)")
out = torch.empty((N, m.sample_count), device="cuda")
logits = torch.empty((N, m.n_classes), device="cuda")
for rate in (44100, 22050):
    src = int(np.ceil(m.sample_count * rate / m.sample_rate))
    x = torch.from_numpy(np.tile(synth.synth_segments(8, src, rate), (N // 8, 1))).cuda()
    dt = timeit(lambda: clf.resample_device(ctx, x.data_ptr(), src, src, rate, m.sample_rate, out.data_ptr(), m.sample_count, m.sample_count, N), ctx.synchronize)
    print(f"resample {rate} -> {m.sample_rate}: {dt*1e3:.2f} ms / {N} segments = {dt/N*1e6:.2f} us/seg ({N/dt:.0f} seg/s)")
    def both():
        clf.resample_device(ctx, x.data_ptr(), src, src, rate, m.sample_rate, out.data_ptr(), m.sample_count, m.sample_count, N)
        clf.forward_device(ctx, out.data_ptr(), N, logits.data_ptr())
    dt = timeit(both, ctx.synchronize)
    print(f"  resample + classify: {N/dt:.0f} seg/s")
ctx.close(); clf.close()
m = synth.build_model("perch_v2"); p = "/tmp/perch.bhm"; mf.write_model(p, m)
clf = BirdClassifier(p, precision=PREC); ctx = clf.create_batch_context(N)
print("perch fused blocks:", clf.fused_blocks())
x = torch.from_numpy(np.tile(synth.synth_segments(8, m.sample_count, m.sample_rate), (N // 8, 1))).cuda()
logits = torch.empty((N, m.n_classes), device="cuda")
dt = timeit(lambda: clf.forward_device(ctx, x.data_ptr(), N, logits.data_ptr()), ctx.synchronize)
print(f"perch-shaped (5 s / 32 kHz, 14795 classes): {N/dt:.0f} seg/s")
ctx.set_profiling(True); clf.forward_device(ctx, x.data_ptr(), N, logits.data_ptr())
print({k: round(v[0], 2) for k, v in ctx.stage_ms().items()})

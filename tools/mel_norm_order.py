"""Would the front-end stay exact if the min / max normalisation moved BEHIND the folded GEMM?

    in front (what ships):   T = Gf^T fold(a x - c)            a = 2 / (max - min + eps), c = a min + 1
    behind (the proposal):   T = a Gf^T fold(p x) / p - 2 c Gf^T 1     (p: a power-of-two pre-scale; fold(x)[j] = x[j+1] + x[L-1-j])

Both are emulated in numpy with the split-f16 product of the device (operands split into f16 hi + lo, hi*hi + hi*lo + lo*hi summed
in f32) on BirdNET's first branch (L = 2048, hop 278, 96 mels) and compared with float64: loud audio, and the quiet recording
with a DC offset that `tests/test_parity_gpu.py::test_front_end_on_quiet_audio_with_a_dc_offset` runs on the device.
    python tools/mel_norm_order.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from birda_amd import synth

m = synth.build_model("birdnet_v24_tiny")
br = m.branches[0]
L, H, K = br.frame_length, br.frame_step, br.frame_length // 2
W = m.weight(br.mel_w_off, br.n_bins * br.n_mels).reshape(br.n_bins, br.n_mels).astype(np.float64)
n = np.arange(L)
G = (0.5 - 0.5 * np.cos(2 * np.pi * n / L))[:, None] * np.cos(2 * np.pi * ((n[:, None] * np.arange(br.n_bins)[None, :]) % L) / L) @ W   # [L, mels]
Gf = G[1:K + 1]                                                                  # folded operator rows j = 0 .. K-1 <-> sample j + 1
sp = 2.0 ** (13 - np.ceil(np.log2(np.abs(Gf).max())))                            # the library's operand pre-scale (f16_scale_exponent)
Gs = (Gf * sp).astype(np.float32)
Gh = Gs.astype(np.float16); Gl = (Gs - Gh.astype(np.float32)).astype(np.float16)


def split_gemm(Y):
    """[frames, K] f32 x Gf -> [frames, mels], as three f16 products with f32 accumulation"""
    Yh = Y.astype(np.float16); Yl = (Y - Yh.astype(np.float32)).astype(np.float16)
    f = lambda a, b: (a.astype(np.float32) @ b.astype(np.float32))
    return ((f(Yh, Gh) + f(Yh, Gl)) + f(Yl, Gh)) / np.float32(sp)


def frames(x, t):
    return np.stack([x[i * H + 1: i * H + 1 + K] + x[i * H + L - 1: i * H + L - 1 - K: -1] for i in t])


rng = np.random.default_rng(1)
S = m.sample_count
tt = np.arange(S) / m.sample_rate
tone = np.sin(2 * np.pi * 1234.0 * tt)
cases = {"loud (0.4 tone + 0.1 noise)": 0.4 * tone + 0.1 * rng.standard_normal(S),
         "quiet on DC (0.3 + 1e-4 (tone + noise))": 0.3 + 1e-4 * (0.5 * tone + rng.standard_normal(S)),
         "quiet, no DC (1e-4 tone)": 1e-4 * tone}
t = np.arange(0, br.n_frames, 37)
ones = Gf.sum(axis=0)
print("max |T - T64| / max |T64| over %d frames x %d mels of branch 0, split-f16 product emulated in numpy" % (len(t), br.n_mels))
for name, x in cases.items():
    x = x.astype(np.float32)
    mn, mx = np.float32(x.min()), np.float32(x.max())
    a = np.float32(2.0) / ((mx - mn) + np.float32(m.norm_eps))
    xn64 = (x.astype(np.float64) - float(mn)) * float(a) - 1.0
    T64 = frames(xn64, t) @ Gf
    front = split_gemm(frames(((x - mn) * a - np.float32(1.0)).astype(np.float32), t).astype(np.float32))
    p = np.float32(2.0 ** -np.ceil(np.log2(max(abs(float(mn)), abs(float(mx))))))          # |p x| in (0.5, 1]
    raw = split_gemm(frames((x * p).astype(np.float32), t).astype(np.float32))
    behind = (a / p) * raw - np.float32(2.0) * (a * mn + np.float32(1.0)) * ones.astype(np.float32)
    scale = np.abs(T64).max()
    print("  %-42s in front %.2e   behind %.2e   (DC term / signal term: %.0f)" % (
        name, np.abs(front - T64).max() / scale, np.abs(behind - T64).max() / scale,
        abs(2.0 * (float(a) * float(mn) + 1.0)) * np.abs(ones).max() / scale))

"""Reads a model's spectrogram front-end OFF THE ONNX GRAPH (SURVEY.md section 7 hard part (ii), Appendix B) -- by probing.

How the published BirdNET / Perch files spell their front-end (an `STFT` node, a DFT written as a strided `Conv`, frames gathered
and multiplied by cos / sin matrices, the mel matrix separate or folded into the DFT weights, the exponent a constant or computed
from a learned scalar, ...) cannot be known offline.  So nothing here matches a spelling.  The nodes between the audio input and
the first 2-D convolution are run by `onnx_eval.Evaluator` (numpy, float64) on probe signals, and the parameters of the one
front-end shape the BHM1 container and the device kernels implement are fitted to the responses:

    x_n   = 2 ((x - min x) / (max x - min x + eps) - 0.5)                         (per segment)
    T_b   = frames(x_n; L_b, H_b) . G_b,      G_b = diag(hann_L) . cos(2 pi k n / L) . W_b   [L x n_mels]
    S_b   = scale_b (T_b^2)^expo_b + shift_b,  mel axis optionally reversed, stacked as channels [N, C, n_mels, n_frames]

  1. the spectrogram tensor = the data input of the first Conv with a 2-D kernel; the branch tensors T_b = the inputs of the
     first squaring nodes (`Mul(t, t)` / `Pow(t, c)`) on the way there;
  2. tail T_b -> S: constants fed as T_b give scale, shift, exponent (three values determine them, two more check the form);
     ramps fed as T_b give the axis order and the mel flip;
  3. eps: the same impulse on a signal of range 2 and of range 0.002;
  4. H_b: the last frame an impulse reaches bounds it, a shifted probe confirms it; G_b: H_b impulses (one per residue class of
     the frame step, extremes of the signal pinned so that the normalisation stays fixed) give every row;
  5. G_b is factored over the Hann-windowed cosines (least squares, DC row pinned to zero): a residual means the graph's window /
     transform is not the one the kernels fold, and the conversion is refused with that message;
  6. the sub-graph is re-run on random audio (loud, quiet with a DC offset) and compared with the closed form above evaluated
     from the fitted, float32-rounded parameters.

Anything the evaluator cannot run or the checks reject raises `RecoverError`; `convert.model_from_graph` then needs a front-end
manifest (a BHM1 file carrying the family's values) as before.  Offline the only graphs to try this on are the ones the tests
write (four spellings, tests/test_frontend_recover.py): the method is spelling-agnostic by construction, its operator coverage
against the real files is not verifiable here.

The sample rate is not in the graph (the reference takes it from the model type's config, src/inference/classifier.rs:360-377),
so it is an argument.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import modelfile as mf
from . import onnx_io as ox
from .onnx_eval import EvalError, Evaluator


class RecoverError(ValueError):
    pass


@dataclass
class Recovered:
    frontend: mf.Model            # no layers: family, rates, eps, branches, mel matrices in the blob
    spectrogram: str              # the tensor the conv stack starts from
    report: Dict[str, object] = field(default_factory=dict)


def hann_cos_operator(L: int) -> np.ndarray:
    """A[n][k] = hann_periodic[n] cos(2 pi k n / L), k = 0 .. L/2: what the device folds with the mel matrix (api.hip build_gf)."""
    n = np.arange(L)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / L)
    k = np.arange(L // 2 + 1)
    return w[:, None] * np.cos(2.0 * np.pi * ((n[:, None] * k[None, :]) % L) / L)


def closed_form_spectrogram(x: np.ndarray, eps: float, branches: List[mf.Branch], mel_w: List[np.ndarray]) -> np.ndarray:
    """[N, S] -> [N, C, n_mels, n_frames] in float64: the front-end of SURVEY.md Appendix B as the kernels compute it."""
    x = np.asarray(x, np.float64)
    mn, mx = x.min(axis=1, keepdims=True), x.max(axis=1, keepdims=True)
    xn = (x - mn) * (2.0 / ((mx - mn) + eps)) - 1.0
    out = []
    for b, w in zip(branches, mel_w):
        G = hann_cos_operator(b.frame_length) @ np.asarray(w, np.float64)
        fr = np.lib.stride_tricks.sliding_window_view(xn, b.frame_length, axis=1)[:, :: b.frame_step][:, : b.n_frames]
        T = fr @ G                                                      # [N, frames, mels]
        expo = 1.0 / (1.0 + math.exp(b.mag_scale))
        with np.errstate(divide="ignore"):
            o = np.where(T == 0.0, 0.0, np.exp(expo * np.log(T * T))) * b.out_scale + b.out_shift
        o = o.transpose(0, 2, 1)
        out.append(o[:, ::-1] if (b.flags & 1) else o)
    return np.stack(out, axis=1)


def _const_value(ev: Evaluator, name: str) -> Optional[np.ndarray]:
    try:
        return ev.run({}, [name])[0]
    except EvalError:
        return None


def recover_frontend(g: ox.Graph, sample_rate: int, family: int = 0, audio_input: Optional[str] = None,
                     probe_batch: int = 16) -> Recovered:
    ev = Evaluator(g)
    dyn_inputs = [v for v in g.inputs if v.name not in g.initializers]
    if not dyn_inputs:
        raise RecoverError("the graph has no input")
    vi = next((v for v in dyn_inputs if v.name == audio_input), None) if audio_input else dyn_inputs[0]
    if vi is None:
        raise RecoverError(f"no graph input named {audio_input!r}")
    tail_dims = list(vi.shape[1:])
    if not tail_dims or any(not isinstance(d, int) for d in tail_dims):
        raise RecoverError(f"input {vi.name!r} has shape {vi.shape}: the sample count must be static")
    big = [d for d in tail_dims if d > 1]
    if len(big) != 1:
        raise RecoverError(f"input {vi.name!r} has shape {vi.shape}: expected one sample axis")
    S = int(big[0])

    def feed(x2d: np.ndarray) -> Dict[str, np.ndarray]:
        return {vi.name: np.asarray(x2d, np.float64).reshape((x2d.shape[0],) + tuple(tail_dims))}

    # 1. the spectrogram tensor and the branch tensors ---------------------------------------------------------------------
    spec = None
    for n in g.nodes:
        if n.op_type == "Conv" and len(n.inputs) > 1 and n.inputs[1] in g.initializers:
            w = g.initializers[n.inputs[1]]
            if w.ndim == 4 and w.shape[2] > 1 and w.shape[3] > 1 and ev.depends_on(n.inputs[0], vi.name):
                spec = n.inputs[0]
                break
    if spec is None:
        raise RecoverError("no 2-D convolution downstream of the audio input: nowhere to enter the conv stack")
    front = ev.ancestors([spec], stop=[vi.name])
    squarers: List[int] = []
    for i in front:
        n = g.nodes[i]
        if n.op_type == "Mul" and len(n.inputs) == 2 and n.inputs[0] == n.inputs[1]:
            squarers.append(i)
        elif n.op_type == "Pow" and _const_value(ev, n.inputs[1]) is not None and np.asarray(_const_value(ev, n.inputs[1])).size == 1:
            squarers.append(i)
    first: List[str] = []
    for i in squarers:
        t = g.nodes[i].inputs[0]
        behind = set(ev.ancestors([t], stop=[vi.name]))
        if not any(j in behind for j in squarers) and t not in first and ev.depends_on(t, vi.name):
            first.append(t)
    if not first:
        raise RecoverError("no squaring node (Mul(t, t) / Pow(t, const)) between the audio input and the spectrogram")

    rng = np.random.default_rng(0xB1DA)
    x0 = rng.uniform(-0.7, 0.7, (1, S))
    try:
        ref = ev.run(feed(x0), [spec] + first)
    except EvalError as e:
        raise RecoverError(f"the front-end cannot be evaluated: {e}") from None
    spec0, t_shapes = ref[0], [r.shape for r in ref[1:]]
    if spec0.ndim != 4:
        raise RecoverError(f"spectrogram tensor {spec!r} has rank {spec0.ndim}")
    _, C, Hs, Ws = spec0.shape
    if len(first) != C:
        raise RecoverError(f"{len(first)} squared tensors feed a {C}-channel spectrogram: one branch per channel expected")

    def tail(values: List[np.ndarray]) -> np.ndarray:
        return ev.run({t: v for t, v in zip(first, values)}, [spec])[0][0]        # [C, H, W]

    def consts(v: float, only: Optional[int] = None, other: float = 1.0) -> List[np.ndarray]:
        return [np.full(s, v if (only is None or only == i) else other, np.float64) for i, s in enumerate(t_shapes)]

    # 2. the element-wise tail ---------------------------------------------------------------------------------------------
    f1 = tail(consts(1.0))
    chan_of: List[int] = []
    for b in range(C):
        d = np.abs(tail(consts(2.0, only=b)) - f1).reshape(C, -1).max(axis=1)
        hit = [c for c in range(C) if d[c] > 0]
        if len(hit) != 1:
            raise RecoverError(f"branch tensor {first[b]!r} reaches channels {hit}: not one channel per branch")
        chan_of.append(hit[0])
    if sorted(chan_of) != list(range(C)):
        raise RecoverError(f"branches map to channels {chan_of}")
    fv = {v: tail(consts(v)) for v in (0.5, 1.0, 2.0, 3.0, -1.0)}
    tails: List[Tuple[float, float, float]] = []   # (expo, scale, shift) per branch
    for b in range(C):
        c = chan_of[b]
        vals = {}
        for v, arr in fv.items():
            a = arr[c]
            if not np.all(np.isfinite(a)) or np.ptp(a) > 1e-9 * max(1.0, float(np.abs(a).max())):
                raise RecoverError(f"channel {c}: the tail after the squaring is not one scalar function for the whole branch "
                                   "(per-mel affine / normalisation layers are not representable in the container)")
            vals[v] = float(a.reshape(-1)[0])
        den = vals[1.0] - vals[0.5]
        if den == 0.0 or (vals[2.0] - vals[1.0]) / den <= 0.0:
            raise RecoverError(f"channel {c}: the tail does not depend on the squared value")
        expo = math.log((vals[2.0] - vals[1.0]) / den, 4.0)
        scale = (vals[2.0] - vals[1.0]) / (4.0 ** expo - 1.0)
        shift = vals[1.0] - scale
        tol = 1e-9 * max(1.0, abs(scale), abs(shift))
        if abs(scale * 9.0 ** expo + shift - vals[3.0]) > tol * 10 or abs(vals[-1.0] - vals[1.0]) > tol:
            raise RecoverError(f"channel {c}: the tail is not scale * (t^2)^p + shift (an even power law)")
        if not (0.0 < expo < 1.0):
            raise RecoverError(f"channel {c}: exponent {expo} is outside (0, 1), not 1 / (1 + exp(mag_scale))")
        tails.append((expo, scale, shift))

    # axis order and mel flip: a ramp along one axis of T_b must come out along H (mel) or W (time) of its channel
    axes: List[Tuple[int, int, bool]] = []   # (mel axis, time axis, flip) per branch, axes of T_b
    for b in range(C):
        shp = t_shapes[b]
        var = [a for a, d in enumerate(shp) if d > 1]
        if len(var) != 2:
            raise RecoverError(f"branch tensor {first[b]!r} has shape {shp}: expected a (mel, time) matrix per segment")
        expo, scale, shift = tails[b]
        mel_ax = time_ax = None
        flip = False
        for a in var:
            ramp = 1.0 + np.arange(shp[a], dtype=np.float64) / shp[a]
            vals = consts(1.0)
            vals[b] = np.broadcast_to(ramp.reshape([-1 if i == a else 1 for i in range(len(shp))]), shp).copy()
            out = tail(vals)[chan_of[b]]                                   # [H, W]
            want = scale * (ramp * ramp) ** expo + shift
            tol = 1e-9 * max(1.0, float(np.abs(want).max()))
            if shp[a] == Hs and np.ptp(out, axis=1).max() <= tol and np.abs(out[:, 0] - want).max() <= tol:
                mel_ax = a
            elif shp[a] == Hs and np.ptp(out, axis=1).max() <= tol and np.abs(out[::-1, 0] - want).max() <= tol:
                mel_ax, flip = a, True
            elif shp[a] == Ws and np.ptp(out, axis=0).max() <= tol and np.abs(out[0, :] - want).max() <= tol:
                time_ax = a
            else:
                raise RecoverError(f"branch {b}: axis {a} of {first[b]!r} does not map onto the mel or the time axis of the spectrogram")
        if mel_ax is None or time_ax is None:
            raise RecoverError(f"branch {b}: could not tell the mel axis from the time axis")
        axes.append((mel_ax, time_ax, flip))

    def branch_matrix(t: np.ndarray, b: int) -> np.ndarray:
        """T_b of a batch -> [N, frames, mels]"""
        mel_ax, time_ax, _ = axes[b]
        keep = [0, time_ax, mel_ax]
        t = np.transpose(t, keep + [a for a in range(t.ndim) if a not in keep])
        return t.reshape(t.shape[0], t.shape[1], t.shape[2])

    # 3 + 4. the linear part, through the normalisation ----------------------------------------------------------------------
    # probe signals: zero, extremes pinned at the first two samples (min = -1, max = +1 whatever else the probe holds), impulses
    # of 0.5: T is affine in the signal while min / max do not move, so differences against the base response are exact
    U = 0.5
    base = np.zeros(S)
    base[0], base[1] = -1.0, 1.0

    def responses(b: int, positions: List[int], scale_sig: float = 1.0) -> np.ndarray:
        """delta T_b [len(positions), frames, mels] for impulses at `positions` (signal scaled by scale_sig as a whole)"""
        outs = []
        t0 = branch_matrix(ev.run(feed((base * scale_sig)[None]), [first[b]])[0], b)[0]
        for i in range(0, len(positions), probe_batch):
            ps = positions[i:i + probe_batch]
            x = np.tile(base, (len(ps), 1))
            for r, p in enumerate(ps):
                x[r, p] += U
            outs.append(branch_matrix(ev.run(feed(x * scale_sig), [first[b]])[0], b) - t0[None])
        return np.concatenate(outs, axis=0)

    eps_est: List[float] = []
    branches: List[mf.Branch] = []
    mel_ws: List[np.ndarray] = []
    report: Dict[str, object] = {"spectrogram": spec, "branch_tensors": list(first), "channels": chan_of}
    for b in range(C):
        n_frames, n_mels = Ws, Hs
        # the last frame an impulse reaches: t_hi = floor(p / H) (three neighbouring positions: a Hann window's first row is zero)
        p0 = (S * 3) // 4
        d = responses(b, [p0, p0 + 1, p0 + 2])
        mag = np.abs(d).max(axis=2)                                          # [3, frames]
        live = mag > 1e-13 * max(float(mag.max()), 1e-300)
        if not live.any():
            raise RecoverError(f"branch {b}: an impulse at sample {p0} does not reach {first[b]!r}")
        t_hi = max(int(np.nonzero(live[r])[0].max()) for r in range(3) if live[r].any())
        if t_hi < 1:
            raise RecoverError(f"branch {b}: fewer than two frames")
        lo_h, hi_h = p0 / (t_hi + 1.0), (p0 + 2.0) / t_hi
        cands = [h for h in range(max(1, int(math.floor(lo_h))), int(math.ceil(hi_h)) + 1)]
        H = None
        # (TWO impulse positions, 37 samples apart: the operator is even about L / 2, so one impulse landing on row L / 2 + j passes
        #  a wrong step H - 2 j as well -- onnx_frontend.hpp, seeded plan 158)
        pm, pm2 = S // 2, S // 2 + 37
        dm, dm2 = responses(b, [pm, pm2])
        for h in cands:
            if pm2 + h >= S:
                continue
            dh, dh2 = responses(b, [pm + h, pm2 + h])
            top = max(float(np.abs(dm).max()), float(np.abs(dm2).max()), 1e-300)
            if max(np.abs(dh[1:] - dm[:-1]).max(), np.abs(dh2[1:] - dm2[:-1]).max()) <= 1e-11 * top and np.abs(dm).max() > 0:
                H = h
                break
        if H is None:
            raise RecoverError(f"branch {b}: no frame step in {cands} makes the response shift-invariant")
        if n_frames * H - H >= S:
            raise RecoverError(f"branch {b}: {n_frames} frames of step {H} do not fit {S} samples")
        l_max = S - (n_frames - 1) * H                                       # n_frames = (S - L) / H + 1 rounded down
        l_min = max(S - n_frames * H + 1, 1)
        # eps from the same impulse on a signal 1000 x smaller: delta T = 2 s u / (2 s + eps) . G[row]
        s_small = 1e-3
        ds = responses(b, [pm], scale_sig=s_small)[0]
        sel = np.abs(dm) > 0.1 * np.abs(dm).max()
        r = float(np.median(dm[sel] / ds[sel]))
        if abs(r * s_small - 1.0) < 1e-9:
            raise RecoverError("the front-end does not normalise by the segment's range (min / max): not the container's front-end")
        eps = 2.0 * s_small * (1.0 - r) / (r * s_small - 1.0)
        eps = 0.0 if abs(eps) < 1e-12 else eps
        if eps < 0 or eps > 1e-2:
            raise RecoverError(f"normalisation epsilon {eps} is not plausible")
        eps_est.append(eps)
        kappa = 2.0 / (2.0 + eps)
        # every row of G: impulses at t0 H + r, r = 0 .. H - 1; frame t sees row p - t H
        t0 = min(n_frames - 1, (l_max + H - 1) // H + 1)
        if t0 * H + H - 1 >= S:
            t0 = (S - H) // H
        pos = [t0 * H + r for r in range(H)]
        d = responses(b, pos) / (kappa * U)                                  # [H, frames, mels]
        G = np.zeros((l_max + H, n_mels))
        seen = np.zeros(l_max + H, bool)
        for ri, p in enumerate(pos):
            for t in range(n_frames):
                nrow = p - t * H
                if 0 <= nrow < G.shape[0]:
                    G[nrow] = d[ri, t]
                    seen[nrow] = True
        rowmag = np.abs(G).max(axis=1)
        nz = np.nonzero(rowmag > 1e-13 * rowmag.max())[0]
        L = int(nz.max()) + 1
        if not seen[:L].all() or not (l_min <= L <= l_max):
            raise RecoverError(f"branch {b}: support of the frame operator ends at {L}, outside [{l_min}, {l_max}] implied by {n_frames} frames")
        if L % 2:
            raise RecoverError(f"branch {b}: odd frame length {L}")
        G = G[:L]
        # 5. G = diag(hann) . cos . W with W[0] = 0
        A = hann_cos_operator(L)
        W1, *_ = np.linalg.lstsq(A[:, 1:], G, rcond=None)
        W = np.concatenate([np.zeros((1, n_mels)), W1], axis=0)
        resid = float(np.abs(A @ W - G).max() / max(np.abs(G).max(), 1e-300))
        if resid > 1e-6:   # (float32 operator weights in the graph leave ~1e-8)
            raise RecoverError(f"branch {b}: the frame operator is not a Hann-windowed real DFT followed by a mel matrix "
                               f"(relative residual {resid:.2e}): window or transform differ from what the kernels fold")
        expo, scale, shift = tails[b]
        br = mf.Branch(L, H, n_mels, n_frames, 0.0, 0.0, float(math.log(1.0 / expo - 1.0)), float(scale), float(shift),
                       1 if axes[b][2] else 0, 0)
        # (fmin / fmax are informational in the container: the band the matrix covers, from its non-zero rows)
        rows = np.nonzero(np.abs(W).max(axis=1) > 1e-5 * np.abs(W).max())[0]
        if rows.size:
            br.fmin = float(max(rows.min() - 1, 0) * sample_rate / L)
            br.fmax = float(min(rows.max() + 1, L // 2) * sample_rate / L)
        order = chan_of[b]
        branches.append((order, br))
        mel_ws.append((order, W.astype(np.float32)))
        report[f"branch{b}"] = {"L": L, "H": H, "expo": expo, "scale": scale, "shift": shift, "flip": axes[b][2],
                                "operator_residual": resid}
    if max(eps_est) - min(eps_est) > 1e-9:
        raise RecoverError(f"branches disagree on the normalisation epsilon: {eps_est}")
    eps = float(np.mean(eps_est))
    branches = [br for _, br in sorted(branches, key=lambda t: t[0])]
    mel_ws = [w for _, w in sorted(mel_ws, key=lambda t: t[0])]

    # the container: mel matrices at the start of the blob, 64-byte aligned
    chunks, off = [], 0
    for br, w in zip(branches, mel_ws):
        pad = (-off) % 16
        if pad:
            chunks.append(np.zeros(pad, np.float32))
            off += pad
        br.mel_w_off = off
        chunks.append(w.reshape(-1))
        off += w.size
    fe = mf.Model(family, int(sample_rate), S, S / float(sample_rate), 0, 0, mf.OUT_NONE, 0, Hs, Ws, np.float32(eps).item(),
                  branches, [], np.concatenate(chunks))

    # 6. the whole sub-graph against the closed form, on signals it has not seen
    xs = np.stack([rng.uniform(-0.9, 0.9, S), 0.31 + 0.004 * rng.standard_normal(S)])
    got = ev.run(feed(xs), [spec])[0]
    want = closed_form_spectrogram(xs, fe.norm_eps, [mf.Branch(**{**br.__dict__, "mag_scale": float(np.float32(br.mag_scale)),
                                                                  "out_scale": float(np.float32(br.out_scale)),
                                                                  "out_shift": float(np.float32(br.out_shift))}) for br in branches], mel_ws)
    err = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-300))
    report["verification_max_rel_err"] = err
    report["norm_eps"] = eps
    if not np.isfinite(err) or err > 2e-5:
        # (|v|^(2 expo) is not Lipschitz at 0: where a mel projection all but vanishes the float32 rounding of the fitted matrix
        #  moves the pixel by |dv|^(2 expo); such a front-end is held to the graph in front of the power law instead -- as
        #  onnx_frontend.hpp does, same tolerance)
        lin = 0.0
        for c, br in enumerate(branches):
            expo = 1.0 / (1.0 + math.exp(float(np.float32(br.mag_scale))))
            sc, sh = float(np.float32(br.out_scale)), float(np.float32(br.out_shift))
            ua = np.maximum((got[:, c] - sh) / sc, 0.0) ** (0.5 / expo)
            ub = np.maximum((want[:, c] - sh) / sc, 0.0) ** (0.5 / expo)
            lin = max(lin, float(np.abs(ua - ub).max() / max(ub.max(), 1e-300)))
        report["verification_rel_err_before_power_law"] = lin
        if not np.isfinite(err) or not np.isfinite(lin) or lin > 1e-5:
            raise RecoverError(f"recovered front-end differs from the graph on random audio (relative error {err:.2e}, {lin:.2e} in front of the power law)")
    return Recovered(fe, spec, report)

"""The library reads the spectrogram front-end OFF THE `.onnx` GRAPH itself (birda_amd/csrc/onnx_frontend.hpp).

The reference hands ONNX Runtime the model file (src/inference/classifier.rs:269-283) and whatever front-end that graph holds is
what runs: `mag_scale` is a learned scalar, the band edges, the affine and the mel matrices are the
model's.  Round 4's native route skipped those nodes and asserted a family table -- a trained file whose constants differ was
answered wrongly without a word.  Now the nodes are run by a float64 evaluator inside the library on probe signals, the container's
parameters are fitted to the responses and verified against the closed form the kernels compute, or the file is refused.

Offline the graphs are the ones `convert.frontend_nodes` writes, in four deliberately different spellings, from front-ends that
DIFFER from the family table (mag_scale 0.9, affine (1.3, 0.2), fmax 2 800 Hz); the Python recovery (birda_amd/frontend_recover.py)
is the second witness.  CPU tests go through bh_onnx_to_bhm / bh_onnx_eval (host only); the GPU test compares logits.
"""
import ctypes as C
import os
import time

import numpy as np
import pytest

from birda_amd import _lib, convert, modelfile as mf, onnx_io as ox, synth

SPELLINGS = ("conv1d", "stft", "complex", "fused")


def retrained(kind, mag_scale=0.9, scale=1.3, shift=0.2, fmax0=2800.0):
    """a model whose front-end is NOT the family table's: another learned exponent, another spectrogram affine, another band"""
    m = synth.build_model(kind)
    blob = m.blob.copy()
    for i, b in enumerate(m.branches):
        b.mag_scale, b.out_scale, b.out_shift = mag_scale, scale, shift
        if i == 0:
            b.fmax = fmax0
            w = synth.linear_to_mel_weight_matrix(b.n_mels, b.n_bins, m.sample_rate, b.fmin, b.fmax)
            blob[b.mel_w_off:b.mel_w_off + w.size] = w.reshape(-1)
    m.blob = blob
    return m


def _mel(m, b):
    return np.asarray(m.blob[b.mel_w_off:b.mel_w_off + b.n_bins * b.n_mels]).reshape(b.n_bins, b.n_mels)


def _native(g, tmp_path, name="m"):
    L = _lib.load()
    p, q = str(tmp_path / (name + ".onnx")), str(tmp_path / (name + ".bhm"))
    with open(p, "wb") as f:
        f.write(ox.dump(g))
    rc = L.bh_onnx_to_bhm(p.encode(), q.encode())
    return rc, L.bh_last_error().decode(), (mf.read_model(q) if rc == 0 else None)


def _same_front_end(m, m2, spelling):
    assert (m2.sample_rate, m2.sample_count, m2.spec_h, m2.spec_w, len(m2.branches)) == (m.sample_rate, m.sample_count, m.spec_h, m.spec_w, len(m.branches))
    assert m2.norm_eps == pytest.approx(m.norm_eps, rel=1e-5)
    for a, b in zip(m.branches, m2.branches):
        assert (a.frame_length, a.frame_step, a.n_mels, a.n_frames) == (b.frame_length, b.frame_step, b.n_mels, b.n_frames)
        assert b.mag_scale == pytest.approx(a.mag_scale, abs=1e-5)
        assert (b.out_scale, b.out_shift) == pytest.approx((a.out_scale, a.out_shift), rel=1e-6)
        wa, wb = _mel(m, a), _mel(m2, b)
        if spelling == "fused":     # the flip lives in the operator there: same front-end, mel columns reversed, no flip flag
            assert b.flags & 1 == 0
            wb = wb[:, ::-1]
        else:
            assert b.flags & 1 == a.flags & 1
        assert np.abs(wa - wb).max() < 1e-6
        assert b.fmin <= max(a.fmin, 1.0) + m.sample_rate / a.frame_length and b.fmax >= a.fmax - m.sample_rate / a.frame_length


@pytest.mark.parametrize("spelling", SPELLINGS)
def test_front_end_is_read_off_the_graph_whatever_the_spelling(spelling, tmp_path):
    from oracle import oracle as O
    m = retrained("mini")
    g = convert.graph_from_model(m, frontend_spelling=spelling)
    rc, msg, m2 = _native(g, tmp_path)
    assert rc == 0, msg
    _same_front_end(m, m2, spelling)
    assert [(L.op, L.act, L.res_tensor) for L in m2.layers] == [(L.op, L.act, L.res_tensor) for L in m.layers]
    # the second witness: the Python recovery on the same bytes (numpy least squares against this closed-form factorisation)
    py = convert.model_from_graph(ox.load(ox.dump(g)), None, sample_rate=m.sample_rate)
    for a, b in zip(py.branches, m2.branches):
        assert (a.frame_length, a.frame_step, a.n_frames, a.flags) == (b.frame_length, b.frame_step, b.n_frames, b.flags)
        assert np.float32(a.mag_scale) == pytest.approx(np.float32(b.mag_scale), abs=1e-6) and a.out_scale == pytest.approx(b.out_scale, rel=1e-6)
        assert np.abs(_mel(py, a) - _mel(m2, b)).max() < 1e-6
    # ... and the oracle gives the converted container the logits of the model the graph was written from
    pa, pb = str(tmp_path / "a.bhm"), str(tmp_path / "b.bhm")
    mf.write_model(pa, m)
    mf.write_model(pb, m2)
    segs = synth.synth_segments(3, m.sample_count, m.sample_rate, start=5)
    la, lb = O.OracleModel(pa).forward(segs), O.OracleModel(pb).forward(segs)
    assert np.abs(la - lb).max() <= 2e-5 * max(1.0, np.abs(la).max())
    # the family table's values would NOT have given them (what round 4 did silently)
    mt = synth.build_model("mini")
    pt = str(tmp_path / "t.bhm")
    mf.write_model(pt, mt)
    assert np.abs(O.OracleModel(pt).forward(segs) - la).max() > 1e-3 * np.abs(la).max()


@pytest.mark.parametrize("j,L,H", [(2, 512, 261), (-2, 448, 321), (-1, 512, 383), (1, 448, 275), (2, 640, 334)])
def test_frame_step_is_not_mistaken_when_the_probe_lands_beside_the_window_centre(j, L, H, tmp_path):
    """A Hann-windowed cosine operator is even about L / 2: an impulse that lands on row L / 2 + j of a frame gives, one frame on and
    2 j samples short of the true step, the same numbers (row L / 2 - j).  With ONE probe position the recovery read seeded random
    plan 158 (L 512, H 261 over 12 000 samples: the probe at S / 2 lands on row 258) as H = 257 and then refused the file for a
    front-end it is (found by a soak run of tests/test_random_plans_gpu.py's checks over seeds 100-180, round 6).  Both readers now
    probe two positions 37 samples apart.  Here: frame steps chosen so that the probe at S / 2 lands on row L / 2 + j."""
    S = 12000
    assert (S // 2) % H == L // 2 + j
    plan = dict(sr=48000, n=S, branches=[(L, H, 22, 0.0, 3000.0)], stem=16, stages=[(1, 3, 1, 16, 1)], head=32, classes=10)
    m = synth.build_model("custom", plan=plan)
    assert m.branches[0].frame_step == H
    for spelling in ("complex", "conv1d"):
        g = convert.graph_from_model(m, frontend_spelling=spelling)
        rc, msg, m2 = _native(g, tmp_path, name=f"s{L}_{H}{spelling}")
        assert rc == 0, msg
        _same_front_end(m, m2, spelling)
        py = convert.model_from_graph(ox.load(ox.dump(g)), None, sample_rate=m.sample_rate)
        assert (py.branches[0].frame_length, py.branches[0].frame_step) == (L, H)


@pytest.mark.parametrize("kind,spelling", [("birdnet_v24_tiny", "conv1d"), ("birdnet_v24_tiny", "stft"), ("perch_v2_tiny", "complex")])
def test_full_size_front_ends_are_read_in_seconds(kind, spelling, tmp_path):
    """BirdNET v2.4's two branches (2 048 / 278 and 1 024 / 280 over 144 000 samples) and the 128-mel 32 kHz branch: six probe rows
    per branch instead of one per residue of the hop, a blocked multi-threaded float64 GEMM under the DFT-as-Conv spelling"""
    m = retrained(kind) if kind.startswith("birdnet") else retrained(kind, fmax0=15000.0)
    g = convert.graph_from_model(m, frontend_spelling=spelling)
    t = time.perf_counter()
    rc, msg, m2 = _native(g, tmp_path)
    took = time.perf_counter() - t
    assert rc == 0, msg
    _same_front_end(m, m2, spelling)
    assert took < 60.0, took        # (2-3 s on this container's 8 cores; the bound only catches a lost GEMM path)


def test_front_ends_the_kernels_cannot_express_are_refused_not_assumed(tmp_path):
    """every refusal of tests/test_frontend_recover.py, through the library: BH_ERR_UNSUPPORTED (-6) and a message that names the
    property or the operator"""
    m = synth.build_model("mini")

    def graph(spelling="conv1d"):
        return convert.graph_from_model(m, frontend_spelling=spelling)

    # a Hamming window: the frame operator is no longer in the span of the Hann-windowed cosines
    g = graph()
    for b, br in enumerate(m.branches):
        L = br.frame_length
        n = np.arange(L)
        ham = 0.54 - 0.46 * np.cos(2 * np.pi * n / L)
        ang = 2 * np.pi * ((n[None, :] * np.arange(br.n_bins)[:, None]) % L) / L
        g.initializers[f"fe{b}_dft"] = (ham[None, :] * np.cos(ang))[:, None, :].astype(np.float32)
    rc, msg, _ = _native(g, tmp_path, "hamming")
    assert rc == -6 and "Hann-windowed" in msg and "refused rather than assumed" in msg
    # a per-mel affine after the power law
    g = graph()
    g.initializers["fe0_sc"] = np.linspace(0.5, 1.0, m.branches[0].n_mels).astype(np.float32)
    rc, msg, _ = _native(g, tmp_path, "permel")
    assert rc == -6 and "one scalar function" in msg
    # a magnitude spectrogram (re^2 + im^2): linear in front of the square, but not a real DFT
    g = graph("complex")
    for b, br in enumerate(m.branches):
        g.initializers[f"fe{b}_s1"] = np.asarray([2 * br.n_bins], np.int64)
        mel = g.initializers[f"fe{b}_mel"]
        g.initializers[f"fe{b}_mel"] = np.concatenate([mel, mel], axis=1)
    rc, msg, _ = _native(g, tmp_path, "magnitude")
    assert rc == -6 and "Hann-windowed" in msg
    # no min / max normalisation at all
    g = graph()
    next(n for n in g.nodes if n.outputs[0] == "fe0_sig").inputs[0] = "audio"
    next(n for n in g.nodes if n.outputs[0] == "fe1_sig").inputs[0] = "audio"
    rc, msg, _ = _native(g, tmp_path, "nonorm")
    assert rc == -6 and "range" in msg
    # an operator outside the evaluator's set: named
    g = graph()
    k = next(i for i, n in enumerate(g.nodes) if n.outputs[0] == "fe_xn")
    g.nodes.insert(k + 1, ox.Node("Loop", ["fe_xn"], ["fe_loop"]))
    for n in g.nodes:
        if n.op_type == "Unsqueeze" and n.inputs[0] == "fe_xn":
            n.inputs[0] = "fe_loop"
    rc, msg, _ = _native(g, tmp_path, "loop")
    assert rc == -6 and "operator Loop" in msg
    # a learned exponent outside (0, 1) -- e.g. a log-mel front-end -- is not the power law the kernels compute
    g = graph("conv1d")
    pw = [k for k in g.initializers if k.endswith("_ex")]
    assert pw, sorted(g.initializers)[:40]
    for k in pw:
        g.initializers[k] = np.asarray(1.5, np.float32).reshape(g.initializers[k].shape)
    rc, msg, _ = _native(g, tmp_path, "expo")
    assert rc == -6 and "exponent" in msg


def _eval(nodes, inits, feed_name, feed, target, tmp_path):
    L = _lib.load()
    g = ox.Graph(nodes=nodes, initializers=inits)
    p = str(tmp_path / "g.onnx")
    with open(p, "wb") as f:
        f.write(ox.dump(g))
    dims = (C.c_int64 * 8)()
    rank = C.c_uint32()
    cap = 1 << 20
    out = np.zeros(cap, np.float64)
    if feed is not None:
        feed = np.ascontiguousarray(feed, np.float64)
        fd = (C.c_int64 * max(1, feed.ndim))(*feed.shape)
        rc = L.bh_onnx_eval(p.encode(), feed_name.encode(), feed.ctypes.data_as(C.POINTER(C.c_double)), fd, feed.ndim, target.encode(),
                            out.ctypes.data_as(C.POINTER(C.c_double)), cap, dims, C.byref(rank))
    else:
        rc = L.bh_onnx_eval(p.encode(), None, None, None, 0, target.encode(), out.ctypes.data_as(C.POINTER(C.c_double)), cap, dims, C.byref(rank))
    if rc != 0:
        raise RuntimeError(L.bh_last_error().decode())
    shape = tuple(dims[i] for i in range(rank.value))
    return out[:int(np.prod(shape, dtype=np.int64))].reshape(shape)


def test_evaluator_operators_against_torch_and_numpy(tmp_path):
    """the C++ evaluator's operators with their corner semantics (the same cases tests/test_frontend_recover.py holds the Python
    evaluator to)"""
    import torch
    F = torch.nn.functional
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 3, 50))
    w = rng.standard_normal((4, 3, 7)).astype(np.float32)
    b = rng.standard_normal(4).astype(np.float32)
    got = _eval([ox.Node("Conv", ["x", "w", "b"], ["y"], {"strides": [3], "pads": [2, 1]})], {"w": w, "b": b}, "x", x, "y", tmp_path)
    want = F.conv1d(F.pad(torch.from_numpy(x), (2, 1)), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=3).numpy()
    assert got.shape == want.shape and np.allclose(got, want, atol=1e-12)
    x2 = rng.standard_normal((1, 4, 9, 11))
    w2 = rng.standard_normal((4, 1, 3, 3)).astype(np.float32)
    got = _eval([ox.Node("Conv", ["x", "w"], ["y"], {"strides": [2, 1], "group": 4, "auto_pad": "SAME_UPPER"})], {"w": w2}, "x", x2, "y", tmp_path)
    want = F.conv2d(F.pad(torch.from_numpy(x2), (1, 1, 1, 1)), torch.from_numpy(w2).double(), stride=(2, 1), groups=4).numpy()
    assert got.shape == want.shape and np.allclose(got, want, atol=1e-12)
    w3 = rng.standard_normal((5, 4, 3, 2)).astype(np.float32)
    got = _eval([ox.Node("Conv", ["x", "w"], ["y"], {"strides": [1, 2], "pads": [1, 0, 2, 1]})], {"w": w3}, "x", x2, "y", tmp_path)
    want = F.conv2d(F.pad(torch.from_numpy(x2), (0, 1, 1, 2)), torch.from_numpy(w3).double(), stride=(1, 2)).numpy()
    assert got.shape == want.shape and np.allclose(got, want, atol=1e-12)
    # STFT against torch.stft (no centring, one-sided): a power-of-two frame (the FFT path) and an odd one (the direct DFT)
    sig = rng.standard_normal((2, 400))
    for L in (64, 50):
        win = np.hanning(L + 1)[:L].astype(np.float32)
        got = _eval([ox.Node("STFT", ["s", "step", "win", "len"], ["y"], {"onesided": 1})],
                    {"step": np.asarray(10, np.int64), "win": win, "len": np.asarray(L, np.int64)}, "s", sig[:, :, None], "y", tmp_path)
        ref = torch.stft(torch.from_numpy(sig), L, hop_length=10, window=torch.from_numpy(win).double(), center=False, return_complex=True)
        assert np.allclose(got[..., 0], ref.real.numpy().transpose(0, 2, 1), atol=1e-10)
        assert np.allclose(got[..., 1], ref.imag.numpy().transpose(0, 2, 1), atol=1e-10)
    # MatMul with broadcast batch axes, Gemm with transB / alpha / beta
    a = rng.standard_normal((3, 1, 5, 6))
    bm = rng.standard_normal((4, 6, 7)).astype(np.float32)
    got = _eval([ox.Node("MatMul", ["a", "b"], ["y"])], {"b": bm}, "a", a, "y", tmp_path)
    assert np.allclose(got, np.matmul(a, bm.astype(np.float64)), atol=1e-12)
    a2 = rng.standard_normal((5, 6))
    wg, bg = rng.standard_normal((7, 6)).astype(np.float32), rng.standard_normal(7).astype(np.float32)
    got = _eval([ox.Node("Gemm", ["a", "w", "b"], ["y"], {"transB": 1, "alpha": 0.5, "beta": 2.0})], {"w": wg, "b": bg}, "a", a2, "y", tmp_path)
    assert np.allclose(got, 0.5 * a2 @ wg.T.astype(np.float64) + 2.0 * bg, atol=1e-12)
    # shape operators with their corner semantics
    a = rng.standard_normal((2, 5, 6))
    rev = _eval([ox.Node("Slice", ["a", "s", "e", "ax", "st"], ["y"])],
                {"s": np.asarray([-1], np.int64), "e": np.asarray([-(2 ** 62)], np.int64), "ax": np.asarray([1], np.int64),
                 "st": np.asarray([-1], np.int64)}, "a", a, "y", tmp_path)
    assert np.array_equal(rev, a[:, ::-1])
    sl = _eval([ox.Node("Slice", ["a", "s", "e", "ax", "st"], ["y"])],
               {"s": np.asarray([1, 0], np.int64), "e": np.asarray([2 ** 62, -1], np.int64), "ax": np.asarray([2, 1], np.int64),
                "st": np.asarray([2, 3], np.int64)}, "a", a, "y", tmp_path)
    assert np.array_equal(sl, a[:, 0:-1:3, 1::2])
    rs = _eval([ox.Node("Reshape", ["a", "shape"], ["y"])], {"shape": np.asarray([0, -1], np.int64)}, "a", a, "y", tmp_path)
    assert rs.shape == (2, 30) and np.array_equal(rs, a.reshape(2, 30))
    us = _eval([ox.Node("Unsqueeze", ["a", "ax"], ["y"])], {"ax": np.asarray([0, 4], np.int64)}, "a", a, "y", tmp_path)
    assert us.shape == (1, 2, 5, 6, 1)
    mm = _eval([ox.Node("ReduceMax", ["a"], ["y"], {"axes": [1, 2], "keepdims": 0})], {}, "a", a, "y", tmp_path)
    assert np.array_equal(mm, a.max(axis=(1, 2)))
    mn = _eval([ox.Node("ReduceMean", ["a", "ax"], ["y"], {"keepdims": 1})], {"ax": np.asarray([-1], np.int64)}, "a", a, "y", tmp_path)
    assert np.allclose(mn, a.mean(axis=-1, keepdims=True), atol=1e-15)
    tr = _eval([ox.Node("Transpose", ["a"], ["y"], {"perm": [2, 0, 1]})], {}, "a", a, "y", tmp_path)
    assert np.array_equal(tr, a.transpose(2, 0, 1))
    ga = _eval([ox.Node("Gather", ["a", "i"], ["y"], {"axis": 2})], {"i": np.asarray([[5, 0], [-1, 2]], np.int64)}, "a", a, "y", tmp_path)
    assert np.array_equal(ga, np.take(a, np.asarray([[5, 0], [5, 2]]), axis=2))
    cc = _eval([ox.Node("Concat", ["a", "c"], ["y"], {"axis": 1})], {"c": np.ones((2, 2, 6), np.float32)}, "a", a, "y", tmp_path)
    assert np.array_equal(cc, np.concatenate([a, np.ones((2, 2, 6))], axis=1))
    pd = _eval([ox.Node("Pad", ["a", "p", "v"], ["y"])], {"p": np.asarray([0, 1, 2, 0, 0, 3], np.int64), "v": np.asarray(7.0, np.float32)}, "a", a, "y", tmp_path)
    assert np.array_equal(pd, np.pad(a, [(0, 0), (1, 0), (2, 3)], constant_values=7.0))
    ex = _eval([ox.Node("Expand", ["a", "s"], ["y"])], {"s": np.asarray([3, 2, 5, 6], np.int64)}, "a", a, "y", tmp_path)
    assert np.array_equal(ex, np.broadcast_to(a, (3, 2, 5, 6)))
    bn = [rng.uniform(0.5, 1.5, 5).astype(np.float32), rng.normal(0, 1, 5).astype(np.float32), rng.normal(0, 1, 5).astype(np.float32),
          rng.uniform(0.5, 1.5, 5).astype(np.float32)]
    got = _eval([ox.Node("BatchNormalization", ["a", "g", "b", "m", "v"], ["y"], {"epsilon": 1e-3})], dict(zip("gbmv", bn)), "a", a, "y", tmp_path)
    sh = (1, 5, 1)
    assert np.allclose(got, (a - bn[2].reshape(sh)) / np.sqrt(bn[3].reshape(sh).astype(np.float64) + np.float64(np.float32(1e-3))) * bn[0].reshape(sh) + bn[1].reshape(sh), atol=1e-12)
    # integer shape arithmetic stays integer: Shape -> Gather -> Mul -> Concat -> Reshape
    nodes = [ox.Node("Shape", ["a"], ["sh"]), ox.Node("Gather", ["sh", "i1"], ["d1"], {"axis": 0}), ox.Node("Mul", ["d1", "two"], ["d2"]),
             ox.Node("Concat", ["m1", "d2"], ["shape"], {"axis": 0}), ox.Node("Reshape", ["a", "shape"], ["y"])]
    got = _eval(nodes, {"i1": np.asarray([1], np.int64), "two": np.asarray([2], np.int64), "m1": np.asarray([-1], np.int64)}, "a", a, "y", tmp_path)
    assert got.shape == (6, 10)
    # an intermediate tensor can be fed: only what lies below it is computed; an operator outside the set is named
    g_nodes = [ox.Node("Exp", ["a"], ["b"]), ox.Node("Unknown", ["b"], ["c"]), ox.Node("Neg", ["c"], ["d"])]
    assert np.array_equal(_eval(g_nodes, {}, "c", np.ones(3), "d", tmp_path), -np.ones(3))
    with pytest.raises(RuntimeError, match="Unknown"):
        _eval(g_nodes, {}, "a", np.ones(3), "d", tmp_path)
    # float64 and scalar-attribute constants are read
    nodes = [ox.Node("Constant", [], ["k"], {"value_float": 2.5}), ox.Node("Mul", ["a", "k"], ["y"])]
    assert np.array_equal(_eval(nodes, {}, "a", a, "y", tmp_path), a * 2.5)


@pytest.mark.gpu
@pytest.mark.parametrize("spelling", ["stft", "conv1d"])
def test_classifier_created_on_a_retrained_onnx_file_gives_its_own_logits(tmp_path, spelling, oracle_lib):
    """VERDICT r4 next #1's bar: birdnet_v24 written with mag_scale = 0.9, affine (1.3, 0.2) and fmax = 2 800 Hz, in two spellings
    (the STFT operator; a DFT written as a strided Conv) -> bh_classifier_create("x.onnx") gives the logits of the BHM1 container
    written from the same model, within 2e-5 of the logit scale (round 4: the family table's front-end, silently)."""
    from birda_amd.classifier import BirdClassifier
    m = retrained("birdnet_v24")
    onnx_path, bhm_path, table_path = str(tmp_path / "model.onnx"), str(tmp_path / "model.bhm"), str(tmp_path / "table.bhm")
    with open(onnx_path, "wb") as f:
        f.write(ox.dump(convert.graph_from_model(m, frontend_spelling=spelling)))
    mf.write_model(bhm_path, m)
    mf.write_model(table_path, synth.build_model("birdnet_v24"))
    segs = synth.synth_segments(3, m.sample_count, m.sample_rate, start=77)
    out = {}
    for route, path in (("onnx", onnx_path), ("bhm", bhm_path), ("table", table_path)):
        clf = BirdClassifier(path, None)
        ctx = clf.create_batch_context(4)
        out[route] = clf.predict_logits(ctx, segs)
        ctx.close(); clf.close()
    scale = max(1.0, float(np.abs(out["bhm"]).max()))
    assert np.isfinite(out["onnx"]).all() and np.abs(out["onnx"] - out["bhm"]).max() <= 2e-5 * scale
    assert np.abs(out["table"] - out["bhm"]).max() > 1e-3 * scale      # the assumption round 4 made would have been visible here
    # ... and the INDEPENDENT implementation on the retrained constants (the comparison above is HIP against HIP):
    # the plain-C oracle reads the same retrained container -- mag_scale 0.9, affine (1.3, 0.2), fmax 2 800 Hz -- and both device
    # routes are held to it
    ref = oracle_lib.OracleModel(bhm_path).forward(segs)
    oscale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(out["bhm"] - ref).max() <= 2e-5 * oscale and np.abs(out["onnx"] - ref).max() <= 2e-5 * oscale
    assert np.abs(out["table"] - ref).max() > 1e-3 * oscale


def test_constant_spellings_and_repeated_slice_axes_are_held_to_their_payload(tmp_path):
    """ADVICE r5 (medium x 2).  (1) A Constant whose `value_float` attribute carries no float (here: an integer) was registered as a
    one-element tensor WITHOUT data and read -- a crash inside bh_onnx_eval / bh_classifier_create on an untrusted file; and any
    node with a tensor attribute called `value` (ConstantOfShape) had that tensor hoisted in its output's place.  (2) Slice with a
    repeated axis clamped against the original extent and read past the array (x[10], starts [5, 9], axes [0, 0] -> x[14])."""
    x = np.arange(10, dtype=np.float64)
    # the good spellings still work
    got = _eval([ox.Node("Constant", [], ["c"], {"value_float": 2.5}), ox.Node("Add", ["x", "c"], ["y"])], {}, "x", x, "y", tmp_path)
    assert np.array_equal(got, x + 2.5)
    got = _eval([ox.Node("Constant", [], ["c"], {"value_floats": [1.0, 2.0]}), ox.Node("Identity", ["c"], ["y"])], {}, "x", x, "y", tmp_path)
    assert np.array_equal(got, [1.0, 2.0])
    got = _eval([ox.Node("Constant", [], ["c"], {"value": np.asarray([3, 4], np.int64)}), ox.Node("Identity", ["c"], ["y"])], {}, "x", x, "y", tmp_path)
    assert np.array_equal(got, [3, 4])
    # `value_float` holding an integer: malformed, by name (was: a segmentation fault)
    with pytest.raises(RuntimeError, match="malformed Constant"):
        _eval([ox.Node("Constant", [], ["c"], {"value_float": 3}), ox.Node("Add", ["x", "c"], ["y"])], {}, "x", x, "y", tmp_path)
    with pytest.raises(RuntimeError, match="malformed Constant"):
        _eval([ox.Node("Constant", [], ["c"], {"value_int": 3.0}), ox.Node("Add", ["x", "c"], ["y"])], {}, "x", x, "y", tmp_path)
    with pytest.raises(RuntimeError, match="without an output or without a value"):
        _eval([ox.Node("Constant", [], ["c"], {"value_string": "a"}), ox.Node("Add", ["x", "c"], ["y"])], {}, "x", x, "y", tmp_path)
    # a `value` tensor on a node that is not a Constant stays that node's attribute: the evaluator answers for ConstantOfShape itself
    with pytest.raises(RuntimeError, match="ConstantOfShape"):
        _eval([ox.Node("ConstantOfShape", ["s"], ["c"], {"value": np.asarray([7.0], np.float32)}), ox.Node("Identity", ["c"], ["y"])],
              {"s": np.asarray([3], np.int64)}, "x", x, "y", tmp_path)
    # Slice: a repeated axis is refused; the ordinary two-axis slice is untouched
    inits = {"st": np.asarray([5, 9], np.int64), "en": np.asarray([8, 10], np.int64), "ax": np.asarray([0, 0], np.int64)}
    with pytest.raises(RuntimeError, match="given twice"):
        _eval([ox.Node("Slice", ["x", "st", "en", "ax"], ["y"])], inits, "x", x, "y", tmp_path)
    x2 = np.arange(24, dtype=np.float64).reshape(4, 6)
    inits = {"st": np.asarray([1, -4], np.int64), "en": np.asarray([3, 100], np.int64), "ax": np.asarray([0, 1], np.int64), "sp": np.asarray([1, 2], np.int64)}
    got = _eval([ox.Node("Slice", ["x", "st", "en", "ax", "sp"], ["y"])], inits, "x", x2, "y", tmp_path)
    assert np.array_equal(got, x2[1:3, -4:100:2])

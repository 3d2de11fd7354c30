"""The converter reads the spectrogram front-end OFF THE GRAPH by probing it (birda_amd/frontend_recover.py,
birda_amd/onnx_eval.py; SURVEY.md section 7 hard part (ii): the front-end's quirks "must be read off the ONNX graph").

The reference never sees the front-end: it is inside the .onnx file ONNX Runtime executes
(src/inference/classifier.rs:269-283).  Offline there is no such file, so the graphs here are written by
`convert.frontend_nodes` in four deliberately different spellings; what the tests establish is that the recovery does not depend on
the spelling, that it reproduces the parameters the graph was written from, that the converted model gives the oracle the same
logits, and that front-ends the container cannot express are refused with a reason instead of being approximated."""
import math

import numpy as np
import pytest
import torch

from birda_amd import convert, modelfile as mf, onnx_io as ox, synth
from birda_amd.frontend_recover import RecoverError, closed_form_spectrogram, recover_frontend
from birda_amd.onnx_eval import EvalError, Evaluator

SPELLINGS = ("conv1d", "stft", "complex", "fused")


def _mel(m, b):
    return m.weight(b.mel_w_off, b.n_bins * b.n_mels).reshape(b.n_bins, b.n_mels)


@pytest.mark.parametrize("spelling", SPELLINGS)
def test_front_end_is_recovered_whatever_the_spelling(spelling, tmp_path):
    from oracle import oracle as O
    m = synth.build_model("mini")
    g = ox.load(ox.dump(convert.graph_from_model(m, frontend_spelling=spelling)))
    assert g.inputs[0].name == "audio" and g.inputs[0].shape == ["N", m.sample_count]
    m2 = convert.model_from_graph(g, None, sample_rate=m.sample_rate)
    assert (m2.sample_rate, m2.sample_count, m2.spec_h, m2.spec_w, len(m2.branches)) == (m.sample_rate, m.sample_count, m.spec_h,
                                                                                        m.spec_w, len(m.branches))
    assert m2.segment_duration == pytest.approx(m.segment_duration) and m2.norm_eps == pytest.approx(m.norm_eps, rel=1e-5)
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
        # the informational band edges bracket the band the matrix covers
        assert b.fmin <= max(a.fmin, 1.0) + m.sample_rate / a.frame_length and b.fmax >= a.fmax - m.sample_rate / a.frame_length
    assert [L.op for L in m2.layers] == [L.op for L in m.layers]
    # the same logits from the oracle (the mel matrices differ in the last bits: least-squares fit, float32 rounding)
    pa, pb = str(tmp_path / "a.bhm"), str(tmp_path / "b.bhm")
    mf.write_model(pa, m)
    mf.write_model(pb, m2)
    segs = synth.synth_segments(3, m.sample_count, m.sample_rate, start=5)
    la, lb = O.OracleModel(pa).forward(segs), O.OracleModel(pb).forward(segs)
    assert np.abs(la - lb).max() <= 2e-5 * max(1.0, np.abs(la).max())


def test_closed_form_is_the_oracles_front_end(tmp_path):
    """the float64 closed form the recovery verifies against = the oracle's spectrogram (so "verified" means the device's)"""
    from oracle import oracle as O
    m = synth.build_model("mini")
    p = str(tmp_path / "m.bhm")
    mf.write_model(p, m)
    segs = synth.synth_segments(2, m.sample_count, m.sample_rate, start=9)
    om = O.OracleModel(p)
    want = np.stack([om.frontend(s).reshape(len(m.branches), m.spec_h, m.spec_w) for s in segs])
    got = closed_form_spectrogram(segs, m.norm_eps, m.branches, [_mel(m, b) for b in m.branches])
    assert got.shape == want.shape and np.abs(got - want).max() <= 2e-5 * np.abs(want).max()


def _graph(m, spelling="conv1d"):
    return convert.graph_from_model(m, frontend_spelling=spelling)


def test_front_ends_the_container_cannot_express_are_refused():
    m = synth.build_model("mini")
    # a Hamming window: the frame operator is no longer in the span of the Hann-windowed cosines
    g = _graph(m)
    for b, br in enumerate(m.branches):
        L = br.frame_length
        n = np.arange(L)
        ham = 0.54 - 0.46 * np.cos(2 * np.pi * n / L)
        ang = 2 * np.pi * ((n[None, :] * np.arange(br.n_bins)[:, None]) % L) / L
        assert g.initializers[f"fe{b}_dft"].shape == (br.n_bins, 1, L)
        g.initializers[f"fe{b}_dft"] = (ham[None, :] * np.cos(ang))[:, None, :].astype(np.float32)
    with pytest.raises(RecoverError, match="Hann-windowed"):
        recover_frontend(g, m.sample_rate)
    # a per-mel affine after the power law (a BatchNorm over the mel axis)
    g = _graph(m)
    g.initializers["fe0_sc"] = np.linspace(0.5, 1.0, m.branches[0].n_mels).astype(np.float32)   # broadcasts over [N, frames, mels]
    with pytest.raises(RecoverError, match="one scalar function"):
        recover_frontend(g, m.sample_rate)
    # a magnitude spectrogram (|STFT|^2 = re^2 + im^2): the branch tensor is not a single squared matrix
    g = _graph(m, "complex")
    for b, br in enumerate(m.branches):
        sl = next(n for n in g.nodes if n.outputs[0] == f"fe{b}_re")
        g.initializers[f"fe{b}_s1"] = np.asarray([2 * br.n_bins], np.int64)     # keep the imaginary rows too
        mel = g.initializers[f"fe{b}_mel"]
        g.initializers[f"fe{b}_mel"] = np.concatenate([mel, mel], axis=1)      # mel(re) + mel(im): linear, but not a real DFT
        assert sl.op_type == "Slice"
    with pytest.raises(RecoverError, match="Hann-windowed"):
        recover_frontend(g, m.sample_rate)
    # no min / max normalisation at all
    g = _graph(m)
    nx = next(n for n in g.nodes if n.outputs[0] == "fe0_sig")
    nx.inputs[0] = "audio"
    next(n for n in g.nodes if n.outputs[0] == "fe1_sig").inputs[0] = "audio"
    with pytest.raises(RecoverError, match="range"):
        recover_frontend(g, m.sample_rate)
    # an operator outside the evaluator's set: named in the error, and the converter asks for a manifest
    g = _graph(m)
    k = next(i for i, n in enumerate(g.nodes) if n.outputs[0] == "fe_xn")
    g.nodes.insert(k + 1, ox.Node("Loop", ["fe_xn"], ["fe_loop"]))
    for n in g.nodes:
        if n.op_type == "Unsqueeze" and n.inputs[0] == "fe_xn":
            n.inputs[0] = "fe_loop"
    with pytest.raises(RecoverError, match="Loop"):
        recover_frontend(g, m.sample_rate)
    with pytest.raises(convert.ConvertError, match="manifest"):
        convert.model_from_graph(g, None, sample_rate=m.sample_rate)
    with pytest.raises(convert.ConvertError, match="sample rate"):
        convert.model_from_graph(_graph(m), None)


def test_graph_over_audio_converts_with_a_manifest_too():
    """the front-end nodes are skipped, not mistaken for layers, when a manifest is given and the spectrogram is named"""
    m = synth.build_model("mini")
    m2 = convert.model_from_graph(_graph(m, "stft"), m, spectrogram_input="spectrogram")
    assert len(m2.layers) == len(m.layers)
    for a, b in zip(m.branches, m2.branches):
        assert np.array_equal(_mel(m, a), _mel(m2, b))


def test_evaluator_ops_against_torch():
    rng = np.random.default_rng(3)
    F = torch.nn.functional

    def run(nodes, inits, feeds, out):
        g = ox.Graph(nodes=nodes, initializers=inits)
        return Evaluator(g).run(feeds, [out])[0]

    x = rng.standard_normal((2, 3, 50))
    w = rng.standard_normal((4, 3, 7)).astype(np.float32)
    b = rng.standard_normal(4).astype(np.float32)
    got = run([ox.Node("Conv", ["x", "w", "b"], ["y"], {"strides": [3], "pads": [2, 1]})], {"w": w, "b": b}, {"x": x}, "y")
    want = F.conv1d(F.pad(torch.from_numpy(x), (2, 1)), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=3).numpy()
    assert np.allclose(got, want, atol=1e-12)
    x2 = rng.standard_normal((1, 4, 9, 11))
    w2 = rng.standard_normal((4, 1, 3, 3)).astype(np.float32)
    got = run([ox.Node("Conv", ["x", "w"], ["y"], {"strides": [2, 1], "group": 4, "auto_pad": "SAME_UPPER"})], {"w": w2}, {"x": x2}, "y")
    want = F.conv2d(F.pad(torch.from_numpy(x2), (1, 1, 1, 1)), torch.from_numpy(w2).double(), stride=(2, 1), groups=4).numpy()
    assert np.allclose(got, want, atol=1e-12)
    # STFT against torch.stft (no centring, one-sided)
    sig = rng.standard_normal((2, 400))
    win = np.hanning(65)[:64].astype(np.float32)
    got = run([ox.Node("STFT", ["s", "step", "win", "len"], ["y"], {"onesided": 1})],
              {"step": np.asarray(10, np.int64), "win": win, "len": np.asarray(64, np.int64)}, {"s": sig[:, :, None]}, "y")
    ref = torch.stft(torch.from_numpy(sig), 64, hop_length=10, window=torch.from_numpy(win).double(), center=False, return_complex=True)
    assert np.allclose(got[..., 0], ref.real.numpy().transpose(0, 2, 1), atol=1e-10)
    assert np.allclose(got[..., 1], ref.imag.numpy().transpose(0, 2, 1), atol=1e-10)
    # shape ops with their corner semantics
    a = rng.standard_normal((2, 5, 6))
    rev = run([ox.Node("Slice", ["a", "s", "e", "ax", "st"], ["y"])],
              {"s": np.asarray([-1], np.int64), "e": np.asarray([-(2 ** 62)], np.int64), "ax": np.asarray([1], np.int64),
               "st": np.asarray([-1], np.int64)}, {"a": a}, "y")
    assert np.array_equal(rev, a[:, ::-1])
    rs = run([ox.Node("Reshape", ["a", "shape"], ["y"])], {"shape": np.asarray([0, -1], np.int64)}, {"a": a}, "y")
    assert rs.shape == (2, 30)
    us = run([ox.Node("Unsqueeze", ["a", "ax"], ["y"])], {"ax": np.asarray([0, 4], np.int64)}, {"a": a}, "y")
    assert us.shape == (1, 2, 5, 6, 1)
    mm = run([ox.Node("ReduceMax", ["a"], ["y"], {"axes": [1, 2], "keepdims": 0})], {}, {"a": a}, "y")
    assert np.array_equal(mm, a.max(axis=(1, 2)))
    # an intermediate tensor can be fed: only what lies below it is computed
    g = ox.Graph(nodes=[ox.Node("Exp", ["a"], ["b"]), ox.Node("Unknown", ["b"], ["c"]), ox.Node("Neg", ["c"], ["d"])])
    assert np.array_equal(Evaluator(g).run({"c": np.ones(3)}, ["d"])[0], -np.ones(3))
    with pytest.raises(EvalError, match="Unknown"):
        Evaluator(g).run({"a": np.ones(3)}, ["d"])

"""bh_classifier_create opens the `.onnx` itself (reference ClassifierBuilder::model_path,
src/inference/classifier.rs:269-283; label-count check src/inference/mod.rs:34-37).

The library's own graph walk (birda_amd/csrc/onnx_conv.hpp: protobuf wire format, conv-stack walk, front-end from the family
table) is held to the Python converter (birda_amd/convert.py, the second witness) on the container they produce: the same
header, the same branch table and mel weights, the same layer table, the same weights -- bit for bit.  CPU tests go through
bh_onnx_to_bhm (host only); the GPU test compares logits of a classifier created on the `.onnx` with one created on the BHM1
file the Python converter wrote.
"""
import ctypes as C
import os

import numpy as np
import pytest

from birda_amd import _lib, convert, modelfile as mf, onnx_io as ox, synth


def _native(onnx_path, out_path):
    L = _lib.load()
    rc = L.bh_onnx_to_bhm(onnx_path.encode(), out_path.encode())
    if rc != 0:
        raise RuntimeError(f"rc {rc}: {L.bh_last_error().decode()}")
    return mf.read_model(out_path)


def _weights(m, L):
    nw = {mf.OP_CONV: L.kh * L.kw * L.cin * L.cout, mf.OP_DWCONV: L.kh * L.kw * L.cout, mf.OP_PWCONV: L.cin * L.cout,
          mf.OP_DENSE: L.cin * L.cout}.get(L.op, 0)
    if not nw:
        return np.zeros(0, np.float32), np.zeros(0, np.float32)
    return m.blob[L.w_off:L.w_off + nw], m.blob[L.b_off:L.b_off + L.cout]


def _same_model(a, b, frontend_exact=True):
    """a: the Python converter's model, b: the library's.  Offsets into the blob may differ; everything they point at may not.
    frontend_exact=False: b's front-end was READ OFF an audio-input graph by probing (round 5, onnx_frontend.hpp), a's comes from
    the manifest the graph was written from -- the same front-end to the last few bits of a fit, not bit for bit."""
    for f in ("family", "sample_rate", "sample_count", "n_classes", "embedding_dim", "output_activation", "embedding_tensor", "spec_h", "spec_w"):
        assert getattr(a, f) == getattr(b, f), f
    assert np.float32(a.segment_duration) == np.float32(b.segment_duration)
    assert np.float32(a.norm_eps) == np.float32(b.norm_eps) if frontend_exact else b.norm_eps == pytest.approx(a.norm_eps, rel=1e-5)
    assert len(a.branches) == len(b.branches) and len(a.layers) == len(b.layers)
    for x, y in zip(a.branches, b.branches):
        for f in ("frame_length", "frame_step", "n_mels", "n_frames"):
            assert getattr(x, f) == getattr(y, f), f
        n = x.n_bins * x.n_mels
        wa, wb = a.blob[x.mel_w_off:x.mel_w_off + n], b.blob[y.mel_w_off:y.mel_w_off + n]
        if frontend_exact:
            for f in ("fmin", "fmax", "mag_scale", "out_scale", "out_shift", "flags"):
                assert np.float32(getattr(x, f)) == np.float32(getattr(y, f)), f
            assert wa.tobytes() == wb.tobytes(), "mel weight matrix differs"      # the C++ restatement of the HTK mel matrix, bit for bit
        else:
            assert y.mag_scale == pytest.approx(x.mag_scale, abs=1e-5) and (y.out_scale, y.out_shift) == pytest.approx((x.out_scale, x.out_shift), rel=1e-6)
            wb = wb.reshape(x.n_bins, x.n_mels)
            if (x.flags ^ y.flags) & 1:          # (the 'fused' spelling folds the flip into the operator: columns reversed, no flag)
                wb = wb[:, ::-1]
            assert np.abs(wa.reshape(x.n_bins, x.n_mels) - wb).max() < 1e-6
    for i, (x, y) in enumerate(zip(a.layers, b.layers)):
        for f in ("op", "act", "in_tensor", "res_tensor", "cin", "cout", "kh", "kw", "sh", "sw", "pad_t", "pad_l", "in_h", "in_w", "out_h",
                  "out_w", "in_layout"):
            assert getattr(x, f) == getattr(y, f), (i, f, getattr(x, f), getattr(y, f))
        (wa, ba), (wb, bb) = _weights(a, x), _weights(b, y)
        assert wa.tobytes() == wb.tobytes() and ba.tobytes() == bb.tobytes(), (i, "weights differ")


@pytest.mark.parametrize("kind,gelu,frontend", [("mini", "erf", None), ("mini_b0", "gelu", None), ("mini_se", "erf", None),
                                                ("mini", "erf", "conv1d"), ("mini", "gelu", "stft"), ("mini_se", "erf", "fused"),
                                                ("mini_hg", "erf", "complex"), ("birdnet_v24_tiny", "erf", None),
                                                ("birdnet_v24_tiny", "erf", "conv1d"), ("perch_v2_tiny", "erf", None),
                                                ("birdnet_v24", "erf", None)])
def test_native_graph_walk_matches_the_python_converter(tmp_path, kind, gelu, frontend):
    """Graphs written by the repo's own writer, starting at the spectrogram (front-end: the family table, bit for bit the Python
    converter's manifest) or -- like the published files -- at the audio input (four spellings: the native route reads the
    front-end off the graph by probing, tests/test_onnx_frontend.py; the conv stack behind it bit for bit all the same)."""
    m = synth.build_model(kind)
    g = convert.graph_from_model(m, spell_gelu=gelu, frontend_spelling=frontend)
    data = ox.dump(g)
    onnx_path = str(tmp_path / "model.onnx")
    with open(onnx_path, "wb") as f:
        f.write(data)
    want = convert.model_from_graph(ox.load(data), m, "spectrogram" if frontend else None)
    got = _native(onnx_path, str(tmp_path / "native.bhm"))
    _same_model(want, got, frontend_exact=frontend is None)
    # ... and both are the model the graph was written from (activations, residuals, squeeze-excite gates back in place)
    assert [(L.op, L.act, L.res_tensor) for L in got.layers] == [(L.op, L.act, L.res_tensor) for L in m.layers]


def test_native_graph_walk_on_the_exporter_style_graph(tmp_path):
    """The hand-written graph of tests/test_convert.py (un-folded BatchNormalization, asymmetric SAME padding, Clip(0, 6),
    Sigmoid x Mul, residual Add, ReduceMean, MatMul + Add, Gemm(transB), Softmax): not a graph graph_from_model wrote."""
    from test_convert import hand_written_graph
    g, _, _, base = hand_written_graph()
    data = ox.dump(g)
    onnx_path = str(tmp_path / "hand.onnx")
    with open(onnx_path, "wb") as f:
        f.write(data)
    want = convert.model_from_graph(ox.load(data), base)
    got = _native(onnx_path, str(tmp_path / "hand.bhm"))
    _same_model(want, got)
    assert [L.act for L in got.layers[:4]] == [mf.ACT_RELU6, mf.ACT_SWISH, mf.ACT_NONE, mf.ACT_RELU] and got.layers[2].res_tensor == 1


def test_label_count_mismatch_on_an_onnx_file_needs_no_device(tmp_path):
    """`label count != model output width` (reference src/inference/mod.rs:34-37) is BH_ERR_LABELS for an .onnx model too, and is
    found before a device is asked for (this container has none: a matching label file gets BH_ERR_NO_DEVICE instead)."""
    m = synth.build_model("mini")
    onnx_path = str(tmp_path / "mini.onnx")
    with open(onnx_path, "wb") as f:
        f.write(ox.dump(convert.graph_from_model(m)))
    bad, good = str(tmp_path / "bad.txt"), str(tmp_path / "good.txt")
    synth.write_labels(bad, m.n_classes - 1)
    synth.write_labels(good, m.n_classes)
    L = _lib.load()
    h = C.c_void_p()
    cfg = _lib.BhConfig(onnx_path.encode(), bad.encode(), 5, 0.1, 0, 0)
    assert L.bh_classifier_create(C.byref(cfg), C.byref(h)) == -5 and b"label count" in L.bh_last_error()
    if L.bh_device_count() == 0:
        cfg = _lib.BhConfig(onnx_path.encode(), good.encode(), 5, 0.1, 0, 0)
        assert L.bh_classifier_create(C.byref(cfg), C.byref(h)) == -3


def test_refusals_name_their_reason(tmp_path):
    L = _lib.load()
    m = synth.build_model("mini")

    def rc_of(g, name):
        p = str(tmp_path / name)
        with open(p, "wb") as f:
            f.write(ox.dump(g))
        return L.bh_onnx_to_bhm(p.encode(), str(tmp_path / (name + ".bhm")).encode()), L.bh_last_error().decode()

    # an audio input no family has: the error points at the probing converter
    g = ox.Graph(name="odd", producer="tests")
    g.inputs.append(ox.ValueInfo("audio", ox.FLOAT, ["N", 77777]))
    g.outputs.append(ox.ValueInfo("audio", ox.FLOAT, ["N", 77777]))
    rc, msg = rc_of(g, "odd.onnx")
    assert rc == -2 and "77777 samples" in msg and "onnx_to_bhm.py" in msg
    # an operator outside the conv-stack set is refused by name
    g = convert.graph_from_model(m)
    g.nodes.insert(0, ox.Node("LSTM", ["spectrogram"], ["lstm_out"], name="rnn"))
    rc, msg = rc_of(g, "lstm.onnx")
    assert rc == -2 and "LSTM" in msg
    # not an ONNX file at all
    p = str(tmp_path / "junk.onnx")
    with open(p, "wb") as f:
        f.write(os.urandom(4096))
    assert L.bh_onnx_to_bhm(p.encode(), str(tmp_path / "junk.bhm").encode()) == -2


def test_tensor_with_raw_data_and_a_shorter_float_data_is_read_from_raw_data(tmp_path, monkeypatch):
    """an initializer carrying raw_data AND one float_data element used to be read through float_data, seven
    elements past its end.  raw_data wins (as in onnx's own helpers): the model converts to the same container as the clean file."""
    import struct
    m = synth.build_model("mini")
    g = convert.graph_from_model(m)
    clean = str(tmp_path / "clean.onnx")
    with open(clean, "wb") as f:
        f.write(ox.dump(g))
    ser = ox._ser_tensor

    def both(name, arr):
        out = ser(name, arr)
        if np.asarray(arr).dtype == np.float32 and np.asarray(arr).size >= 8:
            out += ox._key(4, 5) + struct.pack("<f", 12345.0)        # one repeated float_data element beside the raw payload
        return out
    monkeypatch.setattr(ox, "_ser_tensor", both)
    dirty = str(tmp_path / "dirty.onnx")
    with open(dirty, "wb") as f:
        f.write(ox.dump(g))
    monkeypatch.undo()
    assert os.path.getsize(dirty) > os.path.getsize(clean)
    a, b = _native(clean, str(tmp_path / "a.bhm")), _native(dirty, str(tmp_path / "b.bhm"))
    _same_model(a, b)


def test_batchnorm_that_cannot_be_folded_is_refused(tmp_path):
    """BN(conv + residual) and a BN whose convolution has another reader (a skip taken before the BN) cannot
    be expressed by rescaling the convolution's weights: both converters refuse them instead of folding silently."""
    from test_convert import hand_written_graph
    L = _lib.load()

    def both_refuse(g, what, name):
        data = ox.dump(g)
        p = str(tmp_path / name)
        with open(p, "wb") as f:
            f.write(data)
        with pytest.raises(convert.ConvertError, match=what):
            convert.model_from_graph(ox.load(data), hand_written_graph()[3])
        assert L.bh_onnx_to_bhm(p.encode(), (p + ".bhm").encode()) == -2 and what in L.bh_last_error().decode()

    rng = np.random.default_rng(8)

    def bn_params(g, tag, c):
        names = []
        for k, v in (("g", rng.uniform(0.5, 1.5, c)), ("b", rng.normal(0, 0.1, c)), ("m", rng.normal(0, 0.1, c)), ("v", rng.uniform(0.5, 1.5, c))):
            g.initializers[f"{tag}_{k}"] = np.asarray(v, np.float32)
            names.append(f"{tag}_{k}")
        return names
    # (a) pre-activation style: Add -> BatchNormalization
    g = hand_written_graph()[0]
    k = next(i for i, n in enumerate(g.nodes) if n.outputs[0] == "r2")
    g.nodes.insert(k + 1, ox.Node("BatchNormalization", ["r2"] + bn_params(g, "bnr", 8), ["r2n"], {"epsilon": 1e-3}))
    next(n for n in g.nodes if n.outputs[0] == "c3").inputs[0] = "r2n"
    both_refuse(g, "residual Add", "add_bn.onnx")
    # (b) the skip is taken from the convolution's output, in front of its BatchNormalization
    g = hand_written_graph()[0]
    next(n for n in g.nodes if n.outputs[0] == "r2").inputs[1] = "c0"
    both_refuse(g, "other readers", "skip_before_bn.onnx")


@pytest.mark.gpu
@pytest.mark.parametrize("kind,frontend", [("birdnet_v24", "conv1d"), ("perch_v2", None), ("mini_se", None), ("birdnet_v30", None)])
def test_classifier_created_on_the_onnx_file_gives_the_bhm_route_logits(tmp_path, kind, frontend):
    """bh_classifier_create("x.onnx") against the BHM1 container the Python converter writes from the same file: logits bit for
    bit for graphs that start at the spectrogram, within 2e-5 of the logit scale for the audio-input spelling (its front-end is
    read off the graph by probing: the mel matrix is a fit, equal to the manifest's to ~1e-8), in the split-f16 default and on the f32 kernels.  birdnet_v24 with the audio-input spelling (what the published file
    is), perch_v2 at its published size (437 MB), a squeeze-excite stack, and the v3.0 contract (sigmoid inside the graph)."""
    from birda_amd.classifier import BirdClassifier
    m = synth.build_model(kind)
    data = ox.dump(convert.graph_from_model(m, frontend_spelling=frontend))
    onnx_path, bhm_path = str(tmp_path / "model.onnx"), str(tmp_path / "python.bhm")
    with open(onnx_path, "wb") as f:
        f.write(data)
    mf.write_model(bhm_path, convert.model_from_graph(ox.load(data), m, "spectrogram" if frontend else None))
    del data
    segs = synth.synth_segments(3, m.sample_count, m.sample_rate, start=77)
    for prec in ("auto", "f32"):
        out = {}
        for route, path in (("onnx", onnx_path), ("bhm", bhm_path)):
            clf = BirdClassifier(path, None, precision=prec)
            assert clf.info.n_classes == m.n_classes and clf.info.sample_count == m.sample_count and clf.info.model_type == m.family
            ctx = clf.create_batch_context(4)
            out[route] = (clf.predict_logits(ctx, segs), clf.fused_blocks())
            ctx.close(); clf.close()
        assert out["onnx"][1] == out["bhm"][1]
        assert np.isfinite(out["onnx"][0]).all()
        if frontend is None:
            assert (out["onnx"][0] == out["bhm"][0]).all(), (kind, prec)
        else:
            assert np.abs(out["onnx"][0] - out["bhm"][0]).max() <= 2e-5 * max(1.0, float(np.abs(out["bhm"][0]).max())), (kind, prec)

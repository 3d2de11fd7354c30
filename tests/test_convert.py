"""ONNX conv stack <-> BHM1 (birda_amd/convert.py, birda_amd/onnx_io.py): the synthetic models are written
out as ONNX bytes with the dependency-free writer, read back with the dependency-free reader and converted;
the result must be the same layer table, the same weights and -- through the oracle -- the same logits.
(The reference hands the .onnx file to birdnet_onnx::ClassifierBuilder, src/inference/classifier.rs:269-283.)"""
import numpy as np
import pytest

from birda_amd import convert, modelfile as mf, onnx_io as ox, synth

LAYER_KEYS = ("op", "act", "in_tensor", "res_tensor", "cin", "cout", "kh", "kw", "sh", "sw", "pad_t", "pad_l",
              "in_h", "in_w", "out_h", "out_w", "in_layout")


def _weights(m, L):
    nw = {mf.OP_CONV: L.kh * L.kw * L.cin * L.cout, mf.OP_DWCONV: L.kh * L.kw * L.cout, mf.OP_PWCONV: L.cin * L.cout,
          mf.OP_DENSE: L.cin * L.cout, mf.OP_GAP: 0, mf.OP_SCALE: 0}[L.op]
    return m.weight(L.w_off, nw), m.weight(L.b_off, L.cout if nw else 0)


def _same_model(a, b, exact=True):
    assert (a.family, a.sample_rate, a.sample_count, a.n_classes, a.embedding_dim, a.output_activation, a.embedding_tensor,
            a.spec_h, a.spec_w) == (b.family, b.sample_rate, b.sample_count, b.n_classes, b.embedding_dim, b.output_activation,
                                    b.embedding_tensor, b.spec_h, b.spec_w)
    assert len(a.layers) == len(b.layers)
    for i, (x, y) in enumerate(zip(a.layers, b.layers)):
        assert tuple(getattr(x, k) for k in LAYER_KEYS) == tuple(getattr(y, k) for k in LAYER_KEYS), i
        for u, v in zip(_weights(a, x), _weights(b, y)):
            assert np.array_equal(u, v) if exact else np.allclose(u, v, rtol=2e-6, atol=1e-7), i
    for x, y in zip(a.branches, b.branches):
        assert np.array_equal(a.weight(x.mel_w_off, x.n_bins * x.n_mels), b.weight(y.mel_w_off, y.n_bins * y.n_mels))


@pytest.mark.parametrize("kind", ["mini", "mini_b0", "birdnet_v24_tiny", "mini_se"])
@pytest.mark.parametrize("spelling", ["erf", "gelu"])
def test_round_trip_through_onnx_bytes(kind, spelling):
    m = synth.build_model(kind)
    data = ox.dump(convert.graph_from_model(m, spell_gelu=spelling))
    g = ox.load(data)
    assert g.inputs[0].name == "spectrogram" and g.inputs[0].shape[1:] == [len(m.branches), m.spec_h, m.spec_w]
    ops = {n.op_type for n in g.nodes}
    assert "Conv" in ops and ("Erf" in ops) == (spelling == "erf" and kind != "mini_se")   # mini_se: swish, no GELU at all
    _same_model(m, convert.model_from_graph(g, m))


def test_converted_model_gives_the_same_oracle_logits(tmp_path):
    from oracle import oracle as O
    m = synth.build_model("mini")
    m2 = convert.model_from_graph(ox.load(ox.dump(convert.graph_from_model(m))), m)
    pa, pb = str(tmp_path / "a.bhm"), str(tmp_path / "b.bhm")
    mf.write_model(pa, m)
    mf.write_model(pb, m2)
    segs = synth.synth_segments(2, m.sample_count, m.sample_rate, start=3)
    assert np.array_equal(O.OracleModel(pa).forward(segs), O.OracleModel(pb).forward(segs))


def test_exporter_variants_fold_to_the_same_table():
    """BatchNormalization after a conv, MatMul + Add instead of Gemm, auto_pad instead of pads, x * (1/sqrt 2)
    instead of x / sqrt 2, Gemm with transB."""
    m = synth.build_model("mini")
    g = convert.graph_from_model(m)
    rng = np.random.default_rng(7)
    # (1) un-fold the first conv: conv' (scaled weights) + BN must fold back to the original numbers
    conv = next(n for n in g.nodes if n.op_type == "Conv")
    cout = g.initializers[conv.inputs[1]].shape[0]
    gamma, beta = rng.uniform(0.5, 1.5, cout), rng.normal(0, 0.1, cout)
    mean, var, eps = rng.normal(0, 0.1, cout), rng.uniform(0.5, 2.0, cout), 1e-3
    scale = gamma / np.sqrt(var + eps)
    g.initializers[conv.inputs[1]] = (g.initializers[conv.inputs[1]] / scale[:, None, None, None]).astype(np.float32)
    g.initializers[conv.inputs[2]] = ((g.initializers[conv.inputs[2]] - beta) / scale + mean).astype(np.float32)
    for k, v in (("bn_g", gamma), ("bn_b", beta), ("bn_m", mean), ("bn_v", var)):
        g.initializers[k] = v.astype(np.float32)
    i = g.nodes.index(conv)
    old_out = conv.outputs[0]
    conv.outputs[0] = old_out + "_prebn"
    g.nodes.insert(i + 1, ox.Node("BatchNormalization", [conv.outputs[0], "bn_g", "bn_b", "bn_m", "bn_v"], [old_out], {"epsilon": float(eps)}))
    # (2) Gemm -> MatMul + Add
    gemm = next(n for n in g.nodes if n.op_type == "Gemm")
    j = g.nodes.index(gemm)
    g.nodes[j: j + 1] = [ox.Node("MatMul", [gemm.inputs[0], gemm.inputs[1]], [gemm.outputs[0] + "_mm"]),
                         ox.Node("Add", [gemm.outputs[0] + "_mm", gemm.inputs[2]], [gemm.outputs[0]])]
    # (3) pads -> auto_pad on every conv; (4) Div by sqrt 2 -> Mul by 1/sqrt 2 on the first GELU
    for n in g.nodes:
        if n.op_type == "Conv":
            del n.attrs["pads"]
            n.attrs["auto_pad"] = "SAME_UPPER"
    div = next(n for n in g.nodes if n.op_type == "Div")
    div.op_type = "Mul"
    g.initializers[div.inputs[1]] = np.float32(1.0 / np.sqrt(2.0)).reshape(())
    m2 = convert.model_from_graph(ox.load(ox.dump(g)), m)
    _same_model(m, m2, exact=False)
    # Gemm with transposed weights
    g3 = convert.graph_from_model(m)
    gemm = next(n for n in g3.nodes if n.op_type == "Gemm")
    g3.initializers[gemm.inputs[1]] = np.ascontiguousarray(g3.initializers[gemm.inputs[1]].T)
    gemm.attrs["transB"] = 1
    _same_model(m, convert.model_from_graph(ox.load(ox.dump(g3)), m))


def test_unsupported_graphs_are_refused_not_guessed():
    m = synth.build_model("mini")
    g = convert.graph_from_model(m)
    g.nodes.insert(1, ox.Node("LRN", [g.nodes[0].outputs[0]], ["lrn_out"], {"size": 3}))
    with pytest.raises(convert.ConvertError):
        convert.model_from_graph(g, m)
    g = convert.graph_from_model(m)
    next(n for n in g.nodes if n.op_type == "Conv").attrs["dilations"] = [2, 2]
    with pytest.raises(convert.ConvertError):
        convert.model_from_graph(g, m)


def test_wire_format_scalars_and_attributes():
    n = ox.Node("X", ["a"], ["b"], {"f": 0.25, "i": -3, "s": "SAME_UPPER", "ints": [1, -2, 3], "floats": [0.5, 1.5],
                                    "t": np.arange(6, dtype=np.int64).reshape(2, 3)})
    g = ox.Graph(nodes=[n], initializers={"w": np.float32(2.5).reshape(())}, inputs=[ox.ValueInfo("a", ox.FLOAT, ["N", 3])],
                 outputs=[ox.ValueInfo("b", ox.FLOAT, ["N", 3])], opset=13)
    h = ox.load(ox.dump(g))
    a = h.nodes[0].attrs
    assert (a["f"], a["i"], a["s"], a["ints"], a["floats"]) == (0.25, -3, "SAME_UPPER", [1, -2, 3], [0.5, 1.5])
    assert np.array_equal(a["t"], np.arange(6).reshape(2, 3)) and h.opset == 13 and h.inputs[0].shape == ["N", 3]
    assert h.initializers["w"].shape == () and float(h.initializers["w"]) == 2.5


def hand_written_graph():
    """The exporter-style graph of test_hand_written_graph_matches_an_independent_torch_evaluation (also read by the library's own
    C++ graph walk in tests/test_onnx_native.py): returns (graph, torch parameters, stem pads, the model whose front-end it uses)."""
    import torch
    rng = np.random.default_rng(123)
    base = synth.build_model("mini")                       # only its front-end is used
    C0, H, W = len(base.branches), base.spec_h, base.spec_w
    g = ox.Graph(name="hand_written", producer="tests")
    g.inputs.append(ox.ValueInfo("spec", ox.FLOAT, ["N", C0, H, W]))
    P = {}

    def init(name, arr):
        g.initializers[name] = np.asarray(arr, np.float32)
        P[name] = torch.from_numpy(np.asarray(arr, np.float32))
        return name

    def bn(x, c, tag):
        names = [init(f"{tag}_{k}", v) for k, v in (("g", rng.uniform(0.5, 1.5, c)), ("b", rng.normal(0, 0.1, c)),
                                                    ("m", rng.normal(0, 0.1, c)), ("v", rng.uniform(0.5, 1.5, c)))]
        g.nodes.append(ox.Node("BatchNormalization", [x] + names, [tag + "_o"], {"epsilon": 1e-3}))
        return tag + "_o"

    # stem: 3x3 stride 2, TF "SAME" on an even/odd size = pad only bottom/right as needed
    oh, ow = -(-H // 2), -(-W // 2)
    ph, pw = max((oh - 1) * 2 + 3 - H, 0), max((ow - 1) * 2 + 3 - W, 0)
    pads = [ph // 2, pw // 2, ph - ph // 2, pw - pw // 2]
    init("w0", rng.normal(0, 0.3, (8, C0, 3, 3)))
    g.nodes.append(ox.Node("Conv", ["spec", "w0"], ["c0"], {"kernel_shape": [3, 3], "strides": [2, 2], "pads": pads}))
    x = bn("c0", 8, "bn0")
    g.nodes.append(ox.Node("Clip", [x, init("lo", 0.0), init("hi", 6.0)], ["a0"]))
    # depthwise 3x3 + BN + swish
    init("w1", rng.normal(0, 0.4, (8, 1, 3, 3)))
    g.nodes.append(ox.Node("Conv", ["a0", "w1"], ["c1"], {"kernel_shape": [3, 3], "strides": [1, 1], "pads": [1, 1, 1, 1], "group": 8}))
    x = bn("c1", 8, "bn1")
    g.nodes.append(ox.Node("Sigmoid", [x], ["s1"]))
    g.nodes.append(ox.Node("Mul", [x, "s1"], ["a1"]))
    # project 1x1 (with bias) + residual
    init("w2", rng.normal(0, 0.3, (8, 8, 1, 1))); init("b2", rng.normal(0, 0.1, 8))
    g.nodes.append(ox.Node("Conv", ["a1", "w2", "b2"], ["c2"], {"kernel_shape": [1, 1]}))
    g.nodes.append(ox.Node("Add", ["c2", "a0"], ["r2"]))
    # head 1x1 + relu, mean over H, W, flatten, MatMul + Add, Gemm transB, softmax
    init("w3", rng.normal(0, 0.3, (16, 8, 1, 1))); init("b3", rng.normal(0, 0.1, 16))
    g.nodes.append(ox.Node("Conv", ["r2", "w3", "b3"], ["c3"], {"kernel_shape": [1, 1]}))
    g.nodes.append(ox.Node("Relu", ["c3"], ["a3"]))
    g.nodes.append(ox.Node("ReduceMean", ["a3"], ["p3"], {"axes": [2, 3], "keepdims": 1}))
    g.nodes.append(ox.Node("Flatten", ["p3"], ["f3"], {"axis": 1}))
    init("w4", rng.normal(0, 0.3, (16, 12))); init("b4", rng.normal(0, 0.1, 12))
    g.nodes.append(ox.Node("MatMul", ["f3", "w4"], ["m4"]))
    g.nodes.append(ox.Node("Add", ["m4", "b4"], ["d4"]))
    g.nodes.append(ox.Node("Relu", ["d4"], ["a4"]))
    init("w5", rng.normal(0, 0.5, (7, 12))); init("b5", rng.normal(0, 0.1, 7))
    g.nodes.append(ox.Node("Gemm", ["a4", "w5", "b5"], ["logits"], {"transB": 1}))
    g.nodes.append(ox.Node("Softmax", ["logits"], ["prob"], {"axis": -1}))
    g.outputs.append(ox.ValueInfo("prob", ox.FLOAT, ["N", 7]))

    return g, P, pads, base


def test_hand_written_graph_matches_an_independent_torch_evaluation(tmp_path):
    """A graph spelled the way a TF -> ONNX exporter spells it, written node by node HERE (not by graph_from_model) and
    evaluated with torch.nn.functional on the oracle's own spectrogram: the converted model's oracle logits must agree.
    Covers what the round trips cannot: un-folded BatchNormalization, asymmetric SAME padding of a stride-2 conv,
    Clip(0, 6), Sigmoid * x, residual Add, ReduceMean over H, W, MatMul + Add, Gemm(transB = 1), a Softmax output."""
    import torch
    import torch.nn.functional as F
    from oracle import oracle as O
    g, P, pads, base = hand_written_graph()
    C0, H, W = len(base.branches), base.spec_h, base.spec_w
    m = convert.model_from_graph(ox.load(ox.dump(g)), base)
    assert m.n_classes == 7 and m.output_activation == mf.OUT_SOFTMAX and m.embedding_dim == 16
    assert [L.op for L in m.layers] == [mf.OP_CONV, mf.OP_DWCONV, mf.OP_PWCONV, mf.OP_PWCONV, mf.OP_GAP, mf.OP_DENSE, mf.OP_DENSE]
    assert [L.act for L in m.layers[:4]] == [mf.ACT_RELU6, mf.ACT_SWISH, mf.ACT_NONE, mf.ACT_RELU] and m.layers[2].res_tensor == 1
    pa, pb = str(tmp_path / "base.bhm"), str(tmp_path / "conv.bhm")
    mf.write_model(pa, base); mf.write_model(pb, m)
    segs = synth.synth_segments(3, base.sample_count, base.sample_rate, start=11)
    _, spec = O.OracleModel(pa).forward(segs, dump_tensor=0)             # the front-end's output, shared by both sides
    xs = torch.from_numpy(spec.reshape(3, C0, H, W).copy())

    def tbn(x, tag):
        return F.batch_norm(x, P[tag + "_m"], P[tag + "_v"], P[tag + "_g"], P[tag + "_b"], False, 0.0, 1e-3)
    y = F.conv2d(F.pad(xs, (pads[1], pads[3], pads[0], pads[2])), P["w0"], None, stride=2)
    a0 = torch.clamp(tbn(y, "bn0"), 0.0, 6.0)
    y = tbn(F.conv2d(a0, P["w1"], None, padding=1, groups=8), "bn1")
    a1 = y * torch.sigmoid(y)
    r2 = F.conv2d(a1, P["w2"], P["b2"]) + a0
    a3 = F.relu(F.conv2d(r2, P["w3"], P["b3"]))
    f3 = a3.mean((2, 3))
    a4 = F.relu(f3 @ P["w4"] + P["b4"])
    want = (a4 @ P["w5"].T + P["b5"]).numpy()
    got = O.OracleModel(pb).forward(segs)                                 # logits (the softmax is the output activation)
    assert np.abs(got - want).max() <= 2e-5 * max(1.0, float(np.abs(want).max())), float(np.abs(got - want).max())

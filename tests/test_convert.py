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
          mf.OP_DENSE: L.cin * L.cout, mf.OP_GAP: 0}[L.op]
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


@pytest.mark.parametrize("kind", ["mini", "mini_b0", "birdnet_v24_tiny"])
@pytest.mark.parametrize("spelling", ["erf", "gelu"])
def test_round_trip_through_onnx_bytes(kind, spelling):
    m = synth.build_model(kind)
    data = ox.dump(convert.graph_from_model(m, spell_gelu=spelling))
    g = ox.load(data)
    assert g.inputs[0].name == "spectrogram" and g.inputs[0].shape[1:] == [len(m.branches), m.spec_h, m.spec_w]
    ops = {n.op_type for n in g.nodes}
    assert "Conv" in ops and ("Erf" in ops) == (spelling == "erf")
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

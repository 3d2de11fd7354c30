"""ONNX conv stack  <->  BHM1 layer table.

`model_from_graph` turns the classifier part of an ONNX graph -- from the spectrogram tensor
[N, C, H, W] to the logits -- into the layer table `modelfile.py` describes: Conv (group 1 / depthwise
/ 1x1) with BatchNormalization folded, the activation that follows folded into the producing layer
(Relu, Clip 0..6, Sigmoid*x, the Div-Erf-Add-Mul-Mul spelling of exact GELU, or a fused `Gelu`),
residual Add folded into the project conv, squeeze-excite blocks (GlobalAveragePool -> 1x1 Conv -> activation -> 1x1 Conv ->
Sigmoid -> Mul with the feature map) as pool / 1x1 / 1x1 / OP_SCALE layers, GlobalAveragePool / ReduceMean, Flatten, Gemm / MatMul + Add,
final Sigmoid / Softmax as the output activation.  Weights move from ONNX's [Cout, Cin/g, kh, kw] to
the NHWC-friendly layouts of the kernels.

The STFT / mel front-end: either the model family's manifest values through `frontend`, with the
graph entered at the spectrogram tensor named by `spectrogram_input` (default: the graph's first input) -- or, with
`frontend=None` and a sample rate, read off the graph itself by probing (`frontend_recover.py`: how the published BirdNET and
Perch ONNX files spell that part -- STFT op, DFT-as-Conv1d, ... -- is not knowable offline, so no spelling is matched; the nodes
in front of the first 2-D convolution are evaluated on probe signals and the container's parameters fitted to the responses).
The reference loads the same file through `birdnet_onnx::ClassifierBuilder::model_path` (src/inference/classifier.rs:269-283).

`graph_from_model` is the inverse for the synthetic models; it exists so that the converter can be
tested (tests/test_convert.py: model -> ONNX bytes -> model, identical layer tables, weights and oracle
logits).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import modelfile as mf
from . import onnx_io as ox

SQRT2 = math.sqrt(2.0)


class ConvertError(ValueError):
    pass


# ---------------------------------------------------------------------------------------
# model -> ONNX  (test fixture generator)
# ---------------------------------------------------------------------------------------
def frontend_nodes(g: ox.Graph, m: mf.Model, spelling: str) -> None:
    """Appends the spectrogram front-end of `m` to `g` as nodes from `audio` [N, S] to `spectrogram`
    [N, C, n_mels, n_frames], in one of several spellings an exporter might choose.  Fixture generator for
    tests/test_frontend_recover.py: the recovery must not care which one it gets.
      'conv1d'  DFT as a strided Conv (real rows only) -> MatMul mel -> Mul(t, t) -> Pow(const) -> Mul, Add -> Transpose -> Slice(-1)
      'stft'    opset-17 STFT node -> Gather real part -> MatMul mel -> Pow(t, 2) -> Pow(1 / (1 + Exp(mag_scale))) ->
                BatchNormalization-free affine as Sub / Div -> Gather with reversed indices
      'complex' Conv with cos AND -sin rows -> Slice of the real rows -> 1x1 Conv as the mel projection; the normalisation
                written as (x - min) * (2 / (range + eps)) - 1
      'fused'   window, DFT, mel matrix AND the mel flip folded into the Conv weights: [n_mels, 1, L] rows, nothing else linear"""
    def const(name, arr, dtype=np.float32):
        g.initializers[name] = np.asarray(arr, dtype)
        return name

    def node(op, ins, out, **attrs):
        g.nodes.append(ox.Node(op, list(ins), [out], attrs, name=out))
        return out

    ax1 = const("fe_ax1", [1], np.int64)
    mn = node("ReduceMin", ["audio"], "fe_min", axes=[1], keepdims=1)
    mx = node("ReduceMax", ["audio"], "fe_max", axes=[1], keepdims=1)
    rng_ = node("Sub", [mx, mn], "fe_range")
    den = node("Add", [rng_, const("fe_eps", np.float32(m.norm_eps))], "fe_den")
    if spelling == "complex":
        sc = node("Div", [const("fe_two", 2.0), den], "fe_sc")
        xn = node("Sub", [node("Mul", [node("Sub", ["audio", mn], "fe_c"), sc], "fe_cs"), const("fe_one", 1.0)], "fe_xn")
    else:
        x01 = node("Div", [node("Sub", ["audio", mn], "fe_c"), den], "fe_x01")
        xn = node("Mul", [node("Sub", [x01, const("fe_half", 0.5)], "fe_xh"), const("fe_two", 2.0)], "fe_xn")
    chans = []
    for b, br in enumerate(m.branches):
        t = f"fe{b}"
        L, H, bins = br.frame_length, br.frame_step, br.n_bins
        n = np.arange(L)
        hann = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / L)
        ang = 2.0 * np.pi * ((n[:, None] * np.arange(bins)[None, :]) % L) / L
        wcos, wsin = hann[:, None] * np.cos(ang), -hann[:, None] * np.sin(ang)          # [L, bins]
        mel = m.weight(br.mel_w_off, bins * br.n_mels).reshape(bins, br.n_mels).astype(np.float64)
        expo = 1.0 / (1.0 + math.exp(br.mag_scale))
        flip = bool(br.flags & 1)
        if spelling == "stft":
            sig = node("Unsqueeze", [xn, const("fe_ax2", [2], np.int64)], t + "_sig")
            st = node("STFT", [sig, const(t + "_step", H, np.int64), const(t + "_win", hann), const(t + "_len", L, np.int64)],
                      t + "_stft", onesided=1)                                                 # [N, frames, bins, 2]
            re = node("Gather", [st, const(t + "_zero", 0, np.int64)], t + "_re", axis=3)
            lin = node("MatMul", [re, const(t + "_mel", mel)], t + "_lin")                      # [N, frames, mels]
            layout = "tm"
        else:
            sig = node("Unsqueeze", [xn, ax1], t + "_sig")                                      # [N, 1, S]
            if spelling == "conv1d":
                cv = node("Conv", [sig, const(t + "_dft", wcos.T[:, None, :])], t + "_cv", strides=[H], kernel_shape=[L])
                tr = node("Transpose", [cv], t + "_tr", perm=[0, 2, 1])
                lin = node("MatMul", [tr, const(t + "_mel", mel)], t + "_lin")
                layout = "tm"
            elif spelling == "complex":
                w = np.concatenate([wcos.T, wsin.T], axis=0)[:, None, :]                       # [2 bins, 1, L]
                cv = node("Conv", [sig, const(t + "_dft", w)], t + "_cv", strides=[H], kernel_shape=[L])
                re = node("Slice", [cv, const(t + "_s0", [0], np.int64), const(t + "_s1", [bins], np.int64), ax1], t + "_re")
                lin = node("Conv", [re, const(t + "_mel", mel.T[:, :, None])], t + "_lin", kernel_shape=[1])   # [N, mels, frames]
                layout = "mt"
            elif spelling == "fused":
                G = wcos @ mel                                                                  # [L, mels]
                if flip:
                    G = G[:, ::-1]
                lin = node("Conv", [sig, const(t + "_op", np.ascontiguousarray(G.T)[:, None, :])], t + "_lin", strides=[H], kernel_shape=[L])
                layout, flip = "mt", False
            else:
                raise ConvertError(f"front-end spelling {spelling!r}")
        if spelling == "stft":
            sq = node("Pow", [lin, const(t + "_p2", 2.0)], t + "_sq")
            e1 = node("Exp", [const(t + "_mag", np.float32(br.mag_scale))], t + "_e1")
            ex = node("Div", [const(t + "_n1", 1.0), node("Add", [e1, const(t + "_o1", 1.0)], t + "_e2")], t + "_ex")
            pw = node("Pow", [sq, ex], t + "_pw")
            # an affine spelled as a normalisation: (v - mean) / std
            af = node("Div", [node("Sub", [pw, const(t + "_mean", np.float32(-br.out_shift / br.out_scale))], t + "_ctr"),
                              const(t + "_std", np.float32(1.0 / br.out_scale))], t + "_af")
        else:
            sq = node("Mul", [lin, lin], t + "_sq")
            pw = node("Pow", [sq, const(t + "_ex", np.float32(expo))], t + "_pw")
            af = node("Add", [node("Mul", [pw, const(t + "_sc", np.float32(br.out_scale))], t + "_scd"),
                              const(t + "_sh", np.float32(br.out_shift))], t + "_af")
        if layout == "tm":
            af = node("Transpose", [af], t + "_mt", perm=[0, 2, 1])                            # [N, mels, frames]
        if flip and spelling == "stft":
            af = node("Gather", [af, const(t + "_rev", np.arange(br.n_mels - 1, -1, -1), np.int64)], t + "_fl", axis=1)
        elif flip:
            af = node("Slice", [af, const(t + "_f0", [-1], np.int64), const(t + "_f1", [-(2 ** 62)], np.int64), ax1,
                                const(t + "_fs", [-1], np.int64)], t + "_fl")
        chans.append(node("Unsqueeze", [af, ax1], t + "_ch"))
    node("Concat", chans, "spectrogram", axis=1)


def graph_from_model(m: mf.Model, spell_gelu: str = "erf", frontend_spelling: Optional[str] = None) -> ox.Graph:
    """The conv stack of a BHM1 model as an ONNX graph over `spectrogram` [N, C, H, W] -- or, with `frontend_spelling`
    (see `frontend_nodes`), the whole model over `audio` [N, S].
    spell_gelu: 'erf' = Div, Erf, Add, Mul, Mul (what exporters emit below opset 20); 'gelu' = one Gelu node."""
    g = ox.Graph(name="birda_conv_stack", producer="birda_amd.convert")
    if frontend_spelling is None:
        g.inputs.append(ox.ValueInfo("spectrogram", ox.FLOAT, ["N", len(m.branches), m.spec_h, m.spec_w]))
    else:
        g.inputs.append(ox.ValueInfo("audio", ox.FLOAT, ["N", m.sample_count]))
        frontend_nodes(g, m, frontend_spelling)
    names = {0: "spectrogram"}
    flat = set()   # tensors that are [N, C] (after Flatten)

    def const(name: str, arr) -> str:
        g.initializers[name] = np.asarray(arr, np.float32)
        return name

    def activation(x: str, act: int, tag: str) -> str:
        if act == mf.ACT_NONE:
            return x
        if act == mf.ACT_RELU:
            g.nodes.append(ox.Node("Relu", [x], [tag + "_relu"]))
            return tag + "_relu"
        if act == mf.ACT_RELU6:
            g.nodes.append(ox.Node("Clip", [x, const(tag + "_min", 0.0), const(tag + "_max", 6.0)], [tag + "_relu6"]))
            return tag + "_relu6"
        if act == mf.ACT_SWISH:
            g.nodes.append(ox.Node("Sigmoid", [x], [tag + "_sig"]))
            g.nodes.append(ox.Node("Mul", [x, tag + "_sig"], [tag + "_swish"]))
            return tag + "_swish"
        if act == mf.ACT_SIGMOID:
            g.nodes.append(ox.Node("Sigmoid", [x], [tag + "_gate"]))
            return tag + "_gate"
        if act == mf.ACT_GELU_ERF:
            if spell_gelu == "gelu":
                g.nodes.append(ox.Node("Gelu", [x], [tag + "_gelu"], {"approximate": "none"}))
                return tag + "_gelu"
            g.nodes.append(ox.Node("Div", [x, const(tag + "_sqrt2", np.float32(SQRT2))], [tag + "_d"]))
            g.nodes.append(ox.Node("Erf", [tag + "_d"], [tag + "_e"]))
            g.nodes.append(ox.Node("Add", [tag + "_e", const(tag + "_one", 1.0)], [tag + "_a"]))
            g.nodes.append(ox.Node("Mul", [x, tag + "_a"], [tag + "_m"]))
            g.nodes.append(ox.Node("Mul", [tag + "_m", const(tag + "_half", 0.5)], [tag + "_gelu"]))
            return tag + "_gelu"
        raise ConvertError(f"activation {act} has no ONNX spelling here")

    for i, L in enumerate(m.layers):
        tag, x = f"l{i}", names[L.in_tensor]
        if L.op in (mf.OP_CONV, mf.OP_DWCONV, mf.OP_PWCONV):
            if L.op == mf.OP_CONV:
                w = m.weight(L.w_off, L.kh * L.kw * L.cin * L.cout).reshape(L.kh, L.kw, L.cin, L.cout).transpose(3, 2, 0, 1)
                group = 1
            elif L.op == mf.OP_DWCONV:
                w = m.weight(L.w_off, L.kh * L.kw * L.cout).reshape(L.kh, L.kw, L.cout).transpose(2, 0, 1)[:, None]
                group = L.cout
            else:
                w = m.weight(L.w_off, L.cin * L.cout).reshape(L.cin, L.cout).T[:, :, None, None]
                group = 1
            pad_b = max((L.out_h - 1) * L.sh + L.kh - L.in_h - L.pad_t, 0)
            pad_r = max((L.out_w - 1) * L.sw + L.kw - L.in_w - L.pad_l, 0)
            g.nodes.append(ox.Node("Conv", [x, const(tag + "_w", w), const(tag + "_b", m.weight(L.b_off, L.cout))], [tag + "_conv"],
                                   {"group": group, "kernel_shape": [L.kh, L.kw], "strides": [L.sh, L.sw],
                                    "pads": [L.pad_t, L.pad_l, pad_b, pad_r], "dilations": [1, 1]}, name=tag))
            y = tag + "_conv"
            if L.res_tensor != mf.NO_TENSOR:   # our executor: act(conv) + residual
                y = activation(y, L.act, tag)
                g.nodes.append(ox.Node("Add", [y, names[L.res_tensor]], [tag + "_res"]))
                y = tag + "_res"
            else:
                y = activation(y, L.act, tag)
        elif L.op == mf.OP_GAP:
            g.nodes.append(ox.Node("GlobalAveragePool", [x], [tag + "_gap"]))
            feeds_conv = any(M.in_tensor == i + 1 and M.op == mf.OP_PWCONV for M in m.layers)
            if feeds_conv:       # squeeze-excite: the pooled [N, C, 1, 1] map goes on through 1x1 convolutions
                y = tag + "_gap"
            else:
                g.nodes.append(ox.Node("Flatten", [tag + "_gap"], [tag + "_flat"], {"axis": 1}))
                y = tag + "_flat"
                flat.add(i + 1)
        elif L.op == mf.OP_SCALE:
            g.nodes.append(ox.Node("Mul", [x, names[L.res_tensor]], [tag + "_se"]))
            y = tag + "_se"
        elif L.op == mf.OP_DENSE:
            w = m.weight(L.w_off, L.cin * L.cout).reshape(L.cin, L.cout)
            g.nodes.append(ox.Node("Gemm", [x, const(tag + "_w", w), const(tag + "_b", m.weight(L.b_off, L.cout))], [tag + "_fc"],
                                   {"alpha": 1.0, "beta": 1.0, "transB": 0}))
            y = activation(tag + "_fc", L.act, tag)
            flat.add(i + 1)
        else:
            raise ConvertError(f"layer {i}: op {L.op}")
        names[i + 1] = y
    out = names[len(m.layers)]
    if m.output_activation == mf.OUT_SIGMOID:
        g.nodes.append(ox.Node("Sigmoid", [out], ["probabilities"]))
        out = "probabilities"
    elif m.output_activation == mf.OUT_SOFTMAX:
        g.nodes.append(ox.Node("Softmax", [out], ["probabilities"], {"axis": -1}))
        out = "probabilities"
    g.outputs.append(ox.ValueInfo(out, ox.FLOAT, ["N", m.n_classes]))
    return g


# ---------------------------------------------------------------------------------------
# ONNX -> model
# ---------------------------------------------------------------------------------------
class _Blob:
    def __init__(self, seed: Optional[np.ndarray] = None):
        self.chunks: List[np.ndarray] = [] if seed is None or seed.size == 0 else [np.asarray(seed, np.float32).ravel()]
        self.off = sum(c.size for c in self.chunks)

    def put(self, a: np.ndarray) -> int:
        a = np.ascontiguousarray(a, np.float32).ravel()
        pad = (-self.off) % 16      # 64-B alignment: 16-B vector loads on the device
        if pad:
            self.chunks.append(np.zeros(pad, np.float32))
            self.off += pad
        off = self.off
        self.chunks.append(a)
        self.off += a.size
        return off

    def array(self) -> np.ndarray:
        return np.concatenate(self.chunks) if self.chunks else np.zeros(0, np.float32)


def _scalar(g: ox.Graph, name: str) -> Optional[float]:
    a = g.initializers.get(name)
    return float(a.reshape(-1)[0]) if a is not None and a.size == 1 else None


def _collapse_activations(g: ox.Graph):
    """Find the multi-node activation spellings.  Returns (skip: set of node indices,
    act_of: {final output name: (input name, ACT)})."""
    prod = {o: i for i, n in enumerate(g.nodes) for o in n.outputs}
    cons: Dict[str, List[int]] = {}
    for i, n in enumerate(g.nodes):
        for x in n.inputs:
            cons.setdefault(x, []).append(i)
    skip, act_of = set(), {}

    def sole_consumer(name: str, op: str) -> Optional[int]:
        c = cons.get(name, [])
        return c[0] if len(c) == 1 and g.nodes[c[0]].op_type == op else None

    for i, n in enumerate(g.nodes):
        if n.op_type == "Erf":
            # x / sqrt2 (or x * (1/sqrt2)) -> Erf -> + 1 -> * x, * 0.5 (either order)
            j = prod.get(n.inputs[0])
            if j is None or g.nodes[j].op_type not in ("Div", "Mul"):
                continue
            pre = g.nodes[j]
            cst = [(_scalar(g, a), b) for a, b in ((pre.inputs[1], pre.inputs[0]), (pre.inputs[0], pre.inputs[1]))]
            x = None
            for c, other in cst:
                if c is None:
                    continue
                if (pre.op_type == "Div" and abs(c - SQRT2) < 1e-4) or (pre.op_type == "Mul" and abs(c - 1 / SQRT2) < 1e-4):
                    x = other
            k = sole_consumer(n.outputs[0], "Add")
            if x is None or k is None:
                continue
            add = g.nodes[k]
            if not any(abs((_scalar(g, a) or 0.0) - 1.0) < 1e-6 for a in add.inputs):
                continue
            m1 = sole_consumer(add.outputs[0], "Mul")
            if m1 is None:
                continue
            m2 = sole_consumer(g.nodes[m1].outputs[0], "Mul")
            if m2 is None:
                continue
            others = [a for a in g.nodes[m1].inputs if a != add.outputs[0]] + [a for a in g.nodes[m2].inputs if a != g.nodes[m1].outputs[0]]
            has_x = x in others
            has_half = any(abs((_scalar(g, a) or 0.0) - 0.5) < 1e-6 for a in others)
            if has_x and has_half:
                skip |= {j, i, k, m1, m2}
                act_of[g.nodes[m2].outputs[0]] = (x, mf.ACT_GELU_ERF)
        elif n.op_type == "Sigmoid":
            k = sole_consumer(n.outputs[0], "Mul")
            if k is not None and n.inputs[0] in g.nodes[k].inputs:
                skip |= {i, k}
                act_of[g.nodes[k].outputs[0]] = (n.inputs[0], mf.ACT_SWISH)
    return skip, act_of


def model_from_graph(g: ox.Graph, frontend: Optional[mf.Model] = None, spectrogram_input: Optional[str] = None,
                     sample_rate: Optional[int] = None, family: int = 0) -> mf.Model:
    """`frontend` supplies everything the conv stack does not say: family, sample rate / count, segment
    duration, normalisation eps, the STFT / mel branches and their mel weight matrices (its blob is kept as
    the start of the new blob, so `mel_w_off` stays valid).  Without one, the graph must start at the AUDIO input and the
    front-end is read off it by probing (`frontend_recover.recover_frontend`; `sample_rate` is then required: the graph does
    not state it, the reference takes it from the model type's config, src/inference/classifier.rs:360-377)."""
    if frontend is None:
        from .frontend_recover import RecoverError, recover_frontend
        if sample_rate is None:
            raise ConvertError("no front-end manifest: the sample rate must be given to read the front-end off the graph")
        try:
            rec = recover_frontend(g, sample_rate, family)
        except RecoverError as e:
            raise ConvertError(f"front-end not recoverable from the graph ({e}); pass a front-end manifest") from None
        frontend, spectrogram_input = rec.frontend, rec.spectrogram
    spec = spectrogram_input or (g.inputs[0].name if g.inputs else None)
    if spec is None:
        raise ConvertError("graph has no input")
    # nodes that produce the spectrogram (a graph that starts at the audio input) are the front-end, not layers
    prod = {o: i for i, n in enumerate(g.nodes) for o in n.outputs}
    front_nodes, stack = set(), [spec]
    while stack:
        t = stack.pop()
        i = prod.get(t)
        if i is None or i in front_nodes:
            continue
        front_nodes.add(i)
        stack.extend(g.nodes[i].inputs)
    blob = _Blob(frontend.blob[: max((b.mel_w_off + b.n_bins * b.n_mels for b in frontend.branches), default=0)])
    layers: List[mf.Layer] = []
    # name -> (tensor index, C, H, W) ; H = W = 0 for flattened [N, C]
    tmap: Dict[str, Tuple[int, int, int, int]] = {spec: (0, len(frontend.branches), frontend.spec_h, frontend.spec_w)}
    skip, act_of = _collapse_activations(g)
    out_act, emb_tensor, emb_dim = mf.OUT_NONE, 0, 0
    graph_out = {o.name for o in g.outputs}

    def set_act(name_in: str, name_out: str, act: int):
        t = tmap.get(name_in)
        if t is None:
            raise ConvertError(f"activation on unknown tensor {name_in!r}")
        L = layers[t[0] - 1] if t[0] > 0 else None
        if L is None or L.act != mf.ACT_NONE or L.res_tensor != mf.NO_TENSOR:
            raise ConvertError(f"activation after {name_in!r} cannot be folded into its producer")
        L.act = act
        tmap[name_out] = t

    for out_name, (x, act) in act_of.items():
        pass   # applied when the walk reaches the pattern's first node (below)
    first_of_pattern = {}
    for out_name, (x, act) in act_of.items():
        first_of_pattern.setdefault(x, []).append((out_name, act))

    def flush_patterns(x: str):
        for out_name, act in first_of_pattern.pop(x, []):
            set_act(x, out_name, act)

    for i, n in enumerate(g.nodes):
        if i in skip or i in front_nodes:
            continue
        op = n.op_type
        if op == "Conv":
            x = tmap.get(n.inputs[0])
            if x is None:
                raise ConvertError(f"Conv {n.name!r}: input {n.inputs[0]!r} is not on the path from {spec!r}")
            t_in, cin, h, w = x
            W = g.initializers.get(n.inputs[1])
            if W is None or W.ndim != 4:
                raise ConvertError(f"Conv {n.name!r}: weights must be a 4-d initializer")
            cout, cin_g, kh, kw = W.shape
            B = g.initializers.get(n.inputs[2]) if len(n.inputs) > 2 else np.zeros(cout, np.float32)
            group = int(n.attrs.get("group", 1))
            sh, sw = (n.attrs.get("strides") or [1, 1])
            if any(d != 1 for d in (n.attrs.get("dilations") or [1, 1])):
                raise ConvertError(f"Conv {n.name!r}: dilation")
            oh, ow = -(-h // sh), -(-w // sw)
            auto = n.attrs.get("auto_pad", "NOTSET")
            if auto in ("SAME_UPPER", "SAME_LOWER"):
                th, tw = max((oh - 1) * sh + kh - h, 0), max((ow - 1) * sw + kw - w, 0)
                pt, pl = (th // 2, tw // 2) if auto == "SAME_UPPER" else (th - th // 2, tw - tw // 2)
            else:
                pads = n.attrs.get("pads") or [0, 0, 0, 0]
                pt, pl = pads[0], pads[1]
                oh, ow = (h + pads[0] + pads[2] - kh) // sh + 1, (w + pads[1] + pads[3] - kw) // sw + 1
            if group == 1 and kh == 1 and kw == 1 and sh == 1 and sw == 1:
                L = mf.Layer(mf.OP_PWCONV, mf.ACT_NONE, t_in, mf.NO_TENSOR, cin, cout, 1, 1, 1, 1, 0, 0, h, w, h, w, 0,
                             blob.put(W[:, :, 0, 0].T), blob.put(B))
            elif group == cin and cin_g == 1 and cout == cin:
                L = mf.Layer(mf.OP_DWCONV, mf.ACT_NONE, t_in, mf.NO_TENSOR, cin, cout, kh, kw, sh, sw, pt, pl, h, w, oh, ow, 0,
                             blob.put(W[:, 0].transpose(1, 2, 0)), blob.put(B))
            elif group == 1:
                L = mf.Layer(mf.OP_CONV, mf.ACT_NONE, t_in, mf.NO_TENSOR, cin, cout, kh, kw, sh, sw, pt, pl, h, w, oh, ow,
                             1 if t_in == 0 else 0, blob.put(W.transpose(2, 3, 1, 0)), blob.put(B))
            else:
                raise ConvertError(f"Conv {n.name!r}: group {group} with {cin} -> {cout} channels")
            layers.append(L)
            tmap[n.outputs[0]] = (len(layers), cout, L.out_h, L.out_w)
            flush_patterns(n.outputs[0])
        elif op == "BatchNormalization":
            t = tmap.get(n.inputs[0])
            L = layers[t[0] - 1] if t and t[0] > 0 else None
            if L is None or L.op not in (mf.OP_CONV, mf.OP_DWCONV, mf.OP_PWCONV) or L.act != mf.ACT_NONE:
                raise ConvertError("BatchNormalization that does not follow a convolution directly")
            # folding rewrites the convolution's weights: a residual already folded into the layer (BN(conv + x)) or another
            # reader of the convolution's output (a skip taken before the BN) would silently see the wrong values
            if L.res_tensor != mf.NO_TENSOR:
                raise ConvertError("BatchNormalization after a residual Add (BN(conv + x)) cannot be folded into the convolution")
            if sum(n.inputs[0] in q.inputs for q in g.nodes) != 1 or n.inputs[0] in graph_out:
                raise ConvertError("BatchNormalization of a convolution output that has other readers cannot be folded")
            gamma, beta, mean, var = (g.initializers[a].astype(np.float64) for a in n.inputs[1:5])
            scale = gamma / np.sqrt(var + float(n.attrs.get("epsilon", 1e-5)))
            nw = {mf.OP_CONV: L.kh * L.kw * L.cin * L.cout, mf.OP_DWCONV: L.kh * L.kw * L.cout, mf.OP_PWCONV: L.cin * L.cout}[L.op]
            arr = blob.array()
            wv = arr[L.w_off: L.w_off + nw].reshape(-1, L.cout) * scale.astype(np.float32)     # cout is the last axis in all three layouts
            bv = ((arr[L.b_off: L.b_off + L.cout] - mean) * scale + beta).astype(np.float32)
            arr[L.w_off: L.w_off + nw] = wv.reshape(-1)
            arr[L.b_off: L.b_off + L.cout] = bv
            blob.chunks, blob.off = [arr], arr.size
            tmap[n.outputs[0]] = t
            flush_patterns(n.outputs[0])
        elif op in ("Relu", "Gelu", "Clip"):
            if op == "Clip":
                lo = _scalar(g, n.inputs[1]) if len(n.inputs) > 1 else n.attrs.get("min")
                hi = _scalar(g, n.inputs[2]) if len(n.inputs) > 2 else n.attrs.get("max")
                if lo != 0.0 or hi != 6.0:
                    raise ConvertError(f"Clip({lo}, {hi}) is not ReLU6")
            if op == "Gelu" and n.attrs.get("approximate", "none") not in ("none", b"none"):
                act = mf.ACT_GELU_TANH
            else:
                act = {"Relu": mf.ACT_RELU, "Gelu": mf.ACT_GELU_ERF, "Clip": mf.ACT_RELU6}[op]
            set_act(n.inputs[0], n.outputs[0], act)
        elif op == "Add":
            a, b = (tmap.get(x) for x in n.inputs)
            if a is None or b is None:
                # MatMul + Add(bias): handled with the MatMul below
                raise ConvertError(f"Add of {n.inputs}: operands are not both activations on the path")
            # residual: one operand is a 1x1 conv's output (latest layer, no residual yet), the other an older tensor
            for y, r in ((a, b), (b, a)):
                L = layers[y[0] - 1] if y[0] > 0 else None
                if L is not None and L.op == mf.OP_PWCONV and L.res_tensor == mf.NO_TENSOR and r[0] < y[0] and r[1:] == y[1:]:
                    L.res_tensor = r[0]
                    tmap[n.outputs[0]] = y
                    break
            else:
                raise ConvertError(f"Add of {n.inputs}: no 1x1 convolution to fold the residual into")
        elif op in ("GlobalAveragePool", "ReduceMean"):
            t = tmap.get(n.inputs[0])
            if t is None:
                raise ConvertError(f"{op}: input not on the path")
            if op == "ReduceMean":
                axes = n.attrs.get("axes")
                if axes is None and len(n.inputs) > 1:
                    axes = g.initializers[n.inputs[1]].tolist()
                if sorted(a % 4 for a in axes) != [2, 3]:
                    raise ConvertError("ReduceMean over axes other than H, W")
            layers.append(mf.Layer(mf.OP_GAP, mf.ACT_NONE, t[0], mf.NO_TENSOR, t[1], t[1], t[2], t[3], 1, 1, 0, 0, t[2], t[3], 1, 1))
            tmap[n.outputs[0]] = (len(layers), t[1], 1, 1)
            emb_tensor, emb_dim = len(layers), t[1]
        elif op in ("Flatten", "Reshape", "Squeeze", "Identity", "Dropout"):
            t = tmap.get(n.inputs[0])
            if t is None:
                raise ConvertError(f"{op}: input not on the path")
            if t[2] * t[3] != 1:
                raise ConvertError(f"{op} of a {t[2]}x{t[3]} map (only after the global pool)")
            tmap[n.outputs[0]] = t
        elif op in ("Gemm", "MatMul"):
            t = tmap.get(n.inputs[0])
            W = g.initializers.get(n.inputs[1])
            if t is None or W is None or W.ndim != 2:
                raise ConvertError(f"{op}: needs an activation on the path and a 2-d weight initializer")
            if op == "Gemm" and int(n.attrs.get("transB", 0)):
                W = W.T
            if op == "Gemm" and (float(n.attrs.get("alpha", 1.0)) != 1.0 or float(n.attrs.get("beta", 1.0)) != 1.0 or int(n.attrs.get("transA", 0))):
                raise ConvertError("Gemm with alpha / beta / transA")
            cin, cout = W.shape
            if cin != t[1]:
                raise ConvertError(f"{op}: {cin} input features, activation has {t[1]}")
            B = g.initializers.get(n.inputs[2]) if op == "Gemm" and len(n.inputs) > 2 else None
            out = n.outputs[0]
            if B is None:   # MatMul followed by Add(bias)
                nxt = [k for k, nn in enumerate(g.nodes) if nn.op_type == "Add" and out in nn.inputs]
                if len(nxt) == 1:
                    other = [a for a in g.nodes[nxt[0]].inputs if a != out][0]
                    if other in g.initializers and g.initializers[other].size == cout:
                        B, out = g.initializers[other], g.nodes[nxt[0]].outputs[0]
                        skip.add(nxt[0])
            if B is None:
                B = np.zeros(cout, np.float32)
            layers.append(mf.Layer(mf.OP_DENSE, mf.ACT_NONE, t[0], mf.NO_TENSOR, cin, cout, 1, 1, 1, 1, 0, 0, 1, 1, 1, 1, 0,
                                   blob.put(W), blob.put(B.reshape(-1))))
            tmap[out] = (len(layers), cout, 1, 1)
            flush_patterns(out)
        elif op == "Sigmoid" and n.outputs[0] not in graph_out:
            # the gate of a squeeze-excite block: Sigmoid on a pooled [N, C, 1, 1] tensor, consumed by a Mul with the feature map
            t = tmap.get(n.inputs[0])
            if t is None or t[2] * t[3] != 1:
                raise ConvertError("Sigmoid inside the graph that is neither Sigmoid * x nor a squeeze-excite gate")
            set_act(n.inputs[0], n.outputs[0], mf.ACT_SIGMOID)
        elif op == "Mul":
            a, b = (tmap.get(x) for x in n.inputs)
            if a is None or b is None:
                raise ConvertError(f"Mul of {n.inputs}: operands are not both activations on the path")
            for fm, gate in ((a, b), (b, a)):
                if fm[2] * fm[3] > 1 and gate[2] * gate[3] == 1 and gate[1] == fm[1]:
                    layers.append(mf.Layer(mf.OP_SCALE, mf.ACT_NONE, fm[0], gate[0], fm[1], fm[1], 1, 1, 1, 1, 0, 0, fm[2], fm[3], fm[2], fm[3]))
                    tmap[n.outputs[0]] = (len(layers), fm[1], fm[2], fm[3])
                    break
            else:
                raise ConvertError(f"Mul of {n.inputs}: not a feature map times a [N, C, 1, 1] gate")
        elif op in ("Sigmoid", "Softmax"):
            if n.outputs[0] not in graph_out:
                raise ConvertError(f"{op} inside the graph (only the output activation is supported, or Sigmoid * x)")
            out_act = mf.OUT_SIGMOID if op == "Sigmoid" else mf.OUT_SOFTMAX
            tmap[n.outputs[0]] = tmap[n.inputs[0]]
        else:
            raise ConvertError(f"unsupported operator {op} ({n.name!r})")
    if first_of_pattern:
        raise ConvertError(f"activation patterns on tensors that were never produced: {sorted(first_of_pattern)}")
    if not layers or layers[-1].op != mf.OP_DENSE:
        raise ConvertError("the graph does not end in a dense layer")
    final = [tmap.get(o.name) for o in g.outputs]
    if not final or final[0] is None or final[0][0] != len(layers):
        raise ConvertError("the graph output is not the last layer's output")
    return mf.Model(frontend.family, frontend.sample_rate, frontend.sample_count, frontend.segment_duration,
                    layers[-1].cout, emb_dim, out_act, emb_tensor, frontend.spec_h, frontend.spec_w, frontend.norm_eps,
                    [mf.Branch(**{k: getattr(b, k) for k in ("frame_length", "frame_step", "n_mels", "n_frames", "fmin", "fmax",
                                                             "mag_scale", "out_scale", "out_shift", "flags", "mel_w_off")})
                     for b in frontend.branches], layers, blob.array())


def convert_file(onnx_path: str, frontend_bhm: Optional[str], out_path: str, spectrogram_input: Optional[str] = None,
                 sample_rate: Optional[int] = None, family: int = 0) -> mf.Model:
    """ONNX file (+ a BHM1 file carrying the family's front-end, or none and a sample rate: the front-end is then read
    off the graph) -> BHM1 model."""
    m = model_from_graph(ox.load(open(onnx_path, "rb").read()), mf.read_model(frontend_bhm) if frontend_bhm else None,
                         spectrogram_input, sample_rate, family)
    mf.write_model(out_path, m)
    return m


def dense_stack_from_graph(g: ox.Graph, fold_final_sigmoid: bool = True) -> mf.CustomClassifierModel:
    """A dense-stack ONNX graph (the BirdNET geomodel's shape: Gemm / MatMul + Add, Relu, a final Sigmoid or Softmax; the
    reference's fixture tests/fixtures/fixture-geomodel.onnx is Gemm(3 -> 5) + Sigmoid) -> BHC1.  The Python twin of
    birda_amd/csrc/onnx_dense.hpp (which reads the .onnx file inside the library): same chain walk, same refusals.  A final
    Sigmoid becomes the last layer's activation, so every class's score leaves the GEMM activated (bh_range_filter_*)."""
    data_inputs = [vi.name for vi in g.inputs]
    if len(data_inputs) != 1 or len(g.outputs) != 1:
        raise ConvertError("a dense stack has one data input and one output")
    cur, out_name = data_inputs[0], g.outputs[0].name
    layers: List[mf.CustomLayer] = []
    out_act = mf.OUT_NONE
    todo = list(g.nodes)
    while cur != out_name:
        node = next((n for n in todo if n.inputs and n.inputs[0] == cur), None)
        if node is None:
            raise ConvertError(f"graph is not a chain from its input to its output (at {cur!r})")
        todo.remove(node)
        if out_act != mf.OUT_NONE:
            raise ConvertError(f"operator {node.op_type!r} after the output activation")
        op = node.op_type
        if op in ("Gemm", "MatMul"):
            w = g.initializers.get(node.inputs[1]) if len(node.inputs) > 1 else None
            if w is None or w.ndim != 2:
                raise ConvertError(f"{op}: weight must be a 2-D initializer")
            w = np.asarray(w, np.float32)
            alpha = np.float32(node.attrs.get("alpha", 1.0)) if op == "Gemm" else np.float32(1.0)
            beta = np.float32(node.attrs.get("beta", 1.0)) if op == "Gemm" else np.float32(1.0)
            if op == "Gemm" and node.attrs.get("transA", 0):
                raise ConvertError("Gemm: transA is not a dense layer")
            if op == "Gemm" and node.attrs.get("transB", 0):
                w = w.T
            w = (alpha * w).astype(np.float32)
            b = np.zeros(w.shape[1], np.float32)
            if op == "Gemm" and len(node.inputs) > 2 and node.inputs[2]:
                bias = np.asarray(g.initializers[node.inputs[2]], np.float32).reshape(-1)
                b = (beta * np.broadcast_to(bias, (w.shape[1],))).astype(np.float32)
            if layers and layers[-1].w.shape[1] != w.shape[0]:
                raise ConvertError(f"{op}: layer widths do not chain")
            layers.append(mf.CustomLayer(w, b, mf.ACT_NONE))
        elif op == "Add":
            bias = g.initializers.get(node.inputs[1]) if len(node.inputs) == 2 else None
            if not layers or layers[-1].act != mf.ACT_NONE or bias is None or bias.size != layers[-1].w.shape[1]:
                raise ConvertError("Add: only a bias right after MatMul / Gemm is a dense layer")
            layers[-1].b = (layers[-1].b + np.asarray(bias, np.float32).reshape(-1)).astype(np.float32)
        elif op in ("Relu", "Sigmoid"):
            if not layers or layers[-1].act != mf.ACT_NONE:
                raise ConvertError(f"{op} without a dense layer in front of it")
            layers[-1].act = mf.ACT_RELU if op == "Relu" else mf.ACT_SIGMOID
        elif op == "Softmax":
            if not layers or layers[-1].act != mf.ACT_NONE:
                raise ConvertError("Softmax without a dense layer in front of it")
            out_act = mf.OUT_SOFTMAX
        elif op in ("Identity", "Flatten", "Dropout"):
            pass
        else:
            raise ConvertError(f"operator {op!r} is not part of a dense stack (Gemm, MatMul + Add, Relu, Sigmoid, Softmax)")
        cur = node.outputs[0]
    if not layers:
        raise ConvertError("no dense layer between input and output")
    if not fold_final_sigmoid and layers[-1].act == mf.ACT_SIGMOID:
        layers[-1].act, out_act = mf.ACT_NONE, mf.OUT_SIGMOID
    return mf.CustomClassifierModel(int(layers[0].w.shape[0]), out_act, layers)


def convert_geomodel_file(onnx_path: str, out_path: str) -> mf.CustomClassifierModel:
    """geomodel .onnx -> BHC1 (bh_range_filter_create takes either file)."""
    m = dense_stack_from_graph(ox.load(open(onnx_path, "rb").read()))
    mf.write_custom_classifier(out_path, m)
    return m

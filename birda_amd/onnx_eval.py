"""A small numpy evaluator for ONNX graphs (float64 arithmetic): the converter's instrument for reading a model's
spectrogram front-end off the graph by PROBING it instead of guessing how an exporter spelled it.

`convert.recover_frontend` feeds probe signals through the nodes between the audio input and the first 2-D
convolution and fits the front-end parameters of the BHM1 container (frame length / step, the folded
window x DFT x mel operator, exponent, affine, flip) to the responses; the last step re-runs the sub-graph on
random audio and compares it with the closed form the device kernels implement.  What the evaluator needs to
know is therefore the operator SET of such a front-end, not the way the operators are combined.

Only the ops a signal front-end is built from are implemented (element-wise arithmetic, reductions, shape ops,
Conv, MatMul / Gemm, STFT, BatchNormalization); anything else raises `EvalError` naming the op, and the converter
then asks for a front-end manifest instead.  The conv stack is never evaluated here.

Reference: the graph is what the reference hands to ONNX Runtime (src/inference/classifier.rs:269-283); this
evaluator is converter tooling (offline, CPU), not part of the inference path.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

from . import onnx_io as ox


class EvalError(ValueError):
    pass


_F = np.float64


def _is_int(a: np.ndarray) -> bool:
    return np.issubdtype(np.asarray(a).dtype, np.integer)


def _ints(a) -> List[int]:
    return [int(v) for v in np.asarray(a).reshape(-1)]


def _conv(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray], attrs: dict) -> np.ndarray:
    """N-d convolution (1-D and 2-D are what occurs), group 1 or depthwise, dilation 1."""
    nd = x.ndim - 2
    strides = list(attrs.get("strides", [1] * nd))
    pads = list(attrs.get("pads", [0] * (2 * nd)))
    dil = list(attrs.get("dilations", [1] * nd))
    group = int(attrs.get("group", 1))
    auto = attrs.get("auto_pad", "NOTSET")
    if any(d != 1 for d in dil):
        raise EvalError("Conv: dilation")
    k = w.shape[2:]
    if auto in ("SAME_UPPER", "SAME_LOWER"):
        pads = [0] * (2 * nd)
        for i in range(nd):
            out = -(-x.shape[2 + i] // strides[i])
            tot = max((out - 1) * strides[i] + k[i] - x.shape[2 + i], 0)
            lo = tot // 2 if auto == "SAME_UPPER" else tot - tot // 2
            pads[i], pads[nd + i] = lo, tot - lo
    elif auto not in ("NOTSET", "VALID"):
        raise EvalError(f"Conv: auto_pad {auto}")
    if any(pads):
        x = np.pad(x, [(0, 0), (0, 0)] + [(pads[i], pads[nd + i]) for i in range(nd)])
    win = np.lib.stride_tricks.sliding_window_view(x, k, axis=tuple(range(2, 2 + nd)))   # [N, C, o1.., k1..]
    win = win[(slice(None), slice(None)) + tuple(slice(None, None, s) for s in strides)]
    sp = "abc"[:nd]
    kk = "xyz"[:nd]
    if group == 1:
        y = np.einsum(f"nc{sp}{kk},mc{kk}->nm{sp}", win, w, optimize=True)
    elif group == x.shape[1] and w.shape[1] == 1:
        mult = w.shape[0] // group
        wg = w.reshape((group, mult) + tuple(k))
        y = np.einsum(f"nc{sp}{kk},cm{kk}->ncm{sp}", win, wg, optimize=True)
        y = y.reshape((x.shape[0], group * mult) + y.shape[3:])
    else:
        raise EvalError("Conv: grouped (non-depthwise) convolution")
    if b is not None:
        y = y + b.reshape((1, -1) + (1,) * nd)
    return y


def _stft(signal: np.ndarray, step: int, window: Optional[np.ndarray], length: Optional[int], onesided: int) -> np.ndarray:
    """ONNX STFT (opset 17): signal [N, S, 1] (real) -> [N, frames, bins, 2]."""
    if signal.ndim == 3:
        if signal.shape[2] != 1:
            raise EvalError("STFT: complex input")
        signal = signal[:, :, 0]
    L = int(length) if length is not None else int(window.shape[0])
    w = np.ones(L, _F) if window is None else np.asarray(window, _F)
    if w.shape[0] != L:
        raise EvalError("STFT: window length differs from frame_length")
    fr = np.lib.stride_tricks.sliding_window_view(signal, L, axis=1)[:, ::step]          # [N, frames, L]
    bins = L // 2 + 1 if onesided else L
    n = np.arange(L)[:, None] * np.arange(bins)[None, :]
    ang = 2.0 * np.pi * (n % L) / L
    xw = fr * w
    return np.stack([xw @ np.cos(ang), -(xw @ np.sin(ang))], axis=-1)


class Evaluator:
    """Evaluates tensors of a graph on demand.  `run(feeds, targets)`: `feeds` may name ANY tensor (graph inputs or
    intermediate ones); only the ancestors of `targets` below the fed tensors are computed."""

    def __init__(self, g: ox.Graph):
        self.g = g
        self.prod: Dict[str, int] = {o: i for i, n in enumerate(g.nodes) for o in n.outputs if o}
        self.const: Dict[str, np.ndarray] = {}
        for k, a in g.initializers.items():
            self.const[k] = a.astype(_F) if np.issubdtype(a.dtype, np.floating) else a

    # -- graph queries -------------------------------------------------------------------
    def ancestors(self, targets: Iterable[str], stop: Iterable[str] = ()) -> List[int]:
        """Node indices needed for `targets`, in graph order, not looking behind the tensors in `stop`."""
        stop = set(stop)
        need, stack = set(), [t for t in targets]
        while stack:
            t = stack.pop()
            if t in stop or t in self.const or t not in self.prod:
                continue
            i = self.prod[t]
            if i in need:
                continue
            need.add(i)
            stack.extend(x for x in self.g.nodes[i].inputs if x)
        return sorted(need)

    def depends_on(self, tensor: str, source: str) -> bool:
        if tensor == source:
            return True
        return any(source in self.g.nodes[i].inputs for i in self.ancestors([tensor]))

    # -- evaluation ----------------------------------------------------------------------
    def run(self, feeds: Dict[str, np.ndarray], targets: Sequence[str]) -> List[np.ndarray]:
        env: Dict[str, np.ndarray] = dict(self.const)
        for k, v in feeds.items():
            v = np.asarray(v)
            env[k] = v.astype(_F) if np.issubdtype(v.dtype, np.floating) else v
        for i in self.ancestors(targets, stop=feeds.keys()):
            n = self.g.nodes[i]
            try:
                ins = [env[x] if x else None for x in n.inputs]
            except KeyError as e:
                raise EvalError(f"node {n.name or n.op_type}: input {e.args[0]!r} is neither fed nor computed") from None
            outs = self._op(n, ins)
            for o, v in zip(n.outputs, outs):
                if o:
                    env[o] = v
        try:
            return [env[t] for t in targets]
        except KeyError as e:
            raise EvalError(f"tensor {e.args[0]!r} is not in the graph") from None

    def _op(self, n: ox.Node, x: List[Optional[np.ndarray]]) -> List[np.ndarray]:
        op, a = n.op_type, n.attrs
        f = getattr(self, "_op_" + op, None)
        if f is None:
            raise EvalError(f"operator {op} (node {n.name or n.outputs[0]!r}) is outside the front-end operator set")
        r = f(x, a)
        return r if isinstance(r, list) else [r]

    # element-wise
    def _op_Identity(self, x, a): return x[0]
    def _op_Add(self, x, a): return x[0] + x[1]
    def _op_Sub(self, x, a): return x[0] - x[1]
    def _op_Mul(self, x, a): return x[0] * x[1]
    def _op_Neg(self, x, a): return -x[0]
    def _op_Abs(self, x, a): return np.abs(x[0])
    def _op_Sqrt(self, x, a): return np.sqrt(x[0])
    def _op_Exp(self, x, a): return np.exp(x[0])
    def _op_Log(self, x, a):
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.log(x[0])
    def _op_Cos(self, x, a): return np.cos(x[0])
    def _op_Sin(self, x, a): return np.sin(x[0])
    def _op_Reciprocal(self, x, a): return 1.0 / x[0]
    def _op_Relu(self, x, a): return np.maximum(x[0], 0.0)
    def _op_Sigmoid(self, x, a): return 1.0 / (1.0 + np.exp(-x[0]))
    def _op_Softplus(self, x, a): return np.logaddexp(x[0], 0.0)
    def _op_Tanh(self, x, a): return np.tanh(x[0])
    def _op_Min(self, x, a):
        r = x[0]
        for v in x[1:]:
            r = np.minimum(r, v)
        return r
    def _op_Max(self, x, a):
        r = x[0]
        for v in x[1:]:
            r = np.maximum(r, v)
        return r

    def _op_Div(self, x, a):
        if _is_int(x[0]) and _is_int(x[1]):
            return (np.trunc(x[0] / x[1])).astype(x[0].dtype)
        with np.errstate(divide="ignore", invalid="ignore"):
            return x[0] / x[1]

    def _op_Pow(self, x, a):
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.power(x[0], x[1])

    def _op_Clip(self, x, a):
        lo = x[1] if len(x) > 1 and x[1] is not None else a.get("min")
        hi = x[2] if len(x) > 2 and x[2] is not None else a.get("max")
        r = x[0]
        if lo is not None:
            r = np.maximum(r, lo)
        if hi is not None:
            r = np.minimum(r, hi)
        return r

    def _op_Cast(self, x, a):
        to = int(a.get("to", ox.FLOAT))
        if to in (ox.FLOAT, 11, 10, 16):   # float, double, float16, bfloat16: everything floating is float64 here
            return x[0].astype(_F)
        if to in (ox.INT64, ox.INT32):
            return np.trunc(x[0]).astype(np.int64) if not _is_int(x[0]) else x[0].astype(np.int64)
        if to == 9:
            return x[0] != 0
        raise EvalError(f"Cast to data type {to}")

    def _op_Constant(self, x, a):
        for k in ("value", "value_float", "value_int", "value_floats", "value_ints"):
            if k in a:
                v = np.asarray(a[k])
                return v.astype(_F) if np.issubdtype(v.dtype, np.floating) else v.astype(np.int64)
        raise EvalError("Constant without a value")

    def _op_ConstantOfShape(self, x, a):
        v = np.asarray(a.get("value", np.zeros(1, np.float32))).reshape(-1)[0]
        return np.full(_ints(x[0]), v, _F if np.issubdtype(np.asarray(v).dtype, np.floating) else np.int64)

    # reductions
    def _reduce(self, fn, x, a):
        axes = a.get("axes")
        if len(x) > 1 and x[1] is not None:
            axes = _ints(x[1])
        keep = bool(a.get("keepdims", 1))
        if axes is None or len(axes) == 0:
            if int(a.get("noop_with_empty_axes", 0)) and axes is not None:
                return x[0]
            axes = list(range(x[0].ndim))
        return fn(x[0], axis=tuple(int(v) for v in axes), keepdims=keep)

    def _op_ReduceMin(self, x, a): return self._reduce(np.min, x, a)
    def _op_ReduceMax(self, x, a): return self._reduce(np.max, x, a)
    def _op_ReduceSum(self, x, a): return self._reduce(np.sum, x, a)
    def _op_ReduceMean(self, x, a): return self._reduce(np.mean, x, a)

    # shape ops
    def _op_Shape(self, x, a): return np.asarray(x[0].shape, np.int64)
    def _op_Transpose(self, x, a): return np.transpose(x[0], a.get("perm") or None)
    def _op_Flatten(self, x, a):
        ax = int(a.get("axis", 1))
        return x[0].reshape(int(np.prod(x[0].shape[:ax], dtype=np.int64)), -1)
    def _op_Concat(self, x, a): return np.concatenate(x, axis=int(a["axis"]))
    def _op_Expand(self, x, a): return x[0] * np.ones(_ints(x[1]), x[0].dtype)
    def _op_Tile(self, x, a): return np.tile(x[0], _ints(x[1]))
    def _op_Range(self, x, a): return np.arange(x[0], x[1], x[2])
    def _op_Gather(self, x, a): return np.take(x[0], x[1].astype(np.int64), axis=int(a.get("axis", 0)))

    def _op_Reshape(self, x, a):
        shape = _ints(x[1]) if len(x) > 1 and x[1] is not None else list(a["shape"])
        shape = [x[0].shape[i] if (s == 0 and not int(a.get("allowzero", 0))) else s for i, s in enumerate(shape)]
        return x[0].reshape(shape)

    def _op_Squeeze(self, x, a):
        axes = _ints(x[1]) if len(x) > 1 and x[1] is not None else a.get("axes")
        return np.squeeze(x[0], axis=None if axes is None else tuple(int(v) for v in axes))

    def _op_Unsqueeze(self, x, a):
        axes = _ints(x[1]) if len(x) > 1 and x[1] is not None else list(a["axes"])
        r = x[0]
        nd = r.ndim + len(axes)
        for ax in sorted(v % nd for v in axes):
            r = np.expand_dims(r, ax)
        return r

    def _op_Slice(self, x, a):
        if len(x) > 1:
            starts, ends = _ints(x[1]), _ints(x[2])
            axes = _ints(x[3]) if len(x) > 3 and x[3] is not None else list(range(len(starts)))
            steps = _ints(x[4]) if len(x) > 4 and x[4] is not None else [1] * len(starts)
        else:
            starts, ends = list(a["starts"]), list(a["ends"])
            axes = list(a.get("axes", range(len(starts))))
            steps = [1] * len(starts)
        sl = [slice(None)] * x[0].ndim
        for s, e, ax, st in zip(starts, ends, axes, steps):
            d = x[0].shape[ax]
            # ONNX clamps like Python slicing, except that "past the start" with a negative step is spelled with a huge negative end
            if st < 0 and e < -d:
                e = None
            sl[ax] = slice(s, e, st)
        return x[0][tuple(sl)]

    def _op_Pad(self, x, a):
        mode = a.get("mode", "constant")
        if mode != "constant":
            raise EvalError(f"Pad mode {mode}")
        pads = _ints(x[1]) if len(x) > 1 and x[1] is not None else list(a["pads"])
        val = float(np.asarray(x[2]).reshape(-1)[0]) if len(x) > 2 and x[2] is not None else float(a.get("value", 0.0))
        nd = x[0].ndim
        return np.pad(x[0], [(pads[i], pads[nd + i]) for i in range(nd)], constant_values=val)

    # linear algebra
    def _op_MatMul(self, x, a): return np.matmul(x[0], x[1])

    def _op_Gemm(self, x, a):
        A = x[0].T if int(a.get("transA", 0)) else x[0]
        B = x[1].T if int(a.get("transB", 0)) else x[1]
        r = float(a.get("alpha", 1.0)) * (A @ B)
        if len(x) > 2 and x[2] is not None:
            r = r + float(a.get("beta", 1.0)) * x[2]
        return r

    def _op_Conv(self, x, a): return _conv(x[0], x[1], x[2] if len(x) > 2 else None, a)

    def _op_STFT(self, x, a):
        return _stft(x[0], int(np.asarray(x[1]).reshape(-1)[0]), x[2] if len(x) > 2 else None,
                     int(np.asarray(x[3]).reshape(-1)[0]) if len(x) > 3 and x[3] is not None else None, int(a.get("onesided", 1)))

    def _op_BatchNormalization(self, x, a):
        sh = (1, -1) + (1,) * (x[0].ndim - 2)
        return (x[0] - x[3].reshape(sh)) / np.sqrt(x[4].reshape(sh) + float(a.get("epsilon", 1e-5))) * x[1].reshape(sh) + x[2].reshape(sh)

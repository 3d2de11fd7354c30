"""Dependency-free reader / writer for the subset of ONNX that a BirdNET-style classifier graph uses.

birda loads `birdnet.onnx` through `birdnet_onnx::ClassifierBuilder` (reference
`src/inference/classifier.rs:269-283`); the HIP backend reads a BHM1 container
(`modelfile.py`) produced once per model by `convert.py`.  Neither the `onnx` package nor
`onnxruntime` exists on the build / GPU boxes, so the protobuf wire format is decoded here
directly: ModelProto -> GraphProto -> NodeProto / TensorProto / AttributeProto / ValueInfoProto,
field numbers as published in onnx.proto3 [EXT].  The writer emits the same subset and exists so
that the converter can be tested on graphs generated from the synthetic models.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Tuple, Union

import numpy as np

# TensorProto.DataType
FLOAT, INT64, INT32 = 1, 7, 6
# AttributeProto.AttributeType
A_FLOAT, A_INT, A_STRING, A_TENSOR, A_FLOATS, A_INTS = 1, 2, 3, 4, 6, 7


# ---------------------------------------------------------------------------------------
# protobuf wire format
# ---------------------------------------------------------------------------------------
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def fields(buf: bytes) -> Iterator[Tuple[int, int, Union[int, bytes]]]:
    """(field number, wire type, value) of one message; length-delimited values as bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        no, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield no, wt, v


def _enc_varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(no: int, wt: int) -> bytes:
    return _enc_varint((no << 3) | wt)


def _f_varint(no: int, v: int) -> bytes:
    return _key(no, 0) + _enc_varint(v)


def _f_bytes(no: int, v: bytes) -> bytes:
    return _key(no, 2) + _enc_varint(len(v)) + v


def _f_str(no: int, s: str) -> bytes:
    return _f_bytes(no, s.encode("utf-8"))


def _signed(v: int) -> int:
    return v - (1 << 64) if v >= 1 << 63 else v


def _packed_ints(v: Union[int, bytes], wt: int) -> List[int]:
    if wt == 0:
        return [_signed(v)]
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(_signed(x))
    return out


# ---------------------------------------------------------------------------------------
# ONNX structures
# ---------------------------------------------------------------------------------------
@dataclass
class Node:
    op_type: str
    inputs: List[str]
    outputs: List[str]
    attrs: Dict[str, object] = field(default_factory=dict)
    name: str = ""


@dataclass
class ValueInfo:
    name: str
    elem_type: int = FLOAT
    shape: List[Union[int, str]] = field(default_factory=list)


@dataclass
class Graph:
    nodes: List[Node] = field(default_factory=list)
    initializers: Dict[str, np.ndarray] = field(default_factory=dict)
    inputs: List[ValueInfo] = field(default_factory=list)
    outputs: List[ValueInfo] = field(default_factory=list)
    name: str = "graph"
    opset: int = 17
    producer: str = ""


def _parse_tensor(buf: bytes) -> Tuple[str, np.ndarray]:
    dims: List[int] = []
    dtype, name, raw = FLOAT, "", None
    floats: List[float] = []
    ints: List[int] = []
    for no, wt, v in fields(buf):
        if no == 1:
            dims += _packed_ints(v, wt)
        elif no == 2:
            dtype = v
        elif no == 4:   # float_data (packed or repeated 32-bit)
            floats += list(struct.unpack(f"<{len(v) // 4}f", v))
        elif no in (5, 7):   # int32_data / int64_data
            ints += _packed_ints(v, wt)
        elif no == 8:
            name = v.decode("utf-8")
        elif no == 9:
            raw = v
    np_t = {FLOAT: "<f4", INT64: "<i8", INT32: "<i4"}.get(dtype)
    if np_t is None:
        raise ValueError(f"tensor {name!r}: unsupported data type {dtype}")
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np_t).copy()
    elif dtype == FLOAT:
        arr = np.asarray(floats, np.float32)
    else:
        arr = np.asarray(ints, np_t)
    return name, arr.reshape(dims) if dims else arr.reshape(())


def _parse_attr(buf: bytes) -> Tuple[str, object]:
    name, val, floats, ints = "", None, [], []
    for no, wt, v in fields(buf):
        if no == 1:
            name = v.decode("utf-8")
        elif no == 2:
            val = struct.unpack("<f", v)[0]
        elif no == 3:
            val = _signed(v)
        elif no == 4:
            val = v.decode("utf-8", "replace")
        elif no == 5:
            val = _parse_tensor(v)[1]
        elif no == 7:
            floats += list(struct.unpack(f"<{len(v) // 4}f", v))
        elif no == 8:
            ints += _packed_ints(v, wt)
    if floats:
        val = floats
    elif ints:
        val = ints
    return name, val


def _parse_node(buf: bytes) -> Node:
    n = Node("", [], [])
    for no, wt, v in fields(buf):
        if no == 1:
            n.inputs.append(v.decode("utf-8"))
        elif no == 2:
            n.outputs.append(v.decode("utf-8"))
        elif no == 3:
            n.name = v.decode("utf-8")
        elif no == 4:
            n.op_type = v.decode("utf-8")
        elif no == 5:
            k, a = _parse_attr(v)
            n.attrs[k] = a
    return n


def _parse_value_info(buf: bytes) -> ValueInfo:
    vi = ValueInfo("")
    for no, _, v in fields(buf):
        if no == 1:
            vi.name = v.decode("utf-8")
        elif no == 2:   # TypeProto
            for no2, _, v2 in fields(v):
                if no2 == 1:   # tensor_type
                    for no3, _, v3 in fields(v2):
                        if no3 == 1:
                            vi.elem_type = v3
                        elif no3 == 2:   # TensorShapeProto
                            for no4, _, v4 in fields(v3):
                                if no4 == 1:   # Dimension
                                    d: Union[int, str] = "?"
                                    for no5, _, v5 in fields(v4):
                                        if no5 == 1:
                                            d = _signed(v5)
                                        elif no5 == 2:
                                            d = v5.decode("utf-8")
                                    vi.shape.append(d)
    return vi


def load(data: bytes) -> Graph:
    """Parse a serialized ModelProto."""
    g = Graph()
    gbuf: Optional[bytes] = None
    for no, _, v in fields(data):
        if no == 7:
            gbuf = v
        elif no == 2:
            g.producer = v.decode("utf-8", "replace")
        elif no == 8:   # OperatorSetIdProto {domain, version}
            domain, version = "", None
            for no2, _, v2 in fields(v):
                if no2 == 1:
                    domain = v2.decode("utf-8")
                elif no2 == 2:
                    version = v2
            if domain in ("", "ai.onnx") and version is not None:
                g.opset = int(version)
    if gbuf is None:
        raise ValueError("not an ONNX ModelProto: no graph")
    for no, _, v in fields(gbuf):
        if no == 1:
            g.nodes.append(_parse_node(v))
        elif no == 2:
            g.name = v.decode("utf-8")
        elif no == 5:
            name, arr = _parse_tensor(v)
            g.initializers[name] = arr
        elif no == 11:
            g.inputs.append(_parse_value_info(v))
        elif no == 12:
            g.outputs.append(_parse_value_info(v))
    init = set(g.initializers)
    g.inputs = [vi for vi in g.inputs if vi.name not in init]   # old exporters list initializers as inputs
    return g


# ---------------------------------------------------------------------------------------
# writer
# ---------------------------------------------------------------------------------------
def _ser_tensor(name: str, arr: np.ndarray) -> bytes:
    arr = np.asarray(arr)
    dtype = {np.dtype("float32"): FLOAT, np.dtype("int64"): INT64, np.dtype("int32"): INT32}[arr.dtype]
    out = b"".join(_f_varint(1, int(d)) for d in arr.shape)
    out += _f_varint(2, dtype) + _f_str(8, name) + _f_bytes(9, np.ascontiguousarray(arr).astype(arr.dtype.newbyteorder("<")).tobytes())
    return out


def _ser_attr(name: str, v: object) -> bytes:
    out = _f_str(1, name)
    if isinstance(v, float):
        return out + _key(2, 5) + struct.pack("<f", v) + _f_varint(20, A_FLOAT)
    if isinstance(v, (int, np.integer)):
        return out + _f_varint(3, int(v)) + _f_varint(20, A_INT)
    if isinstance(v, str):
        return out + _f_str(4, v) + _f_varint(20, A_STRING)
    if isinstance(v, np.ndarray):
        return out + _f_bytes(5, _ser_tensor("", v)) + _f_varint(20, A_TENSOR)
    if isinstance(v, (list, tuple)) and v and isinstance(v[0], float):
        return out + _f_bytes(7, struct.pack(f"<{len(v)}f", *v)) + _f_varint(20, A_FLOATS)
    if isinstance(v, (list, tuple)):
        return out + _f_bytes(8, b"".join(_enc_varint(int(x)) for x in v)) + _f_varint(20, A_INTS)
    raise TypeError(f"attribute {name!r}: unsupported value {v!r}")


def _ser_value_info(vi: ValueInfo) -> bytes:
    dims = b""
    for d in vi.shape:
        dims += _f_bytes(1, _f_str(2, d) if isinstance(d, str) else _f_varint(1, int(d)))
    tensor_type = _f_varint(1, vi.elem_type) + _f_bytes(2, dims)
    return _f_str(1, vi.name) + _f_bytes(2, _f_bytes(1, tensor_type))


def dump(g: Graph) -> bytes:
    """Serialize a Graph as a ModelProto (ir_version 8)."""
    gb = b""
    for n in g.nodes:
        nb = b"".join(_f_str(1, s) for s in n.inputs) + b"".join(_f_str(2, s) for s in n.outputs)
        nb += _f_str(3, n.name) + _f_str(4, n.op_type) + b"".join(_f_bytes(5, _ser_attr(k, v)) for k, v in n.attrs.items())
        gb += _f_bytes(1, nb)
    gb += _f_str(2, g.name)
    for name, arr in g.initializers.items():
        gb += _f_bytes(5, _ser_tensor(name, arr))
    for vi in g.inputs:
        gb += _f_bytes(11, _ser_value_info(vi))
    for vi in g.outputs:
        gb += _f_bytes(12, _ser_value_info(vi))
    opset = _f_str(1, "") + _f_varint(2, g.opset)
    return _f_varint(1, 8) + _f_str(2, g.producer or "birda_amd.onnx_io") + _f_bytes(7, gb) + _f_bytes(8, opset)

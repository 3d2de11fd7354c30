"""BHM1 model container: the on-disk model format of the birda HIP hot path.

birda itself loads an ONNX file through `birdnet_onnx::ClassifierBuilder`
(reference `src/inference/classifier.rs:269-283`).  Neither ONNX Runtime nor the
BirdNET weights can exist on the GPU box, so the hot path reads a flat,
mmap-friendly container that states exactly what the ONNX graph would state for
this path: model I/O config (`ModelConfig{sample_rate, segment_duration,
sample_count}`, `classifier.rs:360-377`), the spectrogram front-end parameters
 and the conv stack as a layer table with BN already
folded into weights + bias.

Layout (little endian):
  header   256 B            (HEADER_FMT)
  branches n_branches x 64 B (BRANCH_FMT)   front-end STFT/mel branches
  layers   n_layers  x 128 B (LAYER_FMT)    conv stack, tensor i+1 = output of layer i
  blob     f32[]             at header.blob_offset (256-B aligned)

Tensor 0 is the front-end output [B, n_branches, n_mels, n_frames] (planar);
every other tensor is NHWC [B, out_h, out_w, cout].

The same structs are parsed by `birda_amd/csrc/model.hpp` (product) and by
`oracle/birda_oracle.c` (checker); keep the three in sync.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List

import numpy as np

MAGIC = b"BHM1"
VERSION = 1

OP_CONV, OP_DWCONV, OP_PWCONV, OP_GAP, OP_DENSE, OP_SCALE = 1, 2, 3, 4, 5, 6   # OP_SCALE: x[n,h,w,c] * gate[n,c] (squeeze-excite), gate = res_tensor
ACT_NONE, ACT_RELU, ACT_RELU6, ACT_SWISH, ACT_GELU_ERF, ACT_GELU_TANH, ACT_SIGMOID = range(7)
OUT_NONE, OUT_SIGMOID, OUT_SOFTMAX = 0, 1, 2
NO_TENSOR = 0xFFFFFFFF

HEADER_SIZE, BRANCH_SIZE, LAYER_SIZE = 256, 64, 128
# magic, version, family, sample_rate, sample_count, segment_duration, n_classes,
# embedding_dim, n_branches, n_layers, output_activation, embedding_tensor,
# blob_offset, blob_floats, spec_h, spec_w, norm_eps
HEADER_FMT = "<4sIIIIfIIIIIIQQIIf"
# frame_length, frame_step, fft_length, n_bins, n_mels, n_frames, fmin, fmax,
# mag_scale, out_scale, out_shift, flags, mel_w_off
BRANCH_FMT = "<IIIIIIfffffIQ"
# op, act, in_tensor, res_tensor, cin, cout, kh, kw, sh, sw, pad_t, pad_l,
# in_h, in_w, out_h, out_w, in_layout, reserved, w_off, b_off
LAYER_FMT = "<IIIIIIIIIIIIIIIIIIQQ"


@dataclass
class Branch:
    frame_length: int
    frame_step: int
    n_mels: int
    n_frames: int
    fmin: float
    fmax: float
    mag_scale: float
    out_scale: float = 1.0
    out_shift: float = 0.0
    flags: int = 1  # bit0: reverse the mel axis
    mel_w_off: int = 0

    @property
    def n_bins(self) -> int:
        return self.frame_length // 2 + 1


@dataclass
class Layer:
    op: int
    act: int
    in_tensor: int
    res_tensor: int
    cin: int
    cout: int
    kh: int = 1
    kw: int = 1
    sh: int = 1
    sw: int = 1
    pad_t: int = 0
    pad_l: int = 0
    in_h: int = 1
    in_w: int = 1
    out_h: int = 1
    out_w: int = 1
    in_layout: int = 0  # 0 NHWC, 1 planar NCHW (front-end output)
    w_off: int = 0
    b_off: int = 0


@dataclass
class Model:
    family: int
    sample_rate: int
    sample_count: int
    segment_duration: float
    n_classes: int
    embedding_dim: int
    output_activation: int
    embedding_tensor: int
    spec_h: int
    spec_w: int
    norm_eps: float
    branches: List[Branch] = field(default_factory=list)
    layers: List[Layer] = field(default_factory=list)
    blob: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))

    # -- helpers -----------------------------------------------------------
    def weight(self, off: int, n: int) -> np.ndarray:
        return self.blob[off:off + n]

    def macs_per_segment(self) -> int:
        total = 0
        for L in self.layers:
            px = L.out_h * L.out_w
            if L.op == OP_CONV:
                total += px * L.kh * L.kw * L.cin * L.cout
            elif L.op == OP_DWCONV:
                total += px * L.kh * L.kw * L.cout
            elif L.op in (OP_PWCONV, OP_DENSE):
                total += px * L.cin * L.cout
        return total


def write_model(path: str, m: Model) -> None:
    n_b, n_l = len(m.branches), len(m.layers)
    blob_offset = HEADER_SIZE + n_b * BRANCH_SIZE + n_l * LAYER_SIZE
    blob_offset = (blob_offset + 255) // 256 * 256
    blob = np.ascontiguousarray(m.blob, dtype="<f4")
    hdr = struct.pack(HEADER_FMT, MAGIC, VERSION, m.family, m.sample_rate, m.sample_count,
                      m.segment_duration, m.n_classes, m.embedding_dim, n_b, n_l,
                      m.output_activation, m.embedding_tensor, blob_offset, blob.size,
                      m.spec_h, m.spec_w, m.norm_eps)
    with open(path, "wb") as f:
        f.write(hdr.ljust(HEADER_SIZE, b"\0"))
        for b in m.branches:
            rec = struct.pack(BRANCH_FMT, b.frame_length, b.frame_step, b.frame_length, b.n_bins,
                              b.n_mels, b.n_frames, b.fmin, b.fmax, b.mag_scale, b.out_scale,
                              b.out_shift, b.flags, b.mel_w_off)
            f.write(rec.ljust(BRANCH_SIZE, b"\0"))
        for L in m.layers:
            rec = struct.pack(LAYER_FMT, L.op, L.act, L.in_tensor, L.res_tensor, L.cin, L.cout,
                              L.kh, L.kw, L.sh, L.sw, L.pad_t, L.pad_l, L.in_h, L.in_w,
                              L.out_h, L.out_w, L.in_layout, 0, L.w_off, L.b_off)
            f.write(rec.ljust(LAYER_SIZE, b"\0"))
        f.write(b"\0" * (blob_offset - f.tell()))
        f.write(blob.tobytes())


def read_model(path: str) -> Model:
    with open(path, "rb") as f:
        raw = f.read()
    h = struct.unpack_from(HEADER_FMT, raw, 0)
    if h[0] != MAGIC or h[1] != VERSION:
        raise ValueError(f"{path}: not a BHM1 v{VERSION} model")
    (_, _, family, sr, sc, dur, ncls, emb, n_b, n_l, oact, etens, boff, bfl, sh, sw, eps) = h
    m = Model(family, sr, sc, dur, ncls, emb, oact, etens, sh, sw, eps)
    off = HEADER_SIZE
    for _ in range(n_b):
        r = struct.unpack_from(BRANCH_FMT, raw, off)
        m.branches.append(Branch(r[0], r[1], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
        off += BRANCH_SIZE
    for _ in range(n_l):
        r = struct.unpack_from(LAYER_FMT, raw, off)
        m.layers.append(Layer(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10],
                              r[11], r[12], r[13], r[14], r[15], r[16], r[18], r[19]))
        off += LAYER_SIZE
    m.blob = np.frombuffer(raw, dtype="<f4", count=bfl, offset=boff).copy()
    return m


# --------------------------------------------------------------------------------------------------------
# BHC1: custom classifier on embeddings (reference birdnet_onnx::CustomClassifier, src/lib.rs:883-901)
# header 64 B: magic, version, input_dim, n_layers, n_classes, output_activation, blob_offset, blob_floats
# layers n_layers x 32 B: in_dim, out_dim, act, reserved, w_off, b_off        (W [in][out] row-major, b [out])
# --------------------------------------------------------------------------------------------------------
CUSTOM_MAGIC = b"BHC1"
CUSTOM_HEADER_FMT = "<4sIIIIIQQ"
CUSTOM_LAYER_FMT = "<IIIIQQ"


@dataclass
class CustomLayer:
    w: np.ndarray   # [in, out]
    b: np.ndarray   # [out]
    act: int = ACT_NONE


@dataclass
class CustomClassifierModel:
    input_dim: int
    output_activation: int
    layers: List[CustomLayer] = field(default_factory=list)

    @property
    def n_classes(self) -> int:
        return int(self.layers[-1].w.shape[1])


def write_custom_classifier(path: str, m: CustomClassifierModel) -> None:
    chunks, off, recs = [], 0, []
    for L in m.layers:
        w = np.ascontiguousarray(L.w, np.float32)
        b = np.ascontiguousarray(L.b, np.float32)
        pad = (-off) % 16
        if pad:
            chunks.append(np.zeros(pad, np.float32)); off += pad
        w_off = off; chunks.append(w.ravel()); off += w.size
        pad = (-off) % 16
        if pad:
            chunks.append(np.zeros(pad, np.float32)); off += pad
        b_off = off; chunks.append(b.ravel()); off += b.size
        recs.append(struct.pack(CUSTOM_LAYER_FMT, w.shape[0], w.shape[1], L.act, 0, w_off, b_off))
    blob = np.concatenate(chunks) if chunks else np.zeros(0, np.float32)
    blob_offset = (64 + 32 * len(recs) + 255) // 256 * 256
    hdr = struct.pack(CUSTOM_HEADER_FMT, CUSTOM_MAGIC, 1, m.input_dim, len(recs), m.n_classes, m.output_activation,
                      blob_offset, blob.size)
    with open(path, "wb") as f:
        f.write(hdr.ljust(64, b"\0"))
        for r in recs:
            f.write(r)
        f.write(b"\0" * (blob_offset - 64 - 32 * len(recs)))
        f.write(blob.astype("<f4").tobytes())

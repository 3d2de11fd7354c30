"""Generates tests/golden/* (run in the build container; commits are small data files).

Independent implementations used as the source of truth for the MODEL arithmetic:
  * front-end: numpy float64 (np.fft.rfft on Hann-windowed frames, mel matmul, power law)
  * conv stack: torch CPU float64 (F.conv2d NCHW with explicit pads / groups, F.gelu, mean)
and, for the HOST logic, the reference's own unit-test expectations transcribed as data
(reference_unit_cases.json; each case cites the reference test it comes from).

Nothing under /root/reference is imported or copied: the reference is Rust and its model
arithmetic lives in un-vendored crates.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from birda_amd import modelfile as mf, synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def np_frontend(m: mf.Model, x: np.ndarray) -> np.ndarray:
    x = x.astype(np.float64)
    mn, mx = x.min(), x.max()
    x = ((x - mn) / (mx - mn + np.float64(np.float32(m.norm_eps))) - 0.5) * 2.0
    outs = []
    for br in m.branches:
        L, H = br.frame_length, br.frame_step
        idx = np.arange(L)[None, :] + H * np.arange(br.n_frames)[:, None]
        w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(L) / L)
        S = np.fft.rfft(x[idx] * w, axis=1).real
        W = m.blob[br.mel_w_off: br.mel_w_off + br.n_bins * br.n_mels].reshape(br.n_bins, br.n_mels).astype(np.float64)
        s = (S @ W) ** 2
        s = s ** (1.0 / (1.0 + np.exp(np.float64(np.float32(br.mag_scale)))))
        s = s * np.float64(np.float32(br.out_scale)) + np.float64(np.float32(br.out_shift))
        if br.flags & 1:
            s = s[:, ::-1]
        outs.append(s.T)
    return np.stack(outs)  # [branch, mel, frame]


def torch_act(t, act):
    if act == mf.ACT_NONE:
        return t
    if act == mf.ACT_RELU:
        return F.relu(t)
    if act == mf.ACT_RELU6:
        return F.relu6(t)
    if act == mf.ACT_SWISH:
        return F.silu(t)
    if act == mf.ACT_GELU_ERF:
        return F.gelu(t)
    if act == mf.ACT_GELU_TANH:
        return F.gelu(t, approximate="tanh")
    if act == mf.ACT_SIGMOID:
        return torch.sigmoid(t)
    raise ValueError(act)


def torch_forward(m: mf.Model, seg: np.ndarray, keep=()):
    """float64 NCHW forward of one segment; returns (logits, {tensor_index: NHWC array})."""
    spec = torch.from_numpy(np_frontend(m, seg))[None]  # [1, C, H, W]
    tensors = {0: spec}
    blob = torch.from_numpy(m.blob.astype(np.float64))
    for i, L in enumerate(m.layers):
        x = tensors[L.in_tensor]
        if L.op in (mf.OP_CONV, mf.OP_DWCONV, mf.OP_PWCONV):
            if L.op == mf.OP_CONV:
                w = blob[L.w_off: L.w_off + L.kh * L.kw * L.cin * L.cout].reshape(L.kh, L.kw, L.cin, L.cout).permute(3, 2, 0, 1)
                groups = 1
            elif L.op == mf.OP_DWCONV:
                w = blob[L.w_off: L.w_off + L.kh * L.kw * L.cout].reshape(L.kh, L.kw, L.cout).permute(2, 0, 1)[:, None]
                groups = L.cout
            else:
                w = blob[L.w_off: L.w_off + L.cin * L.cout].reshape(L.cin, L.cout).t()[:, :, None, None]
                groups = 1
            b = blob[L.b_off: L.b_off + L.cout]
            pad_b = max((L.out_h - 1) * L.sh + L.kh - L.in_h - L.pad_t, 0)
            pad_r = max((L.out_w - 1) * L.sw + L.kw - L.in_w - L.pad_l, 0)
            xp = F.pad(x, (L.pad_l, pad_r, L.pad_t, pad_b))
            y = F.conv2d(xp, w.contiguous(), b, stride=(L.sh, L.sw), groups=groups)
            assert y.shape[2] == L.out_h and y.shape[3] == L.out_w, (i, y.shape)
        elif L.op == mf.OP_GAP:
            y = x.mean(dim=(2, 3), keepdim=True)
        elif L.op == mf.OP_DENSE:
            w = blob[L.w_off: L.w_off + L.cin * L.cout].reshape(L.cin, L.cout)
            b = blob[L.b_off: L.b_off + L.cout]
            y = (x.reshape(1, L.cin) @ w + b).reshape(1, L.cout, 1, 1)
        elif L.op == mf.OP_SCALE:   # squeeze-excite: the feature map times its [C] gate (the gate rides in res_tensor)
            y = x * tensors[L.res_tensor].reshape(1, -1, 1, 1)
        y = torch_act(y, L.act)
        if L.res_tensor != mf.NO_TENSOR and L.op != mf.OP_SCALE:
            y = y + tensors[L.res_tensor]
        tensors[i + 1] = y
    logits = tensors[len(m.layers)].reshape(-1).numpy()
    kept = {t: tensors[t].permute(0, 2, 3, 1).reshape(-1).numpy() if t else tensors[0].reshape(-1).numpy() for t in keep}
    return logits, kept


# The reference's own unit-test expectations for the host side of the path, as data.
REFERENCE_UNIT_CASES = {
    "chunk_audio": [  # reference src/audio/chunker.rs:77-125
        {"src": "chunker.rs:82-88 test_chunk_audio_no_overlap", "n_samples": 96000, "rate": 48000, "dur": 1.0, "ovl": 0.0, "count": 2, "starts": [0.0, 1.0]},
        {"src": "chunker.rs:91-100 test_chunk_audio_with_overlap", "n_samples": 144000, "rate": 48000, "dur": 1.0, "ovl": 0.5, "count": 6, "starts": [0.0, 0.5]},
        {"src": "chunker.rs:103-109 test_chunk_audio_pads_final_chunk", "n_samples": 60000, "rate": 48000, "dur": 1.0, "ovl": 0.0, "count": 2, "starts": [0.0, 1.0]},
        {"src": "chunker.rs:112-116 test_chunk_audio_empty_input", "n_samples": 0, "rate": 48000, "dur": 1.0, "ovl": 0.0, "count": 0, "starts": []},
        {"src": "chunker.rs:119-124 test_chunk_audio_overlap_equals_duration", "n_samples": 96000, "rate": 48000, "dur": 1.0, "ovl": 1.0, "count": 0, "starts": []},
    ],
    "estimate_segment_count": [  # reference src/output/progress.rs:171-185
        {"src": "progress.rs:173", "duration": 10.0, "seg": 3.0, "ovl": 0.0, "expect": 4},
        {"src": "progress.rs:176", "duration": 10.0, "seg": 3.0, "ovl": 1.0, "expect": 5},
        {"src": "progress.rs:179", "duration": None, "seg": 3.0, "ovl": 0.0, "expect": None},
        {"src": "progress.rs:182", "duration": 10.0, "seg": 3.0, "ovl": 3.0, "expect": None},
        {"src": "progress.rs:183", "duration": 10.0, "seg": 3.0, "ovl": 4.0, "expect": None},
    ],
    "detection_from_label": [  # reference src/output/types.rs:83-107
        {"src": "types.rs:87-98", "label": "Passer domesticus_House Sparrow", "scientific": "Passer domesticus", "common": "House Sparrow"},
        {"src": "types.rs:101-106", "label": "Unknown Species", "scientific": "Unknown Species", "common": "Unknown Species"},
    ],
    "csv": {  # reference src/output/csv.rs:134-284
        "header": "Start (s),End (s),Scientific name,Common name,Confidence,File",
        "bom": [239, 187, 191],
        "rows": [
            {"src": "csv.rs:141-160 test_csv_writer_basic", "label": "Passer domesticus_House Sparrow", "conf": 0.8542, "start": 0.0, "end": 3.0,
             "path": "/path/to/audio.wav", "row": "0.0,3.0,Passer domesticus,House Sparrow,0.8542,/path/to/audio.wav"},
        ],
        "escape": [  # csv.rs:249-253 test_escape_csv
            {"in": "simple", "out": "simple"}, {"in": "with,comma", "out": "\"with,comma\""},
            {"in": "with\"quote", "out": "\"with\"\"quote\""},
        ],
    },
    "resample": {  # reference src/audio/resample.rs:117-385 (property tests; constants :119-170)
        "identity": {"src": "resample.rs:354-359", "samples": [0.1, 0.2, 0.3, 0.4, 0.5], "rate": 48000},
        "length_bounds": [
            {"src": "resample.rs:362-372 test_resample_downsample", "n": 48000, "from": 48000, "to": 32000, "gt": 20000, "lt": 35000},
            {"src": "resample.rs:375-384 test_resample_upsample", "n": 32000, "from": 32000, "to": 48000, "gt": 45000, "lt": 55000},
        ],
        "tone_intact": [
            {"src": "resample.rs:240-250", "tone": 1000.0, "from": 48000, "to": 32000, "n": 48000, "others": [500.0, 2000.0, 4000.0]},
            {"src": "resample.rs:253-277", "tone": 6000.0, "from": 48000, "to": 32000, "n": 48000, "others": [3000.0, 9000.0, 12000.0], "rms_floor": 0.6},
            {"src": "resample.rs:329-338", "tone": 6000.0, "from": 44100, "to": 32000, "n": 44100, "others": [3000.0, 9000.0, 12000.0]},
        ],
        "anti_alias": [
            {"src": "resample.rs:280-307", "tone": 20000.0, "from": 48000, "to": 32000, "n": 48000, "alias": 12000.0, "alias_fraction": 1e-6, "rms_ceiling": 0.1},
            {"src": "resample.rs:310-326", "tone": 20000.0, "from": 44100, "to": 32000, "n": 44100, "rms_ceiling": 0.1},
        ],
        "amplitude": {"src": "resample.rs:341-351", "tone": 1000.0, "from": 48000, "to": 32000, "n": 48000, "tol": 0.05},
        "constants": {"min_tone_power_fraction": 0.5, "dominance_ratio": 100.0, "steady_state_margin": 8},
        "fft_chunks": [  # resample.rs:311-316 comment: 342 chunks for 48k->32k, 3 for 44.1k->32k
            {"from": 48000, "to": 32000, "chunks": 342, "fft_in": 1026, "fft_out": 684},
            {"from": 44100, "to": 32000, "chunks": 3, "fft_in": 1323, "fft_out": 960},
        ],
    },
    "segmenter": [  # reference src/audio/decode.rs:150-202 worked by hand
        {"src": "decode.rs:175-196, 10 s / 3 s / no overlap", "n_samples": 480000, "rate": 48000, "seg": 144000, "ovl": 0,
         "starts": [0, 144000, 288000, 432000], "last_real": 48000},
        {"src": "decode.rs:186-196, 10 s / 3 s / 1 s overlap: trailing tail segment", "n_samples": 480000, "rate": 48000, "seg": 144000, "ovl": 48000,
         "starts": [0, 96000, 192000, 288000, 384000, 432000], "last_real": 48000},
        {"src": "decode.rs:170-172 empty stream", "n_samples": 0, "rate": 48000, "seg": 144000, "ovl": 0, "starts": [], "last_real": 0},
        {"src": "processor.rs:514 C1 trace: 30 s -> 10 segments", "n_samples": 1440000, "rate": 48000, "seg": 144000, "ovl": 0,
         "starts": [0, 144000, 288000, 432000, 576000, 720000, 864000, 1008000, 1152000, 1296000], "last_real": 144000},
    ],
    "source_sizing": [  # reference src/pipeline/processor.rs:67-82
        {"target": 144000, "src_rate": 44100, "dst_rate": 48000, "expect": 132300},
        {"target": 144000, "src_rate": 22050, "dst_rate": 48000, "expect": 66150},
        {"target": 144000, "src_rate": 48000, "dst_rate": 48000, "expect": 144000},
        {"target": 160000, "src_rate": 48000, "dst_rate": 32000, "expect": 240000},
    ],
    "batching": [  # reference src/pipeline/processor.rs:531-545 + :240-258 (C1 trace)
        {"batch_size": 8, "estimated": 10, "effective": 8, "segments": 10, "batches": 2, "padded_rows": 6},
        {"batch_size": 16, "estimated": 10, "effective": 10, "segments": 10, "batches": 1, "padded_rows": 0},
        {"batch_size": 8, "estimated": 0, "effective": 8, "segments": 0, "batches": 0, "padded_rows": 0},
    ],
    "scientific_name": [  # reference src/inference/geomodel.rs:198-238
        {"src": "geomodel.rs:199", "label": "Parus major_Great Tit", "expect": "Parus major"},
        {"src": "geomodel.rs:204", "label": "Parus major_Talitiainen", "expect": "Parus major"},
        {"src": "geomodel.rs:209", "label": "Parus major", "expect": "Parus major"},
        {"src": "geomodel.rs:214-217", "label": "Accelerating_and_revving_and_vroom", "expect": "Accelerating_and_revving_and_vroom"},
        {"src": "geomodel.rs:222", "label": "Accordion", "expect": "Accordion"},
        {"src": "geomodel.rs:223", "label": "Dog_Dog", "expect": "Dog_Dog"},
        {"src": "geomodel.rs:228", "label": "Parus major_Great_Tit", "expect": "Parus major"},
        {"src": "geomodel.rs:233", "label": "", "expect": ""},
    ],
    # SpeciesMapping::build + GeomodelScores::project (geomodel.rs:240-409).  "scores": expected score_of() per
    # classifier label, null = None (no geomodel entry).
    "geomodel_projection": [
        {"src": "geomodel.rs:241-256 localized labels", "geomodel": ["Parus major_Great Tit"], "classifier": ["Parus major_Talitiainen"],
         "reported": [], "mapped": 1, "unmatched": 0, "scores": [0.0]},
        {"src": "geomodel.rs:259-266 bare binomial (Perch)", "geomodel": ["Parus major_Great Tit"], "classifier": ["Parus major"],
         "reported": [], "mapped": 1, "unmatched": 0, "scores": [0.0]},
        {"src": "geomodel.rs:269-276 case-insensitive", "geomodel": ["parus major_Great Tit"], "classifier": ["Parus Major_Talitiainen"],
         "reported": [], "mapped": 1, "unmatched": 0, "scores": [0.0]},
        {"src": "geomodel.rs:279-296 unmatched count", "geomodel": ["Parus major_Great Tit"],
         "classifier": ["Parus major_Great Tit", "Accipiter gentilis_Northern Goshawk", "Dog_Dog"],
         "reported": [], "mapped": 1, "unmatched": 2, "scores": [0.0, None, None]},
        {"src": "geomodel.rs:299-312 geomodel-only species ignored",
         "geomodel": ["Parus major_Great Tit", "Petaurista albiventer_White-bellied Giant Flying Squirrel"],
         "classifier": ["Parus major_Great Tit"], "reported": [], "mapped": 1, "unmatched": 0, "scores": [0.0]},
        {"src": "geomodel.rs:315-326 first of two colliding classifier labels", "geomodel": ["Parus major_Great Tit"],
         "classifier": ["Parus major_First", "Parus major_Second"], "reported": [], "mapped": 1, "unmatched": 1, "scores": [0.0, None]},
        {"src": "geomodel.rs:329-335 empty sets", "geomodel": [], "classifier": [], "reported": [], "mapped": 0, "unmatched": 0, "scores": []},
        {"src": "geomodel.rs:338-353 keyed by classifier label", "geomodel": ["Parus major_Great Tit"], "classifier": ["Parus major_Talitiainen"],
         "reported": [["Parus major_Great Tit", 0.8]], "mapped": 1, "unmatched": 0, "scores": [0.8]},
        {"src": "geomodel.rs:356-367 omitted species reads 0", "geomodel": ["Parus major_Great Tit"], "classifier": ["Parus major_Great Tit"],
         "reported": [], "mapped": 1, "unmatched": 0, "scores": [0.0]},
        {"src": "geomodel.rs:370-378 unmatched omitted", "geomodel": ["Parus major_Great Tit"], "classifier": ["Dog_Dog"],
         "reported": [], "mapped": 0, "unmatched": 1, "scores": [None]},
        {"src": "geomodel.rs:381-397 species without classifier match dropped",
         "geomodel": ["Parus major_Great Tit", "Vulpes vulpes_Red Fox"], "classifier": ["Parus major_Great Tit"],
         "reported": [["Parus major_Great Tit", 0.8], ["Vulpes vulpes_Red Fox", 0.9]], "mapped": 1, "unmatched": 0, "scores": [0.8]},
        {"src": "geomodel.rs:400-417 in_range_count", "geomodel": ["Aaa aaa_X", "Bbb bbb_Y", "Ccc ccc_Z"],
         "classifier": ["Aaa aaa_X", "Bbb bbb_Y", "Ccc ccc_Z"],
         "reported": [["Aaa aaa_X", 0.9], ["Bbb bbb_Y", 0.005], ["Ccc ccc_Z", 0.02]], "mapped": 3, "unmatched": 0,
         "scores": [0.9, 0.005, 0.02], "in_range": [[0.01, 2], [0.5, 1], [0.99, 0]]},
    ],
    # filter_predictions (geomodel_filter.rs:84-294).  "table": score per species that has a geomodel entry;
    # "in" / "out": [species, confidence] in order.  threshold 0.01 throughout (:117-123).
    "filter_predictions": [
        {"src": "geomodel_filter.rs:126-140", "table": {"Parus major_x": 0.5}, "policy": "keep", "rerank": False,
         "in": [["Parus major_x", 0.8]], "out": [["Parus major_x", 0.8]]},
        {"src": "geomodel_filter.rs:143-153", "table": {"Parus major_x": 0.005}, "policy": "keep", "rerank": False,
         "in": [["Parus major_x", 0.9]], "out": []},
        {"src": "geomodel_filter.rs:156-166 inclusive threshold", "table": {"Parus major_x": 0.01}, "policy": "keep", "rerank": False,
         "in": [["Parus major_x", 0.9]], "out": [["Parus major_x", 0.9]]},
        {"src": "geomodel_filter.rs:169-180", "table": {"Parus major_x": 0.5}, "policy": "keep", "rerank": False,
         "in": [["Dog_Dog", 0.7]], "out": [["Dog_Dog", 0.7]]},
        {"src": "geomodel_filter.rs:183-193", "table": {"Parus major_x": 0.5}, "policy": "drop", "rerank": False,
         "in": [["Dog_Dog", 0.7]], "out": []},
        {"src": "geomodel_filter.rs:196-206", "table": {"Parus major_x": 0.5}, "policy": "keep", "rerank": True,
         "in": [["Parus major_x", 0.8]], "out": [["Parus major_x", 0.4]]},
        {"src": "geomodel_filter.rs:209-222", "table": {"Parus major_x": 0.5}, "policy": "keep", "rerank": True,
         "in": [["Parus major_x", 0.8], ["Dog_Dog", 0.9]], "out": [["Parus major_x", 0.4]]},
        {"src": "geomodel_filter.rs:225-245", "table": {"Parus major_x": 0.9, "Rara avis_y": 0.02}, "policy": "keep", "rerank": True,
         "in": [["Rara avis_y", 0.80], ["Parus major_x", 0.70]], "out": [["Parus major_x", 0.63], ["Rara avis_y", 0.016]]},
        {"src": "geomodel_filter.rs:248-259", "table": {"Aaa aaa_x": 0.2, "Bbb bbb_y": 0.9}, "policy": "keep", "rerank": False,
         "in": [["Aaa aaa_x", 0.5], ["Bbb bbb_y", 0.4]], "out": [["Aaa aaa_x", 0.5], ["Bbb bbb_y", 0.4]]},
        {"src": "geomodel_filter.rs:278-284", "table": {"Aaa aaa_x": 0.9}, "policy": "keep", "rerank": True, "in": [], "out": []},
        {"src": "geomodel_filter.rs:287-300", "table": {}, "policy": "keep", "rerank": False,
         "in": [["Dog_Dog", 0.7], ["Siren_Siren", 0.6]], "out": [["Dog_Dog", 0.7], ["Siren_Siren", 0.6]]},
    ],
    "pcm": [  # reference src/audio/decode.rs:353-411
        {"src": "decode.rs:372-374 S16 mono", "fmt": "s16", "channels": 1, "in": [0, 16384, -32768, 32767], "out": [0.0, 0.5, -1.0, 0.999969482421875]},
        {"src": "decode.rs:376-385 S16 stereo mean", "fmt": "s16", "channels": 2, "in": [16384, -16384, 32767, 32767], "out": [0.0, 0.999969482421875]},
        {"src": "decode.rs:388-391 S32 mono", "fmt": "s32", "channels": 1, "in": [1073741824, -2147483648], "out": [0.5, -1.0]},
    ],
}


# The models bench.py and the GPU parity tests actually run: float64 torch / numpy logits of a few segments of
# the FULL synthetic stacks, so that the oracle is not the sole authority for them.  kind -> (segments, first segment's seed index)
FULL_MODELS = {"birdnet_v24": (4, 300), "perch_v2": (3, 310), "birdnet_v30": (3, 320), "mini_se": (3, 330),
               "birdnet_v30_sized": (2, 340)}      # (round 6: the v3.0 contract at the published file's size, 553 MB / 21.5 GFLOP)


def gen_full_models(only=()):
    """`only`: regenerate just these kinds and keep the other models' committed vectors as they are (`full kind ...`)"""
    store = {}
    path = os.path.join(OUT, "full_model_vectors.npz")
    if only and os.path.exists(path):
        store = dict(np.load(path))
    for kind, (n, start) in FULL_MODELS.items():
        if only and kind not in only:
            continue
        m = synth.build_model(kind)
        segs = synth.synth_segments(n, m.sample_count, m.sample_rate, start=start)
        rows = []
        for s in segs:
            lg, _ = torch_forward(m, s)
            rows.append(lg)
        store[f"{kind}_logits"] = np.array(rows, np.float32)
        store[f"{kind}_start"] = np.array([start])
        print(kind, store[f"{kind}_logits"].shape, float(np.abs(store[f"{kind}_logits"]).max()), flush=True)
    np.savez_compressed(os.path.join(OUT, "full_model_vectors.npz"), **store)


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "full":   # (minutes of float64 CPU work: on request)
        gen_full_models(tuple(sys.argv[2:]))
        return
    with open(os.path.join(OUT, "reference_unit_cases.json"), "w") as f:
        json.dump(REFERENCE_UNIT_CASES, f, indent=1)

    rng = np.random.default_rng(7)
    # (1) Hann-window rFFT of seeded noise + sine frames, n_fft 2048 / 1024
    fft = {}
    for L in (2048, 1024):
        x = rng.standard_normal(L) * 0.1 + np.sin(2 * np.pi * 37.0 * np.arange(L) / L)
        w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(L) / L)
        X = np.fft.rfft(x * w)
        fft[f"x{L}"] = x
        fft[f"re{L}"] = X.real
        fft[f"im{L}"] = X.imag
    np.savez_compressed(os.path.join(OUT, "hann_rfft.npz"), **fft)

    # (2)+(3) front-end and full forward on the mini model (float64 numpy/torch)
    store = {}
    mini = synth.build_model("mini")
    segs = synth.synth_segments(4, mini.sample_count, mini.sample_rate)
    store["mini_segments_seed_start"] = np.array([0])
    logits, specs, embs, mid = [], [], [], []
    for s in segs:
        lg, kept = torch_forward(mini, s, keep=(0, 5, mini.embedding_tensor))
        logits.append(lg); specs.append(kept[0]); mid.append(kept[5]); embs.append(kept[mini.embedding_tensor])
    store["mini_logits"] = np.array(logits, np.float32)
    store["mini_spec"] = np.array(specs, np.float32)
    store["mini_tensor5"] = np.array(mid, np.float32)
    store["mini_embedding"] = np.array(embs, np.float32)
    # the real v2.4 front-end geometry with the toy stack: one segment, logits + a strided spectrogram
    tiny = synth.build_model("birdnet_v24_tiny")
    seg = synth.synth_segment(3)
    lg, kept = torch_forward(tiny, seg, keep=(0,))
    store["tiny_logits_seg3"] = lg.astype(np.float32)
    store["tiny_spec_seg3_frames_every7"] = kept[0].reshape(2, 96, 511)[:, :, ::7].astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "model_vectors.npz"), **store)

    # (4) mel filterbank digests (drift detection for the restated HTK matrix)
    mel = {}
    for name, (nm, nb, sr, lo, hi) in {"v24_low": (96, 1025, 48000, 0.0, 3000.0), "v24_high": (96, 513, 48000, 500.0, 15000.0)}.items():
        W = synth.linear_to_mel_weight_matrix(nm, nb, sr, lo, hi)
        nzr = np.nonzero(W.any(axis=1))[0]
        mel[name] = {"shape": list(W.shape), "sum": float(W.astype(np.float64).sum()), "first_nonzero_bin": int(nzr[0]),
                     "last_nonzero_bin": int(nzr[-1]), "col_peaks": [int(i) for i in W.argmax(axis=0)[::12]]}
    with open(os.path.join(OUT, "mel_digests.json"), "w") as f:
        json.dump(mel, f, indent=1)
    print("golden written to", OUT, {k: v.shape for k, v in store.items()})


if __name__ == "__main__":
    main()

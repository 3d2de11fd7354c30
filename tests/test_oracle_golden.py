"""Pins the CPU oracle (oracle/birda_oracle.c): model arithmetic against float64
numpy/torch vectors (tools/gen_golden.py), host logic against the reference's own unit-test
expectations (tests/golden/reference_unit_cases.json)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def cases():
    with open(os.path.join(GOLDEN, "reference_unit_cases.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def vectors():
    return np.load(os.path.join(GOLDEN, "model_vectors.npz"))


# ---------------- model arithmetic ----------------
def test_fft_matches_hann_rfft_vectors(oracle_lib):
    g = np.load(os.path.join(GOLDEN, "hann_rfft.npz"))
    for L in (2048, 1024):
        x = g[f"x{L}"]
        w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(L) / L)
        X = oracle_lib.fft(x * w, -1)[: L // 2 + 1]
        assert np.abs(X.real - g[f"re{L}"]).max() < 1e-9
        assert np.abs(X.imag - g[f"im{L}"]).max() < 1e-9


def test_fft_arbitrary_lengths_match_numpy(oracle_lib):
    rng = np.random.default_rng(1)
    for n in (1, 2, 3, 12, 133, 1368, 2052, 2058, 2240, 2646, 4480):  # the resampler's block sizes
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        assert np.abs(oracle_lib.fft(x, -1) - np.fft.fft(x)).max() < 1e-9 * max(n, 8)
        assert np.abs(oracle_lib.fft(x, +1) - np.fft.ifft(x) * n).max() < 1e-9 * max(n, 8)


def test_mel_matrix_digests():
    from birda_amd import synth
    with open(os.path.join(GOLDEN, "mel_digests.json")) as f:
        dig = json.load(f)
    for name, (nm, nb, sr, lo, hi) in {"v24_low": (96, 1025, 48000, 0.0, 3000.0), "v24_high": (96, 513, 48000, 500.0, 15000.0)}.items():
        W = synth.linear_to_mel_weight_matrix(nm, nb, sr, lo, hi)
        d = dig[name]
        assert list(W.shape) == d["shape"]
        assert abs(float(W.astype(np.float64).sum()) - d["sum"]) < 1e-6
        nzr = np.nonzero(W.any(axis=1))[0]
        assert int(nzr[0]) == d["first_nonzero_bin"] and int(nzr[-1]) == d["last_nonzero_bin"]
        assert [int(i) for i in W.argmax(axis=0)[::12]] == d["col_peaks"]
        assert (W[0] == 0).all()          # DC row zeroed, like tf.signal.linear_to_mel_weight_matrix
        assert W.min() >= 0 and W.max() <= 1.0
    # 0-3 kHz at L=2048 touches bins 1..128 only: the pruning lever of SURVEY.md 8d
    assert dig["v24_low"]["last_nonzero_bin"] <= 128


def test_frontend_and_forward_match_float64_vectors_mini(oracle_lib, model_dir, vectors):
    from birda_amd import synth
    path, _, m, _ = model_dir["mini"]
    om = oracle_lib.OracleModel(path)
    segs = synth.synth_segments(4, m.sample_count, m.sample_rate)
    for i in range(4):
        spec = om.frontend(segs[i])
        # |x|^0.45 is not Lipschitz at 0: an fp32 rounding d in the mel projection moves the
        # output by up to |d|^0.45, so the bound is loose in max and tight in mean
        assert np.abs(spec - vectors["mini_spec"][i]).max() < 2e-3
        assert np.abs(spec - vectors["mini_spec"][i]).mean() < 2e-6
    logits, emb = om.forward(segs, want_embeddings=True)
    assert np.abs(logits - vectors["mini_logits"]).max() < 1e-3
    assert np.abs(emb - vectors["mini_embedding"]).max() < 1e-3
    _, t5 = om.forward(segs, dump_tensor=5)
    assert np.abs(t5 - vectors["mini_tensor5"]).max() < 1e-3


def test_v24_frontend_geometry_matches_float64_vectors(oracle_lib, model_dir, vectors):
    from birda_amd import synth
    path, _, m, _ = model_dir["birdnet_v24_tiny"]
    om = oracle_lib.OracleModel(path)
    seg = synth.synth_segment(3)
    spec = om.frontend(seg).reshape(2, 96, 511)[:, :, ::7]
    ref = vectors["tiny_spec_seg3_frames_every7"]
    assert np.abs(spec - ref).max() < 2e-3 and np.abs(spec - ref).mean() < 2e-6
    logits = om.forward(seg[None])
    assert np.abs(logits[0] - vectors["tiny_logits_seg3"]).max() < 1e-3


# The models bench.py and the GPU parity tests run: float64 torch / numpy logits of a few segments of the FULL
# synthetic stacks (tools/gen_golden.py full -> full_model_vectors.npz), so the oracle is not the sole authority for them.
FULL_GOLDEN_TOL = 2e-5    # of max(1, max |logit|): the f32 oracle against the float64 evaluation (measured 1e-6 ... 4e-6)


@pytest.mark.parametrize("kind", ["birdnet_v24", "perch_v2", "birdnet_v30", "mini_se", "birdnet_v30_sized"])
def test_oracle_matches_float64_vectors_of_the_full_models(oracle_lib, tmp_path, kind):
    from birda_amd import modelfile as mf, synth
    g = np.load(os.path.join(GOLDEN, "full_model_vectors.npz"))
    ref = g[f"{kind}_logits"]
    m = synth.build_model(kind)
    path = str(tmp_path / f"{kind}.bhm")
    mf.write_model(path, m)
    segs = synth.synth_segments(ref.shape[0], m.sample_count, m.sample_rate, start=int(g[f"{kind}_start"][0]))
    got = oracle_lib.OracleModel(path).forward(segs)
    scale = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(got - ref).max())
    print(f"{kind}: oracle vs float64 max|dlogit| = {err:.3e} on max|logit| {scale:.2f}")
    assert got.shape == ref.shape and err <= FULL_GOLDEN_TOL * scale, (kind, err)


def test_topk_semantics(oracle_lib):
    logits = np.array([0.0, 3.0, -1.0, 3.0, 2.0, -20.0], np.float32)
    idx, conf = oracle_lib.topk(logits, 1, 5, 0.1)           # sigmoid, top 5, min conf 0.1
    assert idx.tolist() == [1, 3, 4, 0, 2]                    # tie 1 vs 3 -> lower index; 2 has p=0.269
    assert np.allclose(conf, 1 / (1 + np.exp(-logits[idx])), atol=1e-6)
    idx, conf = oracle_lib.topk(logits, 1, 5, 0.5)
    assert idx.tolist() == [1, 3, 4, 0]                       # p >= 0.5 keeps logit 0.0 exactly
    idx, conf = oracle_lib.topk(logits, 2, 3, 0.0)            # softmax
    assert idx.tolist() == [1, 3, 4] and abs(conf.sum() - np.exp(logits[[1, 3, 4]]).sum() / np.exp(logits).sum()) < 1e-6


# ---------------- host logic vs the reference's unit tests ----------------
def test_chunk_audio_cases(oracle_lib, cases):
    L = oracle_lib.lib()
    for c in cases["chunk_audio"]:
        starts = np.zeros(64, np.float32)
        n = L.bo_chunk_audio_count(c["n_samples"], c["rate"], c["dur"], c["ovl"], starts.ctypes.data, 64)
        assert n == c["count"], c["src"]
        assert starts[: len(c["starts"])].tolist() == c["starts"], c["src"]


def test_estimate_segment_count_cases(oracle_lib, cases):
    L = oracle_lib.lib()
    for c in cases["estimate_segment_count"]:
        v = L.bo_estimate_segment_count(int(c["duration"] is not None), c["duration"] or 0.0, c["seg"], c["ovl"])
        assert (None if v < 0 else v) == c["expect"], c["src"]


def test_label_split_and_csv_rows(oracle_lib, cases):
    L = oracle_lib.lib()
    buf = C.create_string_buffer(8192)
    for c in cases["detection_from_label"]:
        n = L.bo_csv_row(c["label"].encode(), 0.0, 3.0, 0.5, b"f.wav", buf)
        assert buf.raw[:n].decode() == f"0.0,3.0,{c['scientific']},{c['common']},0.5000,f.wav\n", c["src"]
    for r in cases["csv"]["rows"]:
        n = L.bo_csv_row(r["label"].encode(), r["start"], r["end"], r["conf"], r["path"].encode(), buf)
        assert buf.raw[:n].decode() == r["row"] + "\n", r["src"]
    for e in cases["csv"]["escape"]:
        n = L.bo_csv_row(b"a_b", 0.0, 3.0, 0.5, e["in"].encode(), buf)
        assert buf.raw[:n].decode() == f"0.0,3.0,a,b,0.5000,{e['out']}\n"
    n = L.bo_csv_header(1, buf)
    assert list(buf.raw[:3]) == cases["csv"]["bom"] and buf.raw[3:n].decode() == cases["csv"]["header"] + "\n"
    n = L.bo_csv_header(0, buf)
    assert buf.raw[:n].decode() == cases["csv"]["header"] + "\n"


def test_pcm_scaling_cases(oracle_lib, cases):
    L = oracle_lib.lib()
    for c in cases["pcm"]:
        frames = len(c["out"])
        out = np.zeros(frames, np.float32)
        if c["fmt"] == "s16":
            a = np.array(c["in"], np.int16)
            L.bo_pcm16_to_mono(a.ctypes.data, frames, c["channels"], out)
        else:
            a = np.array(c["in"], np.int32)
            L.bo_pcm32_to_mono(a.ctypes.data, frames, c["channels"], out)
        assert out.tolist() == c["out"], c["src"]


def test_segmenter_cases(oracle_lib, cases):
    for c in cases["segmenter"]:
        x = (np.arange(c["n_samples"], dtype=np.float32) % 1000 + 1) / 1000.0
        segs = oracle_lib.segment_stream(x, c["seg"], c["ovl"], packet=1152)
        assert [s for _, s in segs] == c["starts"], c["src"]
        for buf, start in segs:
            real = min(c["seg"], c["n_samples"] - start)
            assert np.array_equal(buf[:real], x[start:start + real])
            assert (buf[real:] == 0).all()
        if segs:
            assert min(c["seg"], c["n_samples"] - segs[-1][1]) == c["last_real"]
    with pytest.raises(ValueError):
        oracle_lib.segment_stream(np.zeros(10, np.float32), 4, 4)  # decode.rs:156-162


def test_segmenter_independent_of_packet_size(oracle_lib):
    x = np.random.default_rng(0).standard_normal(50000).astype(np.float32)
    ref = oracle_lib.segment_stream(x, 7000, 1300, packet=1152)
    for packet in (1, 333, 4096, 100000):
        got = oracle_lib.segment_stream(x, 7000, 1300, packet=packet)
        assert [s for _, s in got] == [s for _, s in ref]
        assert all(np.array_equal(a, b) for (a, _), (b, _) in zip(got, ref))


def test_source_sizing_cases(oracle_lib, cases):
    L = oracle_lib.lib()
    for c in cases["source_sizing"]:
        assert L.bo_source_samples(c["target"], c["src_rate"], c["dst_rate"]) == c["expect"]


def test_batching_trace(oracle_lib, cases, model_dir):
    path, _, m, labels = model_dir["mini"]
    om = oracle_lib.OracleModel(path)
    from birda_amd import synth
    for c in cases["batching"]:
        n = c["segments"] * m.sample_count
        x = synth.synth_segments(c["segments"], m.sample_count, m.sample_rate).reshape(-1) if n else np.zeros(0, np.float32)
        csv, st = om.process_stream(labels, x, m.sample_rate, batch_size=c["batch_size"])
        assert st.n_segments == c["segments"]
        assert st.effective_batch == c["effective"]
        assert st.n_batches == c["batches"] and st.n_padded_rows == c["padded_rows"]


# ---------------- resampler vs the reference's property tests (resample.rs:117-385) ----------------
def _sine(freq, rate, n):
    i = np.arange(n, dtype=np.float32)
    return np.sin(np.float32(2) * np.float32(np.pi) * np.float32(freq) * i / np.float32(rate)).astype(np.float32)


def _tone_power(s, rate, f):
    n = float(len(s))
    k = round(n * f / rate)
    w = 2 * np.pi * k / n
    c = 2 * np.cos(w)
    s1 = s2 = 0.0
    for x in s.astype(np.float64):
        s0 = c * s1 + x - s2
        s2, s1 = s1, s0
    return max(s1 * s1 + s2 * s2 - c * s1 * s2, 0.0) / n


def _steady(s, margin=8):
    m = len(s) // margin
    return s[m: len(s) - m]


def _rms(s):
    return float(np.sqrt(np.mean(s.astype(np.float64) ** 2)))


def test_resampler_block_sizes(oracle_lib, cases):
    for c in cases["resample"]["fft_chunks"]:
        assert oracle_lib.resampler_sizes(c["from"], c["to"]) == (c["fft_in"], c["fft_out"])
        assert c["fft_in"] == c["chunks"] * (c["from"] // np.gcd(c["from"], c["to"]))


def test_resampler_reference_properties(oracle_lib, cases):
    R = cases["resample"]
    K = R["constants"]
    ident = np.array(R["identity"]["samples"], np.float32)
    assert np.array_equal(oracle_lib.resample(ident, R["identity"]["rate"], R["identity"]["rate"]), ident)
    for c in R["length_bounds"]:
        x = np.sin(np.arange(c["n"], dtype=np.float32) * np.float32(0.001))
        n = len(oracle_lib.resample(x, c["from"], c["to"]))
        assert c["gt"] < n < c["lt"], c["src"]
    for c in R["tone_intact"]:
        body = _steady(oracle_lib.resample(_sine(c["tone"], c["from"], c["n"]), c["from"], c["to"]), K["steady_state_margin"])
        at = _tone_power(body, c["to"], c["tone"])
        assert at > len(body) / 4.0 * K["min_tone_power_fraction"], c["src"]
        for o in c["others"]:
            assert at > _tone_power(body, c["to"], o) * K["dominance_ratio"], c["src"]
        if "rms_floor" in c:
            assert _rms(body) > c["rms_floor"]
    for c in R["anti_alias"]:
        body = _steady(oracle_lib.resample(_sine(c["tone"], c["from"], c["n"]), c["from"], c["to"]), K["steady_state_margin"])
        if "alias" in c:
            assert _tone_power(body, c["to"], c["alias"]) < len(body) / 4.0 * c["alias_fraction"], c["src"]
        assert _rms(body) < c["rms_ceiling"], c["src"]
    a = R["amplitude"]
    x = _sine(a["tone"], a["from"], a["n"])
    assert abs(_rms(_steady(oracle_lib.resample(x, a["from"], a["to"]))) - _rms(x)) < a["tol"]


def test_process_stream_csv_is_sorted_and_thresholded(oracle_lib, model_dir):
    from birda_amd import synth
    path, _, m, labels = model_dir["mini"]
    om = oracle_lib.OracleModel(path)
    x = synth.synth_segments(7, m.sample_count, m.sample_rate).reshape(-1)
    csv, st, logits = om.process_stream(labels, x, m.sample_rate, min_conf=0.05, batch_size=4, want_logits=True)
    assert csv[:3] == b"\xef\xbb\xbf"
    lines = csv[3:].decode().strip().split("\n")
    assert lines[0] == "Start (s),End (s),Scientific name,Common name,Confidence,File"
    rows = [l.rsplit(",", 2) for l in lines[1:]]
    keys = [(float(l.split(",")[0]), -float(r[1])) for l, r in zip(lines[1:], rows)]
    assert keys == sorted(keys)
    assert all(float(r[1]) >= 0.05 for r in rows)
    assert st.n_segments == 7 and logits.shape == (7, m.n_classes)
    assert st.n_detections == len(rows) and st.n_detections <= 7 * 5


# ---------------- range filter: oracle against the reference's unit-test expectations ----------------
def test_scientific_name_matches_reference_tests(oracle_lib, cases):
    for c in cases["scientific_name"]:
        assert oracle_lib.scientific_name(c["label"]) == c["expect"], c["src"]


def _expected_scores(c):
    return np.asarray([np.nan if v is None else v for v in c["scores"]], np.float32)


def test_geomodel_projection_matches_reference_tests(oracle_lib, cases):
    for c in cases["geomodel_projection"]:
        got, mapped = oracle_lib.project_scores(c["geomodel"], [tuple(r) for r in c["reported"]], c["classifier"])
        want = _expected_scores(c)
        assert mapped == c["mapped"] and len(c["classifier"]) - mapped == c["unmatched"], c["src"]
        assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)]), c["src"]
        for thr, n in c.get("in_range", []):
            assert int((got >= np.float32(thr)).sum()) == n, c["src"]


def filter_case_tables(c):
    """Species of a filter_predictions case -> class indices, score table (NaN = no entry), inputs."""
    species = sorted(set(c["table"]) | {s for s, _ in c["in"]})
    index = {s: i for i, s in enumerate(species)}
    scores = np.asarray([c["table"].get(s, np.nan) for s in species] or [np.nan], np.float32)
    idx = np.asarray([index[s] for s, _ in c["in"]], np.int32)
    conf = np.asarray([p for _, p in c["in"]], np.float32)
    return species, index, scores, idx, conf


def test_filter_predictions_matches_reference_tests(oracle_lib, cases):
    for c in cases["filter_predictions"]:
        species, index, scores, idx, conf = filter_case_tables(c)
        oi, oc = oracle_lib.filter_predictions(idx, conf, scores, 0.01, c["policy"] == "keep", c["rerank"])
        assert [species[i] for i in oi] == [s for s, _ in c["out"]], c["src"]
        assert np.allclose(oc, [p for _, p in c["out"]], atol=1e-6), c["src"]     # the reference's own tolerance (:205)
        if not c["rerank"]:
            assert np.array_equal(oc, [np.float32(p) for _, p in c["out"]]), c["src"]   # confidence untouched: exact (:137-140)


def test_species_list_retain(oracle_lib):
    # classifier.rs:617-640: retain(|p| species_list.contains(&p.species)), order preserved
    oi, oc = oracle_lib.species_retain([4, 1, 3, 0], [0.9, 0.8, 0.7, 0.6], [1, 0, 0, 1, 1])
    assert oi.tolist() == [4, 3, 0] and np.array_equal(oc, np.asarray([0.9, 0.7, 0.6], np.float32))

"""HIP polyphase resampler (birda_amd/csrc/resample.hip) against (a) the oracle's restatement of
rubato's block-FFT resampler (reference src/audio/resample.rs:10-91) on the same inputs, (b) the
reference's OWN resampler tests (resample.rs:117-385, transcribed in reference_unit_cases.json),
and (c) through the pipeline: source-rate WAV -> device resample -> classifier -> CSV (config C5).

Tolerance: RESAMPLE_ATOL = 2e-5 absolute for |x| <= 1 inputs.  The device applies rubato's exact
block operator as a shift-invariant polyphase filter: the two differ by the operator's own
shift-variance (<= 4e-7 per tap, downsampling only), by dropped taps (< 1e-9 relative) and by f32
summation order over ~1 200 taps.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

RESAMPLE_ATOL = 2e-5
PAIRS = [(44100, 48000), (22050, 48000), (48000, 32000), (44100, 32000),
         # up-sampling by large factors and the odd rates of old recorders (polyphase form)
         (8000, 48000), (11025, 48000), (16000, 48000), (24000, 48000), (32000, 48000), (37800, 48000), (8000, 32000), (16000, 32000), (22050, 32000),
         # (round 6) decimation by more than 1.5 -- 88.2 / 96 kHz were taken by the polyphase form and came out 1e-4 from the oracle
         # (rubato's block is not shift-invariant there: resample.hip), 192 kHz and beyond were refused: the block form, a frame =
         # one rubato block, 32 or 16 frames a workgroup
         (96000, 48000), (88200, 48000), (96000, 32000), (192000, 48000), (256000, 48000), (300000, 48000), (192000, 32000), (384000, 48000),
         (250000, 32000), (384000, 32000), (500000, 48000)]


@pytest.fixture(scope="module", params=["f32", "f16x3"])
def clf(model_dir, request):
    """f32: the resampler GEMM on the f32 MFMA; f16x3: on the split-f16 MFMA (resample16_kernel).  Same tolerance."""
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = model_dir["mini"]
    c = BirdClassifier(path, labels, precision=request.param)
    yield c
    c.close()


@pytest.fixture(scope="module")
def cases():
    return json.load(open(os.path.join(GOLDEN, "reference_unit_cases.json")))


def _sine(freq, rate, n):
    return np.sin(2 * np.pi * freq * np.arange(n) / rate).astype(np.float32)


def _steady(s, margin=8):
    m = len(s) // margin
    return s[m: len(s) - m]


def _tone_power(s, rate, freq):
    """Goertzel power at `freq` (the reference's measure, resample.rs:190-211)."""
    n = len(s)
    k = int(0.5 + n * freq / rate)
    w = 2 * np.pi * k / n
    coeff = 2 * np.cos(w)
    s0 = s1 = s2 = 0.0
    for v in s.astype(np.float64):
        s0 = v + coeff * s1 - s2
        s2, s1 = s1, s0
    return (s1 * s1 + s2 * s2 - coeff * s1 * s2) / n


def _rms(s):
    return float(np.sqrt(np.mean(s.astype(np.float64) ** 2)))


@pytest.mark.parametrize("frm,to", PAIRS)
def test_matches_block_fft_restatement(clf, oracle_lib, frm, to):
    rng = np.random.default_rng(frm + to)
    fi, fo = oracle_lib.resampler_sizes(frm, to)
    full = int(np.ceil(144000 * frm / 48000))          # one 3 s raw segment at the source rate
    for n in (full, 5 * fi, 5 * fi + 1, 3 * fi - 7, fi // 3, 1):
        x = np.clip(0.3 * rng.standard_normal(n) + 0.5 * np.sin(2 * np.pi * 1234.5 * np.arange(n) / frm), -1, 1).astype(np.float32)
        want = oracle_lib.resample(x, frm, to)
        got = clf.resample(x, frm, to)
        assert got.shape == want.shape, (n, got.shape, want.shape)
        err = float(np.abs(got - want).max()) if len(want) else 0.0
        assert err <= RESAMPLE_ATOL, (frm, to, n, err)


@pytest.mark.parametrize("amplitude", [1e-4, 3e-7, 3.0e4, 1.0e9])
def test_quiet_and_loud_audio_keep_the_relative_tolerance(clf, oracle_lib, amplitude):
    """The resampler is linear, so its error must scale with the signal: field recordings with |x| ~ 1e-4 (or float WAVs far
    above 1) keep RESAMPLE_ATOL RELATIVE to their amplitude.  In the split-f16 kernel this is the block floating point of
    the staged span -- unscaled, samples of 1e-4 sit next to the f16 subnormals and the lo halves vanish."""
    frm, to = 44100, 48000
    rng = np.random.default_rng(11)
    n = 132300
    base = np.clip(0.3 * rng.standard_normal(n) + 0.5 * np.sin(2 * np.pi * 1234.5 * np.arange(n) / frm), -1, 1)
    base[: n // 3] *= 1e-3                     # a much quieter passage inside the same segment: local spans scale locally
    x = (base * amplitude).astype(np.float32)
    want = oracle_lib.resample(x, frm, to)
    got = clf.resample(x, frm, to)
    assert np.isfinite(got).all()
    err = float(np.abs(got - want).max())
    assert err <= RESAMPLE_ATOL * amplitude, (amplitude, err / amplitude)
    quiet = slice(2000, int(n // 3 * to / frm) - 4000)     # inside the quiet passage, away from its edges
    errq = float(np.abs(got[quiet] - want[quiet]).max())
    assert errq <= RESAMPLE_ATOL * amplitude * 1e-3 * 4, (amplitude, errq / (amplitude * 1e-3))


def test_identity_and_empty(clf):
    x = np.array([0.1, 0.2, 0.3, 0.4, 0.5], np.float32)
    assert np.array_equal(clf.resample(x, 48000, 48000), x)      # resample.rs:11-13, :354-359
    assert clf.resample(np.zeros(0, np.float32), 44100, 48000).size == 0


def test_reference_property_tests_on_the_device_resampler(clf, cases):
    """The reference's own tests (resample.rs:240-384), run against the HIP kernel."""
    R = cases["resample"]
    K = R["constants"]
    for c in R["length_bounds"]:
        x = np.sin(np.arange(c["n"], dtype=np.float32) * np.float32(0.001))
        assert c["gt"] < len(clf.resample(x, c["from"], c["to"])) < c["lt"], c["src"]
    for c in R["tone_intact"]:
        body = _steady(clf.resample(_sine(c["tone"], c["from"], c["n"]), c["from"], c["to"]), K["steady_state_margin"])
        at = _tone_power(body, c["to"], c["tone"])
        assert at > len(body) / 4.0 * K["min_tone_power_fraction"], c["src"]
        for o in c["others"]:
            assert at > _tone_power(body, c["to"], o) * K["dominance_ratio"], c["src"]
        if "rms_floor" in c:
            assert _rms(body) > c["rms_floor"]
    for c in R["anti_alias"]:
        body = _steady(clf.resample(_sine(c["tone"], c["from"], c["n"]), c["from"], c["to"]), K["steady_state_margin"])
        if "alias" in c:
            assert _tone_power(body, c["to"], c["alias"]) < len(body) / 4.0 * c["alias_fraction"], c["src"]
        assert _rms(body) < c["rms_ceiling"], c["src"]
    a = R["amplitude"]
    x = _sine(a["tone"], a["from"], a["n"])
    assert abs(_rms(_steady(clf.resample(x, a["from"], a["to"]))) - _rms(x)) < a["tol"]


def test_batched_device_resample_resizes_like_the_pipeline(clf, oracle_lib):
    """decode_and_stream: resample each raw segment, then resize(segment_samples, 0.0) (processor.rs:86-87)."""
    import torch
    frm, to, seg = 44100, 48000, 12000
    src = int(np.ceil(seg * frm / to))
    rng = np.random.default_rng(3)
    x = (0.5 * rng.standard_normal((5, src))).astype(np.float32)
    d_in = torch.from_numpy(x).cuda()
    ctx = clf.create_batch_context(5)
    for out_len in (seg, seg + 500):                     # truncating and zero-padding resize
        d_out = torch.full((5, out_len + 64), 7.0, device="cuda")
        clf.resample_device(ctx, d_in.data_ptr(), src, src, frm, to, d_out.data_ptr(), out_len + 64, out_len, 5)
        ctx.synchronize()
        got = d_out.cpu().numpy()
        assert (got[:, out_len:] == 7.0).all()           # nothing written past out_len
        for i in range(5):
            r = oracle_lib.resample(x[i], frm, to)
            want = np.zeros(out_len, np.float32)
            want[: min(out_len, len(r))] = r[:out_len]
            assert np.abs(got[i, :out_len] - want).max() <= RESAMPLE_ATOL
    ctx.close()


def test_c5_mixed_rate_files_through_the_pipeline(oracle_lib, model_dir, tmp_path):
    """Config C5 (per-file leg): 22.05 / 44.1 / 48 kHz WAVs -> device resample -> v2.4-shaped model -> CSV."""
    from birda_amd import pipeline, synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, names = model_dir["birdnet_v24_tiny"]
    om = oracle_lib.OracleModel(path)
    clf = BirdClassifier(path, labels, top_k=5, min_confidence=0.05)
    for rate in (22050, 44100, 48000):
        n = int(7.4 * rate)                              # 3 segments, the last one partial
        t = np.arange(n) / rate
        rng = np.random.default_rng(rate)
        x = np.clip(0.1 * rng.standard_normal(n) + 0.3 * np.sin(2 * np.pi * 1500 * t) + 0.3 * np.sin(2 * np.pi * 4200 * t), -1, 1)
        wav = str(tmp_path / f"rec_{rate}.wav")
        synth.write_wav_pcm16(wav, x, rate)
        res = pipeline.process_file(clf, wav, str(tmp_path), min_confidence=0.05, overlap=0.0, batch_size=4, front_end="host")
        pcm = np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)
        mono = np.zeros(pcm.size, np.float32)
        oracle_lib.lib().bo_pcm16_to_mono(pcm.ctypes.data, pcm.size, 1, mono)
        want, st = om.process_stream(names, mono, rate, 0.0, 0.05, 5, 4, True, wav)
        assert res.segments == st.n_segments == 3 and res.batches == st.n_batches
        g, w = open(res.output_path, "rb").read().decode().splitlines(), want.decode().splitlines()
        assert len(g) == len(w) and g[0] == w[0], (rate, len(g), len(w))
        for a, b in zip(g[1:], w[1:]):
            fa, fb = a.rsplit(",", 2), b.rsplit(",", 2)
            assert fa[0] == fb[0] and fa[2] == fb[2] and abs(float(fa[1]) - float(fb[1])) <= 2e-4, (rate, a, b)
        # device front end: PCM16 upload, scaling + windows + resampling on the GPU; same rows
        dev = tmp_path / f"dev_{rate}"
        dev.mkdir()
        rd = pipeline.process_file(clf, wav, str(dev), min_confidence=0.05, overlap=0.0, batch_size=4, front_end="device")
        gd = open(rd.output_path, "rb").read().decode().splitlines()
        assert rd.front_end == "device" and rd.segments == 3 and len(gd) == len(g)
        for a, b in zip(gd[1:], g[1:]):
            fa, fb = a.rsplit(",", 2), b.rsplit(",", 2)
            assert fa[0] == fb[0] and fa[2] == fb[2] and abs(float(fa[1]) - float(fb[1])) <= 1e-4, (rate, a, b)
    clf.close()


def test_pcm16_stream_is_segmented_scaled_and_mixed_on_the_device(oracle_lib, model_dir):
    """bh_predict_pcm16: one int16 upload, then append_samples + next_segment (+ resample) on the GPU.
    Against the oracle's host restatement of the same steps feeding the oracle's forward."""
    from birda_amd.classifier import BirdClassifier
    path, labels, m, names = model_dir["birdnet_v24_tiny"]
    om = oracle_lib.OracleModel(path)
    clf = BirdClassifier(path, labels, top_k=5, min_confidence=0.0)
    ctx = clf.create_batch_context(3)           # smaller than the segment count: several slices
    rng = np.random.default_rng(11)
    for rate, channels, overlap_s in ((48000, 1, 0.0), (48000, 2, 1.0), (44100, 1, 0.5), (22050, 2, 0.0)):
        n = int(8.3 * rate)
        t = np.arange(n) / rate
        x = 0.2 * rng.standard_normal((n, channels)) + 0.4 * np.sin(2 * np.pi * 2100 * t)[:, None]
        pcm = np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)
        ovl = int(np.float32(overlap_s) * np.float32(m.sample_rate))
        res, starts = clf.predict_pcm16(ctx, pcm if channels > 1 else pcm[:, 0], rate, ovl)
        # bh_predict_pcm_rows over several slices of a 3-segment context, with and without the device resampler: every segment
        # handed over once, in order, with the rows the plain call returns
        runs = []
        res_cb, starts_cb = clf.predict_pcm16(ctx, pcm if channels > 1 else pcm[:, 0], rate, ovl,
                                              on_rows=lambda first, rows, st: runs.append((first, rows, st)))
        assert starts_cb == starts and sum(len(r[1]) for r in runs) == len(res) and len(runs) >= (len(res) + 2) // 3
        nxt = 0
        for first, rows, st in runs:
            assert first == nxt and st == starts[first: first + len(rows)]
            for a, b in zip(rows, res[first: first + len(rows)]):
                assert [(p.index, p.confidence) for p in a.predictions] == [(p.index, p.confidence) for p in b.predictions]
            nxt += len(rows)
        # oracle: PCM scaling + mono mix, segmenter at the source rate, resample + resize, forward, top-k
        mono = np.zeros(n, np.float32)
        oracle_lib.lib().bo_pcm16_to_mono(pcm.ctypes.data, n, channels, mono)
        seg_src = int(oracle_lib.lib().bo_source_samples(m.sample_count, rate, m.sample_rate))
        ovl_src = int(oracle_lib.lib().bo_source_samples(ovl, rate, m.sample_rate))
        want = oracle_lib.segment_stream(mono, seg_src, ovl_src)
        assert starts == [s for _, s in want], (rate, channels, overlap_s)
        segs = np.zeros((len(want), m.sample_count), np.float32)
        for i, (raw, _) in enumerate(want):
            r = oracle_lib.resample(raw, rate, m.sample_rate)
            segs[i, : min(len(r), m.sample_count)] = r[: m.sample_count]
        ref = om.forward(segs)
        assert len(res) == len(want)
        for i, r in enumerate(res):
            conf = 1.0 / (1.0 + np.exp(-ref[i].astype(np.float64)))
            for p in r.predictions:                     # every reported class carries the oracle's confidence
                assert abs(p.confidence - conf[p.index]) <= 2e-4, (rate, channels, i, p.index)
            top = np.sort(conf)[::-1]
            assert abs(r.predictions[0].confidence - top[0]) <= 2e-4
    ctx.close(); clf.close()


def test_pcm16_pipeline_with_sub_slices_matches_the_f32_entry_point(model_dir):
    """A stream long enough for the pipelined upload (8-MiB pieces, four sub-slices, overlap between
    slices): bh_predict_pcm16 must report what bh_predict_batch reports for the same segments cut and
    scaled on the host."""
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = model_dir["mini"]
    clf = BirdClassifier(path, labels, top_k=5, min_confidence=0.0)
    S = m.sample_count
    ovl = S // 4
    nseg = 1100                                     # two slices of a 600-segment context, second one ragged
    n = (S - ovl) * (nseg - 1) + S - 37             # last segment short: zero-padded tail
    rng = np.random.default_rng(5)
    t = np.arange(n) / m.sample_rate
    x = 0.2 * rng.standard_normal((n, 2)) + 0.4 * np.sin(2 * np.pi * (900 + 0.05 * np.arange(n) % 2000) * t)[:, None]
    pcm = np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)
    ctx = clf.create_batch_context(600)
    res, starts = clf.predict_pcm16(ctx, pcm, m.sample_rate, ovl)
    assert starts == clf.segment_starts(n, S, ovl) and len(res) == len(starts) >= nseg   # (the segmenter itself: test_abi_and_host)
    nseg = len(starts)
    mono = ((pcm[:, 0].astype(np.float32) / np.float32(32768.0)) + (pcm[:, 1].astype(np.float32) / np.float32(32768.0))) / np.float32(2.0)
    segs = np.zeros((nseg, S), np.float32)
    for i, s in enumerate(starts):
        piece = mono[s: s + S]
        segs[i, : len(piece)] = piece
    want = clf.predict_batch_with_context(ctx, [segs[i] for i in range(600)]) + \
        clf.predict_batch_with_context(ctx, [segs[i] for i in range(600, nseg)])
    for i, (a, b) in enumerate(zip(res, want)):
        assert [p.index for p in a.predictions] == [p.index for p in b.predictions], i
        assert np.allclose([p.confidence for p in a.predictions], [p.confidence for p in b.predictions], atol=2e-6), i
    # bh_predict_pcm_rows: the same rows, handed over run by run as the sub-slices finish -- every segment exactly once, in
    # order, with its start sample, and in more than one run per slice (the sub-slices of a 600-segment slice)
    runs = []
    res2, starts2 = clf.predict_pcm16(ctx, pcm, m.sample_rate, ovl, on_rows=lambda first, rows, st: runs.append((first, rows, st)))
    assert starts2 == starts and len(runs) > 2
    nxt = 0
    for first, rows, st in runs:
        assert first == nxt and len(rows) == len(st) > 0
        assert st == starts[first: first + len(rows)]
        for a, b in zip(rows, res[first: first + len(rows)]):
            assert [p.index for p in a.predictions] == [p.index for p in b.predictions]
            assert [p.confidence for p in a.predictions] == [p.confidence for p in b.predictions]
        nxt += len(rows)
    assert nxt == nseg
    for a, b in zip(res2, res):
        assert [(p.index, p.confidence) for p in a.predictions] == [(p.index, p.confidence) for p in b.predictions]
    # bh_batch_context_set_sub_slices: however a slice is split -- whole, halves, five parts, back to automatic -- the rows are the
    # same bits (every kernel of the forward is independent of the batch it runs in)
    for n_sub in (1, 2, 5, 0):
        ctx.set_sub_slices(n_sub)
        res3, starts3 = clf.predict_pcm16(ctx, pcm, m.sample_rate, ovl)
        assert starts3 == starts
        for a, b in zip(res3, res):
            assert [(p.index, p.confidence) for p in a.predictions] == [(p.index, p.confidence) for p in b.predictions], n_sub
    ctx.close(); clf.close()


def test_device_resampler_rows_are_independent_and_repeatable(clf):
    """Identical raw segments anywhere in a batch resample to bit-identical rows, run after run (the split-f16 kernel keeps
    MFMAs in flight around its operand split: the class of hazard DESIGN.md section 3 lists shows up here first)."""
    import torch
    frm, to, seg = 44100, 48000, 144000
    src = int(np.ceil(seg * frm / to))
    rng = np.random.default_rng(9)
    uniq = (0.4 * rng.standard_normal((4, src))).astype(np.float32)
    order = np.arange(96) % 4
    d_in = torch.from_numpy(uniq[order]).cuda()
    ctx = clf.create_batch_context(96)
    first = None
    for _ in range(6):
        d_out = torch.empty((96, seg), device="cuda")
        clf.resample_device(ctx, d_in.data_ptr(), src, src, frm, to, d_out.data_ptr(), seg, seg, 96)
        ctx.synchronize()
        got = d_out.cpu().numpy()
        for k in range(4):
            rows = got[order == k]
            assert (rows == rows[0]).all()
        if first is None:
            first = got
        assert np.array_equal(first, got)
    ctx.close()

"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs, against the committed golden vectors, and -- at full BASELINE sizes --
through size-independent properties.

Tolerances (fp32 path).  Two fp32 implementations of this model cannot agree bit for bit:
  * the spectrogram ends in |v|^0.45 (square then ^(1/(1+e^1.23))), which is not Lipschitz at
    v = 0: an fp32 rounding d in the mel projection moves a pixel by up to ~|d|^0.45 (~1e-3
    for d ~ 1e-7).  Such pixels are rare (mean |d| stays ~1e-7), so spectrograms are compared
    with a loose max (2e-3) and a tight mean (1e-5);
  * every later layer is Lipschitz; differences come from summation order (oracle: plain
    mul+add in k order; MFMA: fmaf chain in a permuted k order) and from libm vs ocml erff.
LOGIT_RTOL is the stated fp32 logit tolerance: max |dlogit| <= 2e-5 * max(1, max|logit|)
... measured 1.3e-6 relative on the full model (profiles/README.md).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

LOGIT_RTOL = 2e-5
F16_LOGIT_RTOL = 3e-3      # plain f16 MFMA operands (BH_FLAG_F16, BASELINE config 5): measured 9e-4 of the logit scale on the full model
DEGENERATE_RTOL = 1e-2
SPEC_MAX_ATOL = 2e-3
SPEC_MEAN_ATOL = 1e-5


def _logit_close(got, ref):
    scale = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(got - ref).max())
    assert np.isfinite(got).all()
    assert err <= LOGIT_RTOL * scale, f"max|dlogit| {err:.3e} > {LOGIT_RTOL * scale:.3e}"
    return err


@pytest.fixture(scope="module")
def clf_mini(model_dir):
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = model_dir["mini"]
    c = BirdClassifier(path, labels, top_k=5, min_confidence=0.1)
    yield c
    c.close()


@pytest.fixture(scope="module")
def clf_tiny(model_dir):
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = model_dir["birdnet_v24_tiny"]
    c = BirdClassifier(path, labels, top_k=5, min_confidence=0.1)
    yield c
    c.close()


def test_native_library_is_the_one_loaded():
    from birda_amd import _lib
    L = _lib.load()
    assert L.bh_device_count() >= 1
    maps = open("/proc/self/maps").read()
    assert "birda_amd/libbirda_hip.so" in maps


def test_mini_logits_match_golden_vectors_and_oracle(clf_mini, model_dir, oracle_lib):
    from birda_amd import synth
    path, _, m, _ = model_dir["mini"]
    vec = np.load(os.path.join(GOLDEN, "model_vectors.npz"))
    segs = synth.synth_segments(4, m.sample_count, m.sample_rate)
    ctx = clf_mini.create_batch_context(4)
    logits, emb = clf_mini.predict_logits(ctx, segs, want_embeddings=True)
    assert np.abs(logits - vec["mini_logits"]).max() < 1e-3       # float64 torch/numpy vectors
    assert np.abs(emb - vec["mini_embedding"]).max() < 1e-3
    om = oracle_lib.OracleModel(path)
    ref, ref_emb = om.forward(segs, want_embeddings=True)
    _logit_close(logits, ref)
    _logit_close(emb, ref_emb)
    ctx.close()


def test_every_tensor_matches_oracle_layer_by_layer(model_dir, oracle_lib, monkeypatch):
    from birda_amd import synth
    from birda_amd.classifier import BirdClassifier
    monkeypatch.setenv("BIRDA_HIP_KEEP_TENSORS", "1")
    for kind, n in (("mini", 3), ("birdnet_v24_tiny", 2)):
        path, _, m, _ = model_dir[kind]
        segs = synth.synth_segments(n, m.sample_count, m.sample_rate, start=11)
        clf = BirdClassifier(path)
        ctx = clf.create_batch_context(n)
        clf.predict_logits(ctx, segs)
        om = oracle_lib.OracleModel(path)
        for t in range(len(m.layers) + 1):
            ref = om.forward(segs, dump_tensor=t)[1]
            got = clf.read_tensor(ctx, t, n)
            d = np.abs(got - ref)
            scale = max(1.0, float(np.abs(ref).max()))
            assert np.isfinite(got).all(), (kind, t)
            # the spectrogram's rare |d|^0.45 pixels propagate through the first layers
            assert d.max() <= SPEC_MAX_ATOL * scale, (kind, t, d.max())
            assert d.mean() <= SPEC_MEAN_ATOL * scale, (kind, t, d.mean())
        ctx.close()
        clf.close()


def test_v24_frontend_matches_golden_spectrogram(clf_tiny, model_dir, monkeypatch):
    from birda_amd import synth
    from birda_amd.classifier import BirdClassifier
    monkeypatch.setenv("BIRDA_HIP_KEEP_TENSORS", "1")
    path, _, m, _ = model_dir["birdnet_v24_tiny"]
    vec = np.load(os.path.join(GOLDEN, "model_vectors.npz"))
    clf = BirdClassifier(path)
    ctx = clf.create_batch_context(1)
    logits = clf.predict_logits(ctx, synth.synth_segment(3)[None])
    spec = clf.read_tensor(ctx, 0, 1).reshape(2, 96, 511)[:, :, ::7]
    ref = vec["tiny_spec_seg3_frames_every7"]
    assert np.abs(spec - ref).max() < SPEC_MAX_ATOL and np.abs(spec - ref).mean() < SPEC_MEAN_ATOL
    assert np.abs(logits[0] - vec["tiny_logits_seg3"]).max() < 1e-3
    ctx.close()
    clf.close()


def test_full_model_logits_and_topk_match_oracle(full_model, oracle_lib):
    """BirdNET-v2.4-shaped model (51 layers, 6 522 classes), 6 segments incl. edge inputs."""
    from birda_amd import synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, names = full_model
    segs = synth.synth_segments(6, m.sample_count, m.sample_rate, start=100)
    segs[3] = 0.0                                   # all-zero segment = warm-up / padding row
    segs[4] = np.float32(0.25)                      # constant: max - min = 0 -> eps path
    segs[5, ::2] = 1.0; segs[5, 1::2] = -1.0        # full-scale Nyquist square wave
    clf = BirdClassifier(path, labels, top_k=5, min_confidence=0.01)
    ctx = clf.create_batch_context(8)
    logits = clf.predict_logits(ctx, segs)
    om = oracle_lib.OracleModel(path)
    ref = om.forward(segs)
    err = _logit_close(logits[:3], ref[:3])
    print(f"full model max|dlogit| = {err:.3e} on max|logit| {np.abs(ref).max():.2f}")
    # Rows 3-5 are degenerate: after min/max normalisation they are constant (or pure Nyquist),
    # so every mel projection is exactly 0 in exact arithmetic and |fp32 rounding noise|^0.45
    # IS the spectrogram (~1e-3).  No two fp32 implementations agree there, the reference's
    # included; such rows (warm-up and padding rows are all-zero segments) must stay finite
    # and within DEGENERATE_RTOL, and their results are discarded by the pipeline anyway
    # (processor.rs:363-367).
    scale = max(1.0, float(np.abs(ref).max()))
    derr = float(np.abs(logits[3:] - ref[3:]).max())
    print(f"degenerate rows max|dlogit| = {derr:.3e}")
    assert np.isfinite(logits).all() and derr <= DEGENERATE_RTOL * scale
    res = clf.predict_batch_with_context(ctx, list(segs))
    for i, r in enumerate(res[:3]):
        idx, conf = oracle_lib.topk(ref[i], 1, 5, 0.01)
        got_idx = [p.index for p in r.predictions]
        # ties / near-ties may swap under fp32 noise: compare as sets unless well separated
        top = np.sort(ref[i])[::-1][:6]
        if np.min(np.abs(np.diff(top))) > 1e-2:
            assert got_idx == idx.tolist()
        assert np.allclose([p.confidence for p in r.predictions], 1 / (1 + np.exp(-logits[i][got_idx])), atol=1e-6)
        assert all(p.species == names[p.index] for p in r.predictions)
    ctx.close()
    clf.close()


def test_predict_entry_points_agree_and_preserve_order(clf_tiny, model_dir):
    from birda_amd import synth
    _, _, m, _ = model_dir["birdnet_v24_tiny"]
    segs = synth.synth_segments(5, m.sample_count, m.sample_rate, start=40)
    ctx = clf_tiny.create_batch_context(3)          # smaller than n: logits path slices internally
    logits = clf_tiny.predict_logits(ctx, segs)
    one = [clf_tiny.predict(s) for s in segs]
    batch = clf_tiny.predict_batch([s for s in segs])
    ctx5 = clf_tiny.create_batch_context(5)
    with_ctx = clf_tiny.predict_batch_with_context(ctx5, [s for s in segs])
    for a, b, c in zip(one, batch, with_ctx):
        assert [p.index for p in a.predictions] == [p.index for p in b.predictions] == [p.index for p in c.predictions]
        assert np.allclose([p.confidence for p in a.predictions], [p.confidence for p in b.predictions], atol=1e-6)
    # order preservation + batch independence: reversing the batch reverses the rows bit for bit
    rev = clf_tiny.predict_logits(ctx5, segs[::-1])
    assert np.array_equal(rev[::-1], clf_tiny.predict_logits(ctx5, segs))
    # padding rows (zeros) do not change real rows (process_batch pads, processor.rs:240-258)
    padded = np.concatenate([segs[:2], np.zeros((3, m.sample_count), np.float32)])
    assert np.array_equal(clf_tiny.predict_logits(ctx5, padded)[:2], clf_tiny.predict_logits(ctx5, segs[:2]))
    assert ctx.input_buffer_bytes() == 3 * m.sample_count * 4
    ctx.close(); ctx5.close()


def test_pinned_and_registered_host_segments_give_the_same_results(clf_tiny, model_dir):
    """bh_predict_batch_contig uploads straight from pinned memory (bh_host_alloc, or a caller's buffer under
    bh_host_register) and gathers pageable memory through its staging first: the results are identical, also across slices
    (700 segments through a 256-segment context) and after the buffer is unregistered again."""
    from birda_amd import synth
    from birda_amd._lib import BhResult, check
    from birda_amd.classifier import PinnedSegments
    _, _, m, _ = model_dir["birdnet_v24_tiny"]
    n = 700
    uniq = synth.synth_segments(16, m.sample_count, m.sample_rate, start=3)
    host = np.ascontiguousarray(np.tile(uniq, (n // 16 + 1, 1))[:n])
    ctx = clf_tiny.create_batch_context(256)
    L = clf_tiny._L

    def run(ptr):
        arr = (BhResult * n)()
        check(L.bh_predict_batch_contig(clf_tiny._h, ctx._h, ptr, n, arr))
        return [(r.n_pred, list(r.index[: r.n_pred]), [float(c) for c in r.confidence[: r.n_pred]]) for r in arr]

    want = run(host.ctypes.data)
    assert any(w[0] > 0 for w in want)
    pin = PinnedSegments(n, m.sample_count)
    pin.array[:] = host
    assert run(pin.array.ctypes.data) == want
    pin.close()
    other = host.copy()
    check(L.bh_host_register(other.ctypes.data, other.nbytes))
    assert run(other.ctypes.data) == want
    check(L.bh_host_unregister(other.ctypes.data))
    assert run(other.ctypes.data) == want
    assert L.bh_host_alloc(0, None) != 0 and L.bh_host_register(None, 16) != 0
    # the PCM16 stream entry point takes the same short cut
    pcm = np.clip(np.round(host[:300].reshape(-1).astype(np.float64) * 32767.0), -32768, 32767).astype(np.int16)
    want16 = clf_tiny.predict_pcm16(ctx, pcm, m.sample_rate, 0)
    check(L.bh_host_register(pcm.ctypes.data, pcm.nbytes))
    got16 = clf_tiny.predict_pcm16(ctx, pcm, m.sample_rate, 0)
    check(L.bh_host_unregister(pcm.ctypes.data))
    assert got16[1] == want16[1] and [[(p.index, p.confidence) for p in r.predictions] for r in got16[0]] == \
        [[(p.index, p.confidence) for p in r.predictions] for r in want16[0]]
    ctx.close()


def test_host_batch_pipeline_matches_the_device_path(clf_mini, model_dir):
    """The host entry points gather 32-segment chunks on worker threads, copy each chunk on a second
    stream and compute the slice in sub-slices (n >= 512): rows must come back in order and bit-identical
    to one device-resident forward of the same batch, for scattered slices as well as a contiguous block."""
    import torch
    _, _, m, _ = model_dir["mini"]
    from birda_amd import synth
    n = 700                                            # 4 sub-slices of 192 (last 124), 22 chunks
    uniq = synth.synth_segments(50, m.sample_count, m.sample_rate, start=900)
    segs = uniq[np.arange(n) % 50]
    ctx = clf_mini.create_batch_context(n)
    x = torch.from_numpy(segs).cuda()
    dev = torch.empty((n, m.n_classes), device="cuda")
    clf_mini.forward_device(ctx, x.data_ptr(), n, dev.data_ptr()); ctx.synchronize()
    want = dev.cpu().numpy()
    assert np.array_equal(clf_mini.predict_logits(ctx, segs), want)
    res = clf_mini.predict_batch_with_context(ctx, [segs[i] for i in range(n)])      # scattered host slices
    top1 = want.argmax(1)
    for i, r in enumerate(res):
        if r.predictions:
            assert r.predictions[0].index == top1[i]
    small = clf_mini.create_batch_context(96)          # slices of 96 < n: several slices, one sub-slice each
    assert np.array_equal(clf_mini.predict_logits(small, segs), want)
    ctx.close(); small.close()


def test_error_behaviour(clf_tiny, model_dir):
    from birda_amd._lib import BirdaHipError
    _, _, m, _ = model_dir["birdnet_v24_tiny"]
    with pytest.raises(BirdaHipError) as e:          # wrong segment length is rejected, not padded
        clf_tiny.predict(np.zeros(m.sample_count - 1, np.float32))
    assert e.value.code == -1 and "samples" in str(e.value)
    ctx = clf_tiny.create_batch_context(2)
    with pytest.raises(BirdaHipError) as e:          # batch larger than the context
        clf_tiny.predict_batch_with_context(ctx, [np.zeros(m.sample_count, np.float32)] * 3)
    assert e.value.code == -1
    assert clf_tiny.predict_batch([]) == []
    ctx.close()


def test_warmup_registry(model_dir):
    """ensure_warm: per-size independence and idempotence (reference classifier.rs:1114-1173)."""
    from birda_amd.classifier import BirdClassifier
    clf = BirdClassifier(model_dir["mini"][0])
    assert not clf.is_warm(4) and not clf.is_warm(2)
    clf.ensure_warm(4)
    assert clf.is_warm(4) and not clf.is_warm(2)
    clf.ensure_warm(4)
    clf.ensure_warm(2)
    assert clf.is_warm(2) and clf.is_warm(4) and not clf.is_warm(1)
    clf.close()


def test_c1_wav_to_csv_matches_oracle(full_model, oracle_lib, tmp_path):
    """Config C1: 30 s / 48 kHz mono PCM16 WAV -> CSV, defaults (batch 8, overlap 0, BOM)."""
    from birda_amd import pipeline, synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, names = full_model
    x = synth.synth_segments(10, m.sample_count, m.sample_rate).reshape(-1)
    wav = str(tmp_path / "rec30s.wav")
    synth.write_wav_pcm16(wav, x, m.sample_rate)
    clf = BirdClassifier(path, labels, top_k=5, min_confidence=0.1)
    # the reference's structure (decode thread, batches of 8 with zero-padded rows) ...
    res = pipeline.process_file(clf, wav, str(tmp_path), min_confidence=0.1, overlap=0.0, batch_size=8, front_end="host")
    assert res.segments == 10 and res.effective_batch == 8 and res.batches == 2 and res.padded_rows == 6 and res.front_end == "host"
    assert res.output_path.endswith("rec30s.BirdNET.results.csv")
    got = open(res.output_path, "rb").read()
    # ... and the device front end at the backend's default batch size (256, capped to the 10 estimated segments): the
    # PCM16 frames are scaled and windowed on the GPU, nothing is padded, the CSV is byte-identical
    dev_dir = tmp_path / "dev"
    dev_dir.mkdir()
    rd = pipeline.process_file(clf, wav, str(dev_dir), min_confidence=0.1, overlap=0.0)
    assert rd.front_end == "device" and rd.segments == 10 and rd.effective_batch == 10 and rd.batches == 1 and rd.padded_rows == 0
    assert open(rd.output_path, "rb").read() == got
    # oracle on the PCM16-quantised samples, exactly what the decoder produces
    pcm = np.clip(np.round(x.astype(np.float64) * 32767.0), -32768, 32767).astype(np.int16)
    mono = np.zeros(pcm.size, np.float32)
    oracle_lib.lib().bo_pcm16_to_mono(pcm.ctypes.data, pcm.size, 1, mono)
    om = oracle_lib.OracleModel(path)
    want, st, ref_logits = om.process_stream(names, mono, m.sample_rate, 0.0, 0.1, 5, 8, True, wav, want_logits=True)
    assert st.n_segments == 10 and st.n_batches == 2 and st.n_padded_rows == 6
    assert got[:3] == b"\xef\xbb\xbf"
    if got != want:
        # CSV parity modulo fp32 noise: same rows, confidences equal to 4 decimals +- 1e-4
        g, w = got.decode().splitlines(), want.decode().splitlines()
        assert len(g) == len(w) and g[0] == w[0]
        for a, b in zip(g[1:], w[1:]):
            fa, fb = a.rsplit(",", 2), b.rsplit(",", 2)
            assert fa[0] == fb[0] and fa[2] == fb[2] and abs(float(fa[1]) - float(fb[1])) <= 1.01e-4
    assert res.detections == len(got.decode().splitlines()) - 1
    clf.close()


@pytest.mark.parametrize("se_div", [1, 2, 6])
def test_wide_squeeze_excite_gates_match_oracle_at_every_hidden_width(se_div, tmp_path, oracle_lib):
    """The two-launch gate of the blocks beyond 576 expanded channels (kernels_conv.hip se_hidden_kernel + se_gate16_kernel, round 6)
    on gates this repo's own plans do not have: hidden layers of 16-160 units (EfficientNet's own ratio is 1 / 24 of the expanded
    width; here 1 / 6, 1 / 12, 1 / 36), i.e. one, two, ... sixteen threads a hidden unit and channel slices of 256 or 128, expanded
    widths that are no multiple of 256 (600, 960), one gate of exactly 576 channels on the one-launch kernel beside them, and
    launches of 1, 17 and 50 segments (a last group of sixteen that is partly empty).  Held to the oracle at the fp32 logit
    tolerance; a segment's logits must not depend on the launch.  (reference: any published .onnx goes to the classifier,
    src/inference/classifier.rs:269-283.)"""
    from birda_amd import modelfile as mf, synth
    from birda_amd.classifier import BirdClassifier
    plan = dict(sr=48000, n=144000, branches=[(1024, 560, 40, 0.0, 12000.0)], stem=32, act=mf.ACT_SWISH, se=True, se_div=se_div, head=256,
                classes=120, stages=[(1, 3, 1, 24, 1), (6, 3, 2, 96, 2), (6, 5, 2, 100, 2), (6, 3, 2, 160, 2), (4, 3, 1, 232, 1)])
    m = synth.build_model("custom", plan=plan)
    bhm = str(tmp_path / "wide_gates.bhm")
    mf.write_model(bhm, m)
    n_blocks = sum(1 for L in m.layers if L.op == mf.OP_DWCONV)
    segs = synth.synth_segments(50, m.sample_count, m.sample_rate, start=21)
    ref = oracle_lib.OracleModel(bhm).forward(segs[:5])
    scale = max(1.0, float(np.abs(ref).max()))
    for prec in ("f16x3", "f32"):
        clf = BirdClassifier(bhm, None, precision=prec)
        # (f32 MFMA: the planner leaves a wide block whose best entry pads its MFMA work 2.5-fold to the layer kernels -- their gate is the
        #  pool + GEMM form; tests/test_random_plans_gpu.py)
        assert len(clf.fused_blocks()) >= n_blocks - (3 if prec == "f32" else 0), (prec, clf.fused_blocks(), n_blocks)
        outs = {}
        for n in (1, 17, 50):
            ctx = clf.create_batch_context(n)
            outs[n] = clf.predict_logits(ctx, segs[:n])
            ctx.close()
        clf.close()
        assert np.isfinite(outs[50]).all()
        err = float(np.abs(outs[50][:5] - ref).max())
        assert err <= LOGIT_RTOL * scale, (prec, se_div, err, scale)
        assert (outs[17] == outs[50][:17]).all() and (outs[1] == outs[50][:1]).all(), (prec, se_div)


def test_low_latency_flag_splits_late_blocks_and_keeps_its_own_bits(full_model, oracle_lib):
    """BH_FLAG_LOW_LATENCY (round 6): forwards of at most 32 segments run the late blocks 2 / 4 / 8 workgroups
    deep over their expanded channels and add the partial project sums in index order (0.70 -> 0.38 ms for one segment of the
    BirdNET-shaped model).  The depth belongs to the block, never to the launch: within the regime (1, 8, 20, 32 segments) a
    segment's logits are bit-identical whatever launch it ran in; beyond it (64) the flag changes nothing; against the oracle the
    regime holds the fp32 tolerance like every other path; and WITHOUT the flag every launch size gives the same bits, as before."""
    from birda_amd import synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = full_model
    segs = synth.synth_segments(64, m.sample_count, m.sample_rate, start=7)
    ref = oracle_lib.OracleModel(path).forward(segs[:8])
    scale = max(1.0, float(np.abs(ref).max()))
    out = {}
    for ll in (False, True):
        clf = BirdClassifier(path, labels, low_latency=ll)
        for n in (1, 8, 20, 32, 64):
            ctx = clf.create_batch_context(n)
            out[(ll, n)] = clf.predict_logits(ctx, segs[:n])
            ctx.close()
        clf.close()
    for n in (1, 8, 20, 32, 64):                               # plain: one set of bits for every launch size
        assert (out[(False, n)] == out[(False, 64)][:n]).all(), n
    for n in (1, 8, 20):                                       # the regime: its own bits, the same for every launch size inside it
        assert (out[(True, n)] == out[(True, 32)][:n]).all(), n
    assert (out[(True, 64)] == out[(False, 64)]).all()         # beyond 32 segments the flag does nothing
    d = float(np.abs(out[(True, 32)] - out[(False, 32)]).max())
    assert 0.0 < d <= 2e-6 * scale, d                          # another summation order, ~1e-7 of the logit scale
    assert np.abs(out[(True, 8)] - ref).max() <= LOGIT_RTOL * scale and np.abs(out[(False, 8)] - ref).max() <= LOGIT_RTOL * scale


def test_file_descriptor_route_gives_the_mapped_routes_rows(clf_tiny, model_dir, tmp_path, monkeypatch):
    """bh_predict_pcm_fd_rows (round 6): the WAV's data chunk read by `pread` straight into the pinned staging
    buffer gives, row for row and bit for bit, what bh_predict_pcm_rows gives on the same bytes in memory; a stream that ends
    before its frames do is BH_ERR_IO; and bhh_process_file writes the same CSV by either route (BIRDA_HOST_PREAD=1: by descriptor;
    the default stays the mapped route -- the descriptor route measured no faster, profiles/r6_k_pread.txt)."""
    import ctypes as C
    from birda_amd import _lib, pipeline, synth
    path, labels, m, names = model_dir["birdnet_v24_tiny"]
    L = _lib.load()
    x = synth.synth_segments(40, m.sample_count, m.sample_rate, start=2).reshape(-1)[: int(118.3 * m.sample_rate)]
    wav = str(tmp_path / "long.wav")
    synth.write_wav_pcm16(wav, x, m.sample_rate)
    raw = open(wav, "rb").read()
    off = raw.index(b"data") + 8
    pcm = np.frombuffer(raw[off:], np.int16).copy()
    n_frames = pcm.size
    ovl = m.sample_count // 3
    ctx = clf_tiny.create_batch_context(16)          # several slices: the descriptor route walks the file slice by slice
    cap = 128
    def run(fn, *head):
        out = (_lib.BhResult * cap)()
        nseg = C.c_size_t()
        starts = (C.c_uint64 * cap)()
        rc = fn(clf_tiny._h, ctx._h, *head, 1, n_frames, 1, m.sample_rate, ovl, out, cap, C.byref(nseg), starts, _lib.BhRowsFn(), None)
        return rc, [(int(starts[i]), out[i].n_pred, list(out[i].index[:out[i].n_pred]), list(out[i].confidence[:out[i].n_pred])) for i in range(nseg.value)]
    rc_p, rows_p = run(L.bh_predict_pcm_rows, pcm.ctypes.data_as(C.c_void_p))
    fd = os.open(wav, os.O_RDONLY)
    try:
        rc_f, rows_f = run(L.bh_predict_pcm_fd_rows, fd, off)
        assert rc_p == 0 and rc_f == 0, L.bh_last_error()
        assert len(rows_f) == len(rows_p) >= 55 and rows_f == rows_p
        assert os.lseek(fd, 0, os.SEEK_CUR) == 0                       # the descriptor's own offset is not moved
        # frames promised beyond the end of the file
        out = (_lib.BhResult * 4096)()
        nseg = C.c_size_t()
        rc = L.bh_predict_pcm_fd_rows(clf_tiny._h, ctx._h, fd, off, 1, n_frames + 40 * m.sample_count, 1, m.sample_rate, 0, out, 4096, C.byref(nseg), None, _lib.BhRowsFn(), None)
        assert rc == -2, (rc, L.bh_last_error())      # BH_ERR_IO
    finally:
        os.close(fd)
    ctx.close()
    a, b = tmp_path / "mapped", tmp_path / "by_fd"
    a.mkdir(); b.mkdir()
    ra = pipeline.process_file(clf_tiny, wav, str(a), min_confidence=0.05, overlap=1.0)
    assert ra.front_end == "device"
    import subprocess, sys, textwrap
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
        from birda_amd import pipeline
        from birda_amd.classifier import BirdClassifier
        clf = BirdClassifier({path!r}, {labels!r}, top_k=5, min_confidence=0.1)
        r = pipeline.process_file(clf, {wav!r}, {str(b)!r}, min_confidence=0.05, overlap=1.0)
        assert r.front_end == "device"
    """)
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, BIRDA_HOST_PREAD="1"), timeout=300)
    assert open(ra.output_path, "rb").read() == open(os.path.join(str(b), os.path.basename(ra.output_path)), "rb").read()


def test_overlap_and_short_file_batching(clf_tiny, model_dir, oracle_lib, tmp_path):
    from birda_amd import pipeline, synth
    path, _, m, names = model_dir["birdnet_v24_tiny"]
    om = oracle_lib.OracleModel(path)
    x = synth.synth_segments(4, m.sample_count, m.sample_rate, start=7).reshape(-1)[: int(10 * m.sample_rate)]
    wav = str(tmp_path / "ten.wav")
    synth.write_wav_pcm16(wav, x, m.sample_rate)
    pcm = np.clip(np.round(x.astype(np.float64) * 32767.0), -32768, 32767).astype(np.int16)
    mono = np.zeros(pcm.size, np.float32)
    oracle_lib.lib().bo_pcm16_to_mono(pcm.ctypes.data, pcm.size, 1, mono)
    for overlap, batch, n_seg in ((0.0, 16, 4), (1.0, 4, 6), (1.5, 1, 7)):
        res = pipeline.process_file(clf_tiny, wav, str(tmp_path), min_confidence=0.05, overlap=overlap, batch_size=batch, front_end="host")
        want, st = om.process_stream(names, mono, m.sample_rate, overlap, 0.05, 5, batch, True, wav)
        assert res.segments == st.n_segments == n_seg
        assert res.effective_batch == st.effective_batch and res.batches == st.n_batches
        assert res.padded_rows == st.n_padded_rows
        host_csv = open(res.output_path, "rb").read()
        g, w = host_csv.decode().splitlines(), want.decode().splitlines()
        assert len(g) == len(w)
        for a, b in zip(g[1:], w[1:]):
            fa, fb = a.rsplit(",", 2), b.rsplit(",", 2)
            assert fa[0] == fb[0] and abs(float(fa[1]) - float(fb[1])) <= 1.01e-4
        # the device front end (same windows, the trailing overlap remainder included), spans cut at a batch of `batch`
        rd = pipeline.process_file(clf_tiny, wav, str(tmp_path), min_confidence=0.05, overlap=overlap, batch_size=batch, front_end="device")
        assert rd.front_end == "device" and rd.segments == n_seg and rd.padded_rows == 0
        assert open(rd.output_path, "rb").read() == host_csv


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_full_size_batch_properties(full_model, precision):
    """BASELINE configs[1] size (1 000 segments, device resident): properties that need no
    oracle -- determinism, row independence across micro-batch boundaries, checksum of rows.
    (Identical segments anywhere in the batch must give bit-identical rows: this is the test that
    catches instruction-hazard / race bugs that a 4-segment oracle comparison does not.)"""
    import torch
    from birda_amd import synth
    from birda_amd.classifier import BirdClassifier
    path, _, m, _ = full_model
    clf = BirdClassifier(path, precision=precision)
    uniq = synth.synth_segments(8, m.sample_count, m.sample_rate)
    order = np.arange(1000) % 8
    x = torch.from_numpy(uniq[order]).cuda()
    logits = torch.empty((1000, m.n_classes), device="cuda")
    idx = torch.empty((1000, 5), dtype=torch.int32, device="cuda")
    conf = torch.empty((1000, 5), device="cuda")
    ctx = clf.create_batch_context(256)
    clf.forward_device(ctx, x.data_ptr(), 1000, logits.data_ptr(), idx.data_ptr(), conf.data_ptr())
    ctx.synchronize()
    a = logits.cpu().numpy()
    assert np.isfinite(a).all()
    for k in range(8):                                   # identical inputs -> identical rows anywhere in the batch
        rows = a[order == k]
        assert (rows == rows[0]).all()
    ctx2 = clf.create_batch_context(96)                  # different micro-batch slicing, same answers
    logits2 = torch.empty_like(logits)
    for _ in range(6):                                   # (a rare hazard shows up in one pass of a few: repeat)
        clf.forward_device(ctx2, x.data_ptr(), 1000, logits2.data_ptr())
        ctx2.synchronize()
        assert torch.equal(logits, logits2)
    # launches of more than 256 segments run the late blocks on two-segment tiles, smaller ones on their one-segment twins
    # (mbconv_cfgs.inc 133-136, mb_plan_twin): the same bits either way
    ctx3 = clf.create_batch_context(600)
    clf.forward_device(ctx3, x.data_ptr(), 1000, logits2.data_ptr())
    ctx3.synchronize()
    assert torch.equal(logits, logits2)
    ctx3.close()
    ii = idx.cpu().numpy(); cc = conf.cpu().numpy()
    assert ((ii >= -1) & (ii < m.n_classes)).all()
    valid = ii >= 0
    assert (np.diff(np.where(valid, cc, 0.0), axis=1) <= 1e-7).all()      # confidence descending
    assert (cc[valid] >= 0.1).all() and (cc[~valid] == 0).all()
    want_top1 = a.argmax(1)
    has = valid[:, 0]
    assert (ii[has, 0] == want_top1[has]).all()
    ctx.close(); ctx2.close(); clf.close()


# ---- fused MBConv blocks (expand -> depthwise -> project in one launch) -----------------
def test_full_model_runs_every_inverted_residual_block_fused(full_model):
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = full_model
    clf = BirdClassifier(path, labels)
    cfgs = clf.fused_blocks()
    assert len(cfgs) == 16, cfgs          # EfficientNet-B0: 15 inverted-residual blocks + stem/first block
    clf.close()


@pytest.mark.parametrize("precision", ["f16x3", "f16"])
def test_column_task_late_blocks_against_the_row_task_ones(full_model, oracle_lib, monkeypatch, precision):
    """The whole-image late blocks (6x32 and 3x16 under 5x5 / 3x3 kernels) run their depthwise phase as column tasks
    (kernels_mbconv.hip COLTH: padding rows skipped at compile time, one task per thread) in 8-wave workgroups.  They are the planner's choice on the full
    model; forcing the row-task entries (34 ... 43) instead must give the same logits to the mode's tolerance, and both must
    match the oracle."""
    from birda_amd import synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = full_model
    segs = synth.synth_segments(5, m.sample_count, m.sample_rate, start=77)     # odd: two-segment tiles see a tail
    ref = oracle_lib.OracleModel(path).forward(segs)
    scale = max(1.0, float(np.abs(ref).max()))
    out = {}
    # row-task entries that ship: 59 / 60 (half-image tiles of the 5x5 6x32 blocks), 40-43 (3x16 x 2), 55 / 33 (3x3 at 6x32) -- their
    # 4-wave whole-image twins 34 / 36 are measured alternatives now (mbconv_cfgs.inc MB_XENTRY, make EXPERIMENTS=1)
    for tag, prefer in (("column", None), ("row", "59,60,40,42,55" if precision == "f16x3" else "35,37,41,43,33")):
        if prefer:
            monkeypatch.setenv("BIRDA_HIP_MB_PREFER", prefer)
        clf = BirdClassifier(path, labels, precision=precision)
        names = [clf.fused_kernel_name(b) for b in clf.fused_blocks()]
        n_col = sum(1 for n in names if int(n.rstrip(">").split(",")[18]) > 0)
        # 2 x 80->480->80 (3x3), 80->480->112, 2 x 112->672->112, 3 x 192->1152->192, 192->1152->320
        assert n_col == (9 if tag == "column" else 0), names
        if tag == "column":     # ... and they are the 8-wave workgroups (wave grid WM x WN = 8)
            assert all(int(n.split("<")[1].split(",")[6]) * int(n.split(",")[7]) == 8 for n in names if int(n.rstrip(">").split(",")[18]) > 0), names
        ctx = clf.create_batch_context(5)
        out[tag] = clf.predict_logits(ctx, segs)
        ctx.close(); clf.close()
    tol = (F16_LOGIT_RTOL if precision == "f16" else LOGIT_RTOL) * scale
    assert np.abs(out["column"] - ref).max() <= tol and np.abs(out["row"] - ref).max() <= tol
    assert np.abs(out["column"] - out["row"]).max() <= tol


def test_fused_blocks_match_unfused_and_oracle_on_small_images(model_dir, oracle_lib, monkeypatch):
    """mini_b0 = the full B0 channel plan on 16x58 ... 1x4 images: partial tiles everywhere, images
    smaller than a tile, and 5 segments so that the two-segments-per-workgroup tiles see an odd tail."""
    from birda_amd import synth
    from birda_amd.classifier import BirdClassifier
    path, _, m, _ = model_dir["mini_b0"]
    segs = synth.synth_segments(5, m.sample_count, m.sample_rate, start=40)
    ref = oracle_lib.OracleModel(path).forward(segs)
    monkeypatch.setenv("BIRDA_HIP_FUSE", "0")
    clf = BirdClassifier(path)
    assert clf.fused_blocks() == []
    ctx = clf.create_batch_context(8)
    unfused = clf.predict_logits(ctx, segs)
    ctx.close(); clf.close()
    _logit_close(unfused, ref)
    monkeypatch.setenv("BIRDA_HIP_FUSE", "1")
    clf = BirdClassifier(path)
    assert len(clf.fused_blocks()) == 16
    ctx = clf.create_batch_context(8)
    fused = clf.predict_logits(ctx, segs)
    # a context smaller than the batch: slices of 2 (+ odd tail) through the same kernels
    ctx2 = clf.create_batch_context(2)
    sliced = clf.predict_logits(ctx2, segs)
    ctx.close(); ctx2.close(); clf.close()
    _logit_close(fused, ref)
    _logit_close(fused, unfused)
    assert np.array_equal(fused, sliced)


def test_fused_block_outputs_match_oracle_tensor_by_tensor(model_dir, oracle_lib, monkeypatch):
    """Every fused block's OUTPUT tensor against the oracle's (BIRDA_HIP_KEEP_FUSED=1: a debug context that
    still runs the fused kernels), both precisions: a wrong tile or row group shows up here even when the
    logits would average it away."""
    from birda_amd import modelfile as mf, synth
    from birda_amd.classifier import BirdClassifier
    monkeypatch.setenv("BIRDA_HIP_KEEP_TENSORS", "1")
    monkeypatch.setenv("BIRDA_HIP_KEEP_FUSED", "1")
    path, _, m, _ = model_dir["mini_b0"]
    n = 3
    segs = synth.synth_segments(n, m.sample_count, m.sample_rate, start=21)
    om = oracle_lib.OracleModel(path)
    outs = [li + 1 for li, L in enumerate(m.layers) if L.op == mf.OP_PWCONV and L.act == mf.ACT_NONE]   # project layers
    refs = {t: om.forward(segs, dump_tensor=t)[1] for t in outs}
    for prec in ("f32", "f16x3"):
        clf = BirdClassifier(path, precision=prec)
        assert len(clf.fused_blocks()) == 16
        ctx = clf.create_batch_context(n)
        clf.predict_logits(ctx, segs)
        for t in outs:
            got, ref = clf.read_tensor(ctx, t, n), refs[t]
            scale = max(1.0, float(np.abs(ref).max()))
            d = np.abs(got - ref)
            assert np.isfinite(got).all(), (prec, t)
            assert d.max() <= SPEC_MAX_ATOL * scale and d.mean() <= SPEC_MEAN_ATOL * scale, (prec, t, d.max(), d.mean())
        ctx.close(); clf.close()


def test_every_fused_tile_configuration(model_dir, oracle_lib, monkeypatch):
    """Force each SHIPPED tile configuration in turn (mbconv_cfgs.inc MB_ENTRY rows; the MB_XENTRY rows are not part of the product
    build); blocks it cannot run fall back to the layer kernels.  The precision is the configuration's own (template argument 15:
    0 f32 MFMA, 3 split f16, 1 plain f16)."""
    from birda_amd import synth
    from birda_amd._lib import load
    from birda_amd.classifier import BirdClassifier
    path, _, m, _ = model_dir["mini_b0"]
    segs = synth.synth_segments(3, m.sample_count, m.sample_rate, start=7)
    ref = oracle_lib.OracleModel(path).forward(segs)
    scale = max(1.0, float(np.abs(ref).max()))
    L = load()
    import ctypes as C
    buf = C.create_string_buffer(128)
    n_total = 0
    while L.bh_mb_config_name(n_total, buf, 128) > 0:
        n_total += 1
    n_base = n_total // 3                      # one copy of the list per activation (GELU first)
    assert n_base >= 133
    used, shipped = set(), set()
    for cfg in range(n_base):
        L.bh_mb_config_name(cfg, buf, 128)
        args = [int(v) for v in buf.value.decode().split(",")]
        if args[0] == 0:                       # MB_NONE: a measured alternative, not in this build
            continue
        shipped.add(cfg)
        prec = {0: "f32", 3: "f16x3", 1: "f16"}[args[15]]
        monkeypatch.setenv("BIRDA_HIP_MB_CFG", str(cfg))
        clf = BirdClassifier(path, precision=prec)
        blocks = clf.fused_blocks()
        if not blocks:
            clf.close()
            continue
        assert set(blocks) == {cfg}
        used.add(cfg)
        ctx = clf.create_batch_context(4)
        got = clf.predict_logits(ctx, segs)
        if prec == "f16":
            assert np.isfinite(got).all() and np.abs(got - ref).max() <= F16_LOGIT_RTOL * scale, cfg
        else:
            _logit_close(got, ref)
        ctx.close(); clf.close()
    # Not runnable on this model's images: the 1-channel stem variants (21, 47, 66, 115, 116: Perch tests), the column-task entries,
    # which need their exact image height (99-114: full-model test; 111-112, 117-130: Perch tests), and the 8x32 / 4x16 whole-image
    # tiles of the Perch stacks whose k-step counts no block of this model has
    print("shipped", sorted(shipped), "\nused on mini_b0", sorted(used))
    required = {3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 23, 25, 27, 29, 31, 33, 35, 37, 39, 40, 41, 42, 43, 45,
                48, 49, 50, 51, 52, 55, 58, 59, 60, 65}
    assert required <= used, sorted(required - used)


def test_precision_modes_on_the_full_model(full_model, oracle_lib):
    """BH_FLAG_F16X3 (split hi/lo f16 operands, three MFMAs per product) must meet the SAME fp32 logit
    tolerance as the f32 MFMA path; BH_FLAG_F16 (BASELINE config 5: fp16 MFMA conv) its own."""
    from birda_amd import synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = full_model
    segs = synth.synth_segments(4, m.sample_count, m.sample_rate, start=200)
    ref = oracle_lib.OracleModel(path).forward(segs)
    scale = max(1.0, float(np.abs(ref).max()))
    out = {}
    for prec in ("f32", "f16x3", "f16"):
        clf = BirdClassifier(path, labels, precision=prec)
        assert len(clf.fused_blocks()) == 16
        ctx = clf.create_batch_context(4)
        out[prec] = clf.predict_logits(ctx, segs)
        ctx.close(); clf.close()
    e32 = _logit_close(out["f32"], ref)
    e163 = _logit_close(out["f16x3"], ref)
    e16 = float(np.abs(out["f16"] - ref).max())
    print(f"max|dlogit| / scale: f32 {e32 / scale:.2e}  f16x3 {e163 / scale:.2e}  f16 {e16 / scale:.2e}")
    assert np.isfinite(out["f16"]).all() and e16 <= F16_LOGIT_RTOL * scale
    # same top-1 class in every mode
    assert (out["f16"].argmax(1) == ref.argmax(1)).all() and (out["f16x3"].argmax(1) == ref.argmax(1)).all()


@pytest.mark.parametrize("act_name", ["swish", "relu6"])
def test_fused_path_with_other_activations(oracle_lib, tmp_path, act_name):
    """EfficientNet's swish and MobileNet's ReLU6 between the convolutions take the SAME fused kernels as GELU (the activation
    is a template argument of mbconv_kernel / head_gap16 / pw_gemm16): every inverted-residual block of the B0 channel plan runs
    fused, in all three precisions, within the same tolerances against the oracle."""
    from birda_amd import modelfile as mf, synth
    from birda_amd.classifier import BirdClassifier
    act = {"swish": mf.ACT_SWISH, "relu6": mf.ACT_RELU6}[act_name]
    m = synth.build_model("mini_b0", act=act)
    path = str(tmp_path / f"mini_b0_{act_name}.bhm")
    mf.write_model(path, m)
    segs = synth.synth_segments(3, m.sample_count, m.sample_rate, start=11)
    ref = oracle_lib.OracleModel(path).forward(segs)
    scale = max(1.0, float(np.abs(ref).max()))
    n_blocks = sum(1 for L in m.layers if L.op == mf.OP_DWCONV)
    for prec in ("f32", "f16x3", "f16"):
        clf = BirdClassifier(path, precision=prec)
        blocks = clf.fused_blocks()
        assert len(blocks) == n_blocks, (prec, blocks)
        # the activation is the 18th template argument (kernels_mbconv.hip: ..., PREC, PERSIST, ACT, COLTH)
        assert all(clf.fused_kernel_name(b).rstrip(">").split(",")[17] == str(act) for b in blocks)
        ctx = clf.create_batch_context(3)
        got = clf.predict_logits(ctx, segs)
        if prec == "f16":
            assert np.isfinite(got).all() and np.abs(got - ref).max() <= F16_LOGIT_RTOL * scale
        else:
            _logit_close(got, ref)
        ctx.close(); clf.close()


def _rescale_trunk(m, scales):
    """A function-preserving re-parameterisation of a synthetic model: every residual-connected group of project outputs
    (the linear 'trunk' tensors between blocks) is multiplied by a power of two -- project weights and bias times s -- and
    every layer that reads such a tensor gets its weights divided by s.  Exact in f32 arithmetic (powers of two), so the
    oracle's logits do not move; the f16 operand planes, though, see weights 2^-12 / 2^10 times their usual size and
    activations up to 2^10 times larger."""
    import copy
    from birda_amd import modelfile as mf
    m2 = copy.deepcopy(m)
    blob = m2.blob.copy()
    comp = {}          # tensor -> component id
    order = []
    for i, L in enumerate(m2.layers):
        if L.op == mf.OP_PWCONV and L.act == mf.ACT_NONE:
            c = comp[L.res_tensor] if L.res_tensor != mf.NO_TENSOR else len(order)
            if c == len(order):
                order.append(c)
            comp[i + 1] = c
    scale = {c: scales[c % len(scales)] for c in order}
    for i, L in enumerate(m2.layers):
        if L.op in (mf.OP_PWCONV, mf.OP_DENSE) and L.in_tensor in comp:      # a reader of a trunk tensor
            blob[L.w_off:L.w_off + L.cin * L.cout] /= np.float32(scale[comp[L.in_tensor]])
        else:
            assert L.in_tensor not in comp, "only 1x1 layers read trunk tensors in the synthetic stacks"
        if (i + 1) in comp:                                                     # a producer
            s = np.float32(scale[comp[i + 1]])
            blob[L.w_off:L.w_off + L.cin * L.cout] *= s
            blob[L.b_off:L.b_off + L.cout] *= s
    m2.blob = blob
    return m2, len(order)


@pytest.mark.parametrize("precision,scales,tol", [("f16x3", (2.0 ** 10, 2.0 ** 12), LOGIT_RTOL), ("f16", (2.0 ** 10, 2.0 ** 12), F16_LOGIT_RTOL),
                                                  ("f16x3", (2.0 ** -12, 2.0 ** 10), 2e-4), ("f16x3", (2.0 ** -6, 2.0 ** 6), LOGIT_RTOL)])
def test_f16_operand_planes_survive_tiny_and_huge_weights(model_dir, oracle_lib, tmp_path, precision, scales, tol):
    """the hi / lo f16 split must hold the fp32 tolerance when a layer's weights are 2^-12 or 2^10 times
    their usual size (the lo half of an unscaled small weight is an f16 subnormal) and when block inputs are ~1e4.  Every
    f16 weight plane is pre-scaled at create by an exact power of two into [2^13, 2^14) and the scale is undone in the
    f32 epilogue, so the logits must agree with the oracle's -- which, the rescaling being exact, are also the ORIGINAL
    model's logits.
    scales (2^10, 2^12): trunk tensors 1 000 - 4 000 times their usual size (block inputs up to ~5e4, just inside the f16
    range), the expand weights that read them 2^-10 / 2^-12 of theirs, project weights 2^10 / 2^12: the fp32 tolerance holds.
    scales (2^-12, 2^10): half of the trunk tensors 2^-12 of their usual size.  WEIGHTS of any size are covered by the
    pre-scale; block INPUTS that small are not (they are split as they arrive, and below 2^-3 an input's lo half is an f16
    subnormal, absolute resolution 6e-8): the error grows to <= 2e-4 of the logit scale -- stated, tested, far inside
    BH_FLAG_F16's 3e-3; a network with such activations belongs on BH_FLAG_F32.  At 2^-6 the fp32 tolerance still holds."""
    from birda_amd import modelfile as mf, synth
    from birda_amd.classifier import BirdClassifier
    path0, labels, m, _ = model_dir["mini_b0"]
    m2, n_comp = _rescale_trunk(m, list(scales))
    assert n_comp >= 7
    path = str(tmp_path / "rescaled.bhm")
    mf.write_model(path, m2)
    segs = synth.synth_segments(6, m.sample_count, m.sample_rate, start=70)
    ref0 = oracle_lib.OracleModel(path0).forward(segs)
    ref = oracle_lib.OracleModel(path).forward(segs)
    scale = max(1.0, float(np.abs(ref0).max()))
    assert np.abs(ref - ref0).max() <= 2e-6 * scale          # the re-parameterisation is exact up to f32 rounding order
    clf = BirdClassifier(path, labels, precision=precision)
    assert len(clf.fused_blocks()) == 16
    ctx = clf.create_batch_context(8)
    got = clf.predict_logits(ctx, segs)
    err = float(np.abs(got - ref).max())
    print(f"{precision}: rescaled model max|dlogit| = {err:.3e} of scale {scale:.2f}")
    assert np.isfinite(got).all() and err <= tol * scale
    ctx.close(); clf.close()


def test_f16_activation_overflow_is_reported_not_silent(model_dir, oracle_lib, tmp_path):
    """Trunk tensors multiplied by 2^14 (block inputs ~1e5 > 65 504): the f16 operand modes cannot represent them.  That must
    come back as BH_ERR_NONFINITE -- from the host entry points and from bh_batch_context_synchronize after a device-side
    forward -- while BH_FLAG_F32 still computes the right logits; and a segment that ARRIVES with NaN samples is not an error."""
    import torch
    from birda_amd import modelfile as mf, synth
    from birda_amd._lib import BirdaHipError
    from birda_amd.classifier import BirdClassifier
    path0, labels, m, _ = model_dir["mini_b0"]
    m2, _ = _rescale_trunk(m, [2.0 ** 14])
    path = str(tmp_path / "overflow.bhm")
    mf.write_model(path, m2)
    segs = synth.synth_segments(5, m.sample_count, m.sample_rate, start=10)
    ref = oracle_lib.OracleModel(path0).forward(segs)
    scale = max(1.0, float(np.abs(ref).max()))
    clf = BirdClassifier(path, labels, precision="f32")
    ctx = clf.create_batch_context(8)
    assert np.abs(clf.predict_logits(ctx, segs) - ref).max() <= LOGIT_RTOL * scale
    ctx.close(); clf.close()
    for prec in ("f16x3", "f16"):
        clf = BirdClassifier(path, labels, precision=prec)
        ctx = clf.create_batch_context(8)
        with pytest.raises(BirdaHipError) as e:
            clf.predict_batch_with_context(ctx, list(segs))
        assert e.value.code == -8 and "f16 operand range" in str(e.value)
        x = torch.from_numpy(segs).cuda()
        lg = torch.empty((5, m.n_classes), device="cuda"); ti = torch.empty((5, 5), dtype=torch.int32, device="cuda"); tc = torch.empty((5, 5), device="cuda")
        clf.forward_device(ctx, x.data_ptr(), 5, lg.data_ptr(), ti.data_ptr(), tc.data_ptr())
        with pytest.raises(BirdaHipError) as e:
            ctx.synchronize()
        assert e.value.code == -8
        ctx.synchronize()                                       # reported once: the counter is cleared
        ctx.close(); clf.close()
    # the healthy model: NaN samples in, NaN row out, no error
    clf = BirdClassifier(path0, labels, precision="f16x3")
    ctx = clf.create_batch_context(8)
    bad = segs.copy(); bad[2, 100] = np.nan
    res = clf.predict_batch_with_context(ctx, list(bad))
    assert len(res) == 5 and res[2].predictions == []
    ctx.close(); clf.close()


def test_auto_precision_reruns_the_rows_beyond_the_f16_range(model_dir, oracle_lib, tmp_path):
    """BH_FLAG_AUTO (bh_config.flags = 0, the library's default): split-f16 compute that never fails a batch on operand range
    (reference dispatch: processor.rs:269-277; recorded degradation: classifier.rs:742-754).  A trunk scale is looked for at
    which SOME rows of a batch overflow the f16 range and others do not.  Then, from the host entry points and from the
    device-resident path: no error; the rows the split-f16 path could represent are bit-identical to a BH_FLAG_F16X3
    classifier's; the others equal a BH_FLAG_F32 classifier's within the fp32 tolerance (and the oracle's); the fall-back is
    counted and named in the provider status.  With every row beyond the range (2^14) the whole batch is f32-grade."""
    import torch
    from birda_amd import modelfile as mf, synth
    from birda_amd._lib import BirdaHipError
    from birda_amd.classifier import BirdClassifier
    path0, labels, m, _ = model_dir["mini_b0"]
    N = 16
    segs = synth.synth_segments(N, m.sample_count, m.sample_rate, start=40)
    ref = oracle_lib.OracleModel(path0).forward(segs)
    scale = max(1.0, float(np.abs(ref).max()))
    x = torch.from_numpy(segs).cuda()

    def device_forward(clf, ctx, expect_error):
        lg = torch.zeros((N, m.n_classes), device="cuda"); ti = torch.zeros((N, 5), dtype=torch.int32, device="cuda"); tc = torch.zeros((N, 5), device="cuda")
        clf.forward_device(ctx, x.data_ptr(), N, lg.data_ptr(), ti.data_ptr(), tc.data_ptr())
        if expect_error:
            try:
                ctx.synchronize()
            except BirdaHipError as e:
                assert e.code == -8
        else:
            ctx.synchronize()
        return lg.cpu().numpy(), ti.cpu().numpy(), tc.cpu().numpy()

    def split_f16_rows(e):
        m2, _ = _rescale_trunk(m, [2.0 ** e])
        path = str(tmp_path / f"overflow_{e}.bhm")
        mf.write_model(path, m2)
        clf = BirdClassifier(path, labels, precision="f16x3")
        ctx = clf.create_batch_context(N)
        lg16, ti16, _ = device_forward(clf, ctx, True)
        ctx.close(); clf.close()
        bad = ~np.isfinite(lg16).all(axis=1)
        assert ((ti16[:, 0] == -2) == bad).all()            # BH_TOPK_NONFINITE marks exactly the non-finite rows
        return path, lg16, bad

    mixed = None
    for e in (12.2, 12.4, 12.6, 12.8, 13.0, 13.2, 13.4, 13.7, 14.0):
        cand = split_f16_rows(e)
        if 0 < cand[2].sum() < N:
            mixed = cand
            break
    paths = [split_f16_rows(16.0)]                          # (nearly) every row beyond the range
    assert paths[0][2].sum() >= N // 2
    if mixed is not None:
        paths.insert(0, mixed)
    for path, lg16, bad in paths:
        c32 = BirdClassifier(path, labels, precision="f32")
        x32 = c32.create_batch_context(N)
        ref32 = c32.predict_logits(x32, segs)
        res32 = c32.predict_batch_with_context(x32, list(segs))
        x32.close(); c32.close()
        clf = BirdClassifier(path, labels, precision="auto")
        assert clf.info.precision == 0 and clf.fallback_segments() == 0
        ctx = clf.create_batch_context(N)
        got = clf.predict_logits(ctx, segs)                  # host entry point, logits
        assert np.isfinite(got).all()
        assert np.abs(got[bad] - ref32[bad]).max() <= LOGIT_RTOL * scale and np.abs(got - ref).max() <= LOGIT_RTOL * scale
        if lg16 is not None:
            assert (got[~bad] == lg16[~bad]).all()           # rows the split-f16 path served: untouched, bit for bit
        nfb = clf.fallback_segments()
        assert nfb == int(bad.sum())
        res = clf.predict_batch_with_context(ctx, list(segs))   # host entry point, top-k rows
        for i in range(N):
            assert [p.index for p in res[i].predictions] == [p.index for p in res32[i].predictions]
            assert np.allclose([p.confidence for p in res[i].predictions], [p.confidence for p in res32[i].predictions], atol=1e-5)
        dl, di, dc = device_forward(clf, ctx, False)         # device-resident path: re-run inside bh_batch_context_synchronize
        assert (di[:, 0] != -2).all() and np.isfinite(dl).all()
        assert (dl == got).all()
        pcm = np.clip(np.round(segs.reshape(-1) * 32767.0), -32768, 32767).astype(np.int16)   # decoded-stream entry point
        seen = []
        rows, _ = clf.predict_pcm16(ctx, pcm, m.sample_rate, on_rows=lambda first, rr, st: seen.append((first, len(rr))))
        assert len(rows) == N and sum(k for _, k in seen) == N
        assert all(0 <= p.index < m.n_classes and np.isfinite(p.confidence) for r in rows for p in r.predictions)

        def boom(first, rr, st):
            raise KeyError("consumer failed")
        with pytest.raises(KeyError):      # an exception in the rows callback reaches the caller (it used to be printed and dropped)
            clf.predict_pcm16(ctx, pcm, m.sample_rate, on_rows=boom)
        assert clf.fallback_segments() >= 3 * nfb
        st = clf.provider_status()
        assert b"f32 kernels" in st.fallback_reason
        # more than 256 DISTINCT device-resident forwards between two synchronises -- the list of what a
        # synchronise repairs is settled by bh_forward_device itself when it is full, not dropped: still BH_OK, every row repaired
        K = 300
        lgs = torch.zeros((K, 2, m.n_classes), device="cuda"); tis = torch.zeros((K, 2, 5), dtype=torch.int32, device="cuda"); tcs = torch.zeros((K, 2, 5), device="cuda")
        for k in range(K):
            clf.forward_device(ctx, x.data_ptr() + (k % (N // 2)) * 2 * m.sample_count * 4, 2, lgs[k].data_ptr(), tis[k].data_ptr(), tcs[k].data_ptr())
        ctx.synchronize()
        assert bool(torch.isfinite(lgs).all()) and bool((tis[:, :, 0] != -2).all())
        for k in (0, 1, 255, 256, 257, K - 1):
            j = (k % (N // 2)) * 2
            assert (lgs[k].cpu().numpy() == got[j:j + 2]).all(), k
        # ... and a context that is destroyed (parked) WITHOUT a synchronise forgets its pending forwards: the next owner's
        # synchronise must not repair into the first owner's buffers
        lg2 = torch.zeros((N, m.n_classes), device="cuda"); ti2 = torch.zeros((N, 5), dtype=torch.int32, device="cuda"); tc2 = torch.zeros((N, 5), device="cuda")
        clf.forward_device(ctx, x.data_ptr(), N, lg2.data_ptr(), ti2.data_ptr(), tc2.data_ptr())
        ctx.close()
        ctx = clf.create_batch_context(N)
        torch.cuda.synchronize()
        snapshot = lg2.clone()
        ctx.synchronize()                                   # nothing of this owner's is pending
        assert torch.equal(torch.nan_to_num(snapshot), torch.nan_to_num(lg2))
        ctx.close(); clf.close()
    assert mixed is not None, "no trunk scale left some rows inside and some outside the f16 range"


@pytest.mark.parametrize("kind", ["birdnet_v24", "perch_v2", "birdnet_v30", "mini_se", "birdnet_v30_sized"])
def test_hip_matches_float64_vectors_of_the_full_models(tmp_path, kind):
    """The HIP path against the committed float64 torch / numpy logits of the models the bench runs (tests/golden/
    full_model_vectors.npz, tools/gen_golden.py full) -- directly, without the oracle in between."""
    from birda_amd import modelfile as mf, synth
    from birda_amd.classifier import BirdClassifier
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_model_vectors.npz"))
    ref = g[f"{kind}_logits"]
    m = synth.build_model(kind)
    path = str(tmp_path / f"{kind}.bhm")
    mf.write_model(path, m)
    segs = synth.synth_segments(ref.shape[0], m.sample_count, m.sample_rate, start=int(g[f"{kind}_start"][0]))
    scale = max(1.0, float(np.abs(ref).max()))
    # (birdnet_v30 applies its sigmoid inside the model: its outputs are probabilities, scale 1, while the plain-f16 error is
    #  made on pre-sigmoid values of size ~10 -- 3e-3 of THAT scale; measured 3.0e-3 of the probabilities, bound stated at 6e-3)
    f16_tol = 2 * F16_LOGIT_RTOL if kind in ("birdnet_v30", "birdnet_v30_sized") else F16_LOGIT_RTOL
    for prec, tol in (("f32", LOGIT_RTOL), ("f16x3", LOGIT_RTOL), ("auto", LOGIT_RTOL), ("f16", f16_tol)):
        clf = BirdClassifier(path, None, precision=prec)
        ctx = clf.create_batch_context(4)
        got = clf.predict_logits(ctx, segs)
        err = float(np.abs(got - ref).max())
        print(f"{kind} {prec}: HIP vs float64 max|dlogit| = {err:.3e} on max|logit| {scale:.2f}")
        assert np.isfinite(got).all() and err <= tol * scale, (kind, prec, err)
        ctx.close(); clf.close()


def test_sixty_four_distinct_segments_of_the_full_model_match_the_oracle(full_model, oracle_lib):
    """The 64 distinct segments of bench.py's timed batch (SURVEY 8d's generator, seeds 0 .. 63) through the full v2.4-shaped model
    in all three precisions against the oracle: max |dlogit| within the mode's tolerance and the same top-1 class on every one
    (bench.py reports the same figure as cpu_baseline.max_abs_dlogit_vs_oracle; VERDICT r3 weak #12 wants it in the suite)."""
    from birda_amd import synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = full_model
    segs = synth.synth_segments(64, m.sample_count, m.sample_rate)
    ref = oracle_lib.OracleModel(path).forward(segs)
    scale = max(1.0, float(np.abs(ref).max()))
    for prec, tol in (("f32", LOGIT_RTOL), ("f16x3", LOGIT_RTOL), ("f16", F16_LOGIT_RTOL)):
        clf = BirdClassifier(path, labels, precision=prec)
        ctx = clf.create_batch_context(64)
        got = clf.predict_logits(ctx, segs)
        err = float(np.abs(got - ref).max())
        print(f"64 segments, {prec}: max|dlogit| / scale = {err / scale:.3e}")
        assert np.isfinite(got).all() and err <= tol * scale, (prec, err)
        if prec != "f16":
            assert (got.argmax(axis=1) == ref.argmax(axis=1)).all()
        ctx.close(); clf.close()


def test_launches_of_a_few_segments_take_the_narrow_tiles_and_give_the_same_bits(full_model):
    """Round 4: a launch whose narrow tiles number at most 256 (512 until round 5; a one-minute file is 20 segments) runs the late blocks on 8-column
    tiles -- four / two workgroups per image, mbconv_cfgs.inc entries 193-202, kernels_mbconv.hip mb_plan_narrow -- and a launch of
    at most 256 segments the one-segment twins, a larger one the two-segment tiles.  The same segments must come out BIT-identical
    whichever of the three a call takes (a pixel's sums do not depend on the tile it is computed in), in both f16 modes and f32."""
    import ctypes as C
    from birda_amd import _lib, synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = full_model
    lib = _lib.load()
    cfgs = (C.c_int32 * 128)(); layers = (C.c_int32 * 128)()
    n = lib.bh_plan_fused_blocks(path.encode(), 1, cfgs, layers, 128)        # BH_FLAG_F16X3
    n_base = 0
    buf = C.create_string_buffer(128)
    while lib.bh_mb_config_name(n_base, buf, 128) > 0:
        n_base += 1
    n_base //= 3
    listed = {int(cfgs[i]) % n_base for i in range(n)}
    assert {193, 195, 197, 199, 201} <= listed, sorted(listed)             # the narrow twins of the nine whole-image late blocks
    segs = synth.synth_segments(300, m.sample_count, m.sample_rate)
    for prec in ("f16x3", "f16", "f32"):
        clf = BirdClassifier(path, labels, precision=prec)
        big = clf.create_batch_context(300)
        ref = clf.predict_logits(big, segs)                                 # two-segment tiles
        for k in (1, 20, 64, 128, 256):                                     # narrow / narrow / narrow (6x32: 256 tiles) / narrow (3x16 only) / one-segment twins
            ctx = clf.create_batch_context(k)
            got = clf.predict_logits(ctx, segs[:k])
            assert np.array_equal(got, ref[:k]), (prec, k, float(np.abs(got - ref[:k]).max()))
            ctx.close()
        big.close(); clf.close()


def test_squeeze_excite_and_swish_stack_matches_oracle(model_dir, oracle_lib):
    """The EfficientNet original: swish activations and a squeeze-excite gate (pool -> 1x1 -> swish -> 1x1 -> sigmoid -> multiply)
    in every block.  Round 5: such blocks run as pass A of the fused kernel (expand -> depthwise, the depthwise output to HBM once,
    per-tile channel sums), one gate launch and a project GEMM on D x gate -- against the oracle in every precision mode, and against
    the same blocks run layer by layer (BIRDA_HIP_FUSE_SE=0: expand / depthwise / pool / 1x1 / 1x1 / scale / project)."""
    from birda_amd import synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = model_dir["mini_se"]
    from birda_amd import modelfile as mf
    assert sum(1 for L in m.layers if L.op == mf.OP_SCALE) == 4
    segs = synth.synth_segments(5, m.sample_count, m.sample_rate, start=21)
    ref = oracle_lib.OracleModel(path).forward(segs)
    scale = max(1.0, float(np.abs(ref).max()))
    for prec, tol in (("f32", LOGIT_RTOL), ("f16x3", LOGIT_RTOL), ("auto", LOGIT_RTOL), ("f16", F16_LOGIT_RTOL)):
        clf = BirdClassifier(path, labels, precision=prec)
        assert len(clf.fused_blocks()) >= 2, (prec, clf.fused_blocks())
        ctx = clf.create_batch_context(3)
        got = clf.predict_logits(ctx, segs)
        assert np.isfinite(got).all() and np.abs(got - ref).max() <= tol * scale, (prec, float(np.abs(got - ref).max()))
        again = clf.predict_logits(ctx, segs)          # (fixed summation order in the pooled sums: run to run, bit for bit)
        assert (again == got).all()
        ctx.close()
        if prec == "f16x3":
            # many segments, so that the workgroups of pass A, the gate launches and the gated GEMM really overlap in time: the arena
            # plan must keep the block input, D, the per-tile sums (written EARLIER than the layer whose slot they use), the pooled
            # sums, the hidden layer and the gate apart -- and a segment's row must not depend on the launch it ran in
            many = synth.synth_segments(96, m.sample_count, m.sample_rate, start=21)
            ref_many = oracle_lib.OracleModel(path).forward(many)
            big = clf.create_batch_context(96)
            for _ in range(3):
                g96 = clf.predict_logits(big, many)
                assert np.abs(g96 - ref_many).max() <= tol * max(1.0, float(np.abs(ref_many).max()))
            assert (g96[:5] == got).all()
            big.close()
        clf.close()
    import subprocess, sys, textwrap
    # the layer-by-layer path of the same blocks, in a process of its own (the switch is read once per process)
    code = textwrap.dedent(f"""
        import numpy as np, sys
        sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
        from birda_amd import synth
        from birda_amd.classifier import BirdClassifier
        clf = BirdClassifier({path!r}, None, precision="f16x3")
        assert clf.fused_blocks() == [], clf.fused_blocks()
        ctx = clf.create_batch_context(3)
        m = clf.info
        np.save({str(path) + ".layers.npy"!r}, clf.predict_logits(ctx, synth.synth_segments(5, m.sample_count, m.sample_rate, start=21)))
    """)
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, BIRDA_HIP_FUSE_SE="0"), timeout=300)
    layers = np.load(str(path) + ".layers.npy")
    assert np.abs(layers - ref).max() <= LOGIT_RTOL * scale


def test_fused_head_pool_on_a_small_arena(model_dir, oracle_lib):
    """The head conv + GELU + pool launch reads the conv's input while finished workgroups already store pooled rows:
    the arena plan must keep the two apart (on this toy stack first-fit used to put both at offset 0).  Many
    segments, so that workgroups of the launch really overlap in time; every row against the oracle."""
    from birda_amd import synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = model_dir["mini_hg"]
    segs = synth.synth_segments(96, m.sample_count, m.sample_rate, start=40)
    ref = oracle_lib.OracleModel(path).forward(segs)
    scale = max(1.0, float(np.abs(ref).max()))
    for prec, tol in (("f16x3", LOGIT_RTOL), ("f16", F16_LOGIT_RTOL), ("f32", LOGIT_RTOL)):
        clf = BirdClassifier(path, labels, precision=prec)
        ctx = clf.create_batch_context(96)
        for _ in range(3):
            got = clf.predict_logits(ctx, segs)
            assert np.abs(got - ref).max() <= tol * scale, (prec, float(np.abs(got - ref).max()), scale)
        ctx.close(); clf.close()


def test_fused_head_pool_matches_the_layer_kernels(full_model, oracle_lib, monkeypatch):
    """Head 1x1 conv + GELU + global average pool in one launch (f16 modes) against the same layers run
    one by one (BIRDA_HIP_HEAD_GAP=0) and against the oracle; a ragged batch (n not a multiple of the 8
    segments a workgroup owns) exercises the clamped padding rows."""
    from birda_amd import synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = full_model
    segs = synth.synth_segments(11, m.sample_count, m.sample_rate, start=300)
    ref = oracle_lib.OracleModel(path).forward(segs[:3])
    scale = max(1.0, float(np.abs(ref).max()))
    for prec, tol in (("f16x3", LOGIT_RTOL), ("f16", F16_LOGIT_RTOL)):
        got = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("BIRDA_HIP_HEAD_GAP", flag)
            clf = BirdClassifier(path, labels, precision=prec)
            ctx = clf.create_batch_context(16)
            got[flag] = clf.predict_logits(ctx, segs)
            ctx.close(); clf.close()
        assert np.isfinite(got["1"]).all()
        assert np.abs(got["1"] - got["0"]).max() <= tol * scale, prec
        assert np.abs(got["1"][:3] - ref).max() <= tol * scale, prec


# ---- C4: Perch-shaped model (5 s / 32 kHz, one 128-mel branch, 14 795 classes, softmax) --------
def test_perch_shaped_model_matches_oracle(oracle_lib, tmp_path, monkeypatch):
    """the B0-sized stand-in of rounds 1-2 (89 MB, 1.0 GFLOP): fast, every block fused in the f16 modes"""
    from birda_amd import modelfile as mf, synth
    from birda_amd.classifier import BirdClassifier
    m = synth.build_model("perch_v2_tiny")
    path = str(tmp_path / "perch.bhm")
    mf.write_model(path, m)
    assert (m.sample_rate, m.sample_count, m.n_classes) == (32000, 160000, 14795)
    segs = synth.synth_segments(3, m.sample_count, m.sample_rate, start=5)
    clf = BirdClassifier(path, None, top_k=5, min_confidence=0.0, precision="f32")
    assert clf.sample_rate() == 32000 and clf.sample_count() == 160000 and abs(clf.segment_duration() - 5.0) < 1e-6
    print("perch-shaped fused blocks:", clf.fused_blocks())
    assert len(clf.fused_blocks()) == 16      # f32: every block since round 4 (entries 191 / 192 for the 4x16 images)
    assert 21 in clf.fused_blocks()           # the 1-channel stem variant
    ctx = clf.create_batch_context(4)
    logits = clf.predict_logits(ctx, segs)
    ref = oracle_lib.OracleModel(path).forward(segs)
    err = _logit_close(logits, ref)
    print(f"perch-shaped max|dlogit| = {err:.3e} on max|logit| {np.abs(ref).max():.2f}")
    for prec, cfg, tol in (("f16x3", 66, LOGIT_RTOL), ("f16", 47, F16_LOGIT_RTOL)):
        c2 = BirdClassifier(path, None, precision=prec)
        assert cfg in c2.fused_blocks(), (prec, c2.fused_blocks())   # the 1-channel stem, f16 variants
        assert len(c2.fused_blocks()) == 16, c2.fused_blocks()       # whole-image configurations 70-78 cover the 8x32 / 4x16 stages
        x2 = c2.create_batch_context(4)
        got = c2.predict_logits(x2, segs)
        assert np.isfinite(got).all() and np.abs(got - ref).max() <= tol * max(1.0, float(np.abs(ref).max())), prec
        x2.close(); c2.close()
    res = clf.predict_batch_with_context(ctx, list(segs))
    for i, r in enumerate(res):                      # softmax confidences of the kept top-5
        idx, conf = oracle_lib.topk(ref[i], 2, 5, 0.0)
        assert len(r.predictions) == 5
        assert np.allclose([p.confidence for p in r.predictions], conf, rtol=2e-3, atol=1e-7)
    ctx.close(); clf.close()


def test_birdnet_v30_shaped_model_matches_oracle(oracle_lib, tmp_path):
    """VERDICT r2 missing #6: the v3.0 contract (manifests/BirdNET-v3.0-Models.models.json: 5 s / 32 kHz, 160 000 samples, 11 560
    classes, `predictions` already sigmoid-activated inside the graph, a 1 280-d `embeddings` output).  Output activation NONE:
    the values the model emits ARE the confidences; top-k and the min-confidence threshold apply to them as they are."""
    from birda_amd import modelfile as mf, synth
    from birda_amd.classifier import BirdClassifier
    m = synth.build_model("birdnet_v30")
    path = str(tmp_path / "v30.bhm")
    mf.write_model(path, m)
    assert (m.sample_rate, m.sample_count, m.n_classes, m.output_activation, m.layers[-1].act) == (32000, 160000, 11560, mf.OUT_NONE, mf.ACT_SIGMOID)
    segs = synth.synth_segments(3, m.sample_count, m.sample_rate, start=11)
    om = oracle_lib.OracleModel(path)
    ref = om.forward(segs)
    assert ref.min() >= 0.0 and ref.max() <= 1.0                 # probabilities already
    # the comparison is on PROBABILITIES: |dp| <= 0.25 |dlogit|, and the synthetic head's logits reach ~140, so the stated logit
    # tolerances (2e-5 / 3e-3 of the logit scale) become 7e-4 / 0.1 here; measured 1e-5 / 3e-3
    for prec, tol in (("f32", 7e-4), ("f16x3", 7e-4), ("f16", 0.1)):
        clf = BirdClassifier(path, None, top_k=5, min_confidence=0.0, precision=prec)
        info = clf.info
        assert (info.model_type, info.output_activation, info.embedding_dim, info.n_classes) == (2, 0, 1280, 11560)
        assert clf.default_batch_size() == 512
        ctx = clf.create_batch_context(4)
        got, emb = clf.predict_logits(ctx, segs, want_embeddings=True)
        assert emb.shape == (3, 1280)
        assert np.isfinite(got).all() and np.abs(got - ref).max() <= tol, (prec, np.abs(got - ref).max())   # outputs in [0, 1]: absolute
        res = clf.predict_batch_with_context(ctx, list(segs))
        for i, r in enumerate(res):
            idx, conf = oracle_lib.topk(ref[i], 0, 5, 0.0)
            assert len(r.predictions) == 5
            # (saturated classes tie at 1.0: the kept indices are compared through their confidences)
            assert np.allclose([p.confidence for p in r.predictions], np.sort(got[i])[::-1][:5], rtol=0, atol=1e-7)
            assert all(abs(float(ref[i][p.index]) - p.confidence) <= tol for p in r.predictions)
        ctx.close(); clf.close()


def test_perch_sized_model_matches_oracle(oracle_lib, tmp_path):
    """BASELINE configs[3] at the published model's size (VERDICT r2 missing #4): EfficientNet-B3 stage plan with swish on the
    5 s / 32 kHz front-end, 1 536-d embedding, 6 144-wide hidden layer, 14 795 classes -- 437 MB, 2.67 GFLOP per segment
    (manifests/Perch-v2-Models.models.json size_bytes 413 350 933; README 42 vs 183 segments/s).  All 25 expand -> depthwise ->
    project blocks and the one depthwise -> project block without an expand convolution fuse in the f16 modes (tile entries
    115-133), and since round 4 in the f32 mode too (entries 181-190: the path BH_FLAG_AUTO re-runs a row on).  Three precisions
    against the oracle."""
    from birda_amd import modelfile as mf, synth
    from birda_amd.classifier import BirdClassifier
    m = synth.build_model("perch_v2")
    path = str(tmp_path / "perch_sized.bhm")
    mf.write_model(path, m)
    assert 400e6 < os.path.getsize(path) < 460e6
    assert (m.sample_rate, m.sample_count, m.n_classes, m.layers[-1].cin) == (32000, 160000, 14795, 6144)
    segs = synth.synth_segments(3, m.sample_count, m.sample_rate, start=5)
    ref = oracle_lib.OracleModel(path).forward(segs)
    scale = max(1.0, float(np.abs(ref).max()))
    for prec, n_fused, tol in (("f32", 26, LOGIT_RTOL), ("f16x3", 26, LOGIT_RTOL), ("f16", 26, F16_LOGIT_RTOL)):
        clf = BirdClassifier(path, None, top_k=5, min_confidence=0.0, precision=prec)
        info = clf.info
        assert info.embedding_dim == 1536 and info.model_type == 1
        blocks = clf.fused_blocks()
        print(f"perch-sized {prec}: fused blocks {blocks}, {2 * info.macs_per_segment / 1e9:.3f} GFLOP per segment (conv + dense)")
        assert len(blocks) == n_fused, (prec, blocks)
        ctx = clf.create_batch_context(4)
        got = clf.predict_logits(ctx, segs)
        err = float(np.abs(got - ref).max())
        print(f"perch-sized {prec}: max|dlogit| = {err:.3e} on max|logit| {scale:.2f}")
        assert np.isfinite(got).all() and err <= tol * scale, (prec, err)
        if prec == "f16x3":
            res = clf.predict_batch_with_context(ctx, list(segs))
            for i, r in enumerate(res):                      # softmax confidences of the kept top-5
                idx, conf = oracle_lib.topk(ref[i], 2, 5, 0.0)
                assert [p.index for p in r.predictions] == list(idx)
                assert np.allclose([p.confidence for p in r.predictions], conf, rtol=2e-3, atol=1e-7)
        if prec != "f32":
            # a segment's row does not depend on the launch it ran in: 3 segments take the 128 x 128 staged tiles of the gated project
            # GEMM (fewer than 4 096 rows), 80 and 300 take the streaming / row-streaming kernels -- over D in NHWC and, for N = 96 ..
            # 232, blocked (kernels.hpp MbDesc::dblk); pass A of the 4x16 stages runs its one-segment twins up to 256 segments
            # (mb_twin_sums_match: the pooled sums in the two-segment tile's order) and the two-segment tiles beyond
            for nbig in (80, 300):
                big = clf.create_batch_context(nbig)
                gb = clf.predict_logits(big, np.ascontiguousarray(np.tile(segs, (nbig // 3 + 1, 1))[:nbig]))
                assert all((gb[i] == got[i % 3]).all() for i in range(nbig)), (prec, nbig)
                big.close()
        ctx.close(); clf.close()


def test_birdnet_v30_at_its_published_size_matches_oracle(oracle_lib, tmp_path):
    """BirdNET v3.0 at the size of the published file (VERDICT r5 missing #3; reference manifests/BirdNET-v3.0-Models.models.json:
    557 212 256 bytes, 1 280-d `embeddings`, 11 560 classes whose sigmoid sits inside the graph, 5 s at 32 kHz): the B0 stand-in of
    rounds 3-5 was a quarter of a percent of that.  [EXT] The trunk is not published offline; `birdnet_v30_sized` is the
    EfficientNetV2-L stage plan (the one trunk that fits 139 M parameters AND a 1 280-d embedding) with its fused-MBConv stages
    spelled as MBConv, swish and squeeze-excite gates: 553 MB, 83 blocks, 21.5 GFLOP per segment.  76 blocks run fused (the seven of
    the 640-channel stage are beyond the widest tile entries -- 384 channels in, 512 out -- and run layer by layer); logits
    (probabilities: the output layer carries the sigmoid) against the oracle in the f32-grade modes, bit-identical across launch
    sizes."""
    from birda_amd import modelfile as mf, synth
    from birda_amd.classifier import BirdClassifier
    m = synth.build_model("birdnet_v30_sized")
    path = str(tmp_path / "v30_sized.bhm")
    mf.write_model(path, m)
    assert 540e6 < os.path.getsize(path) < 575e6
    assert (m.sample_rate, m.sample_count, m.n_classes, m.embedding_dim, m.output_activation) == (32000, 160000, 11560, 1280, mf.OUT_NONE)
    segs = synth.synth_segments(2, m.sample_count, m.sample_rate, start=15)
    ref, ref_emb = oracle_lib.OracleModel(path).forward(segs, want_embeddings=True)
    assert ref.min() >= 0.0 and ref.max() <= 1.0
    for prec in ("f16x3", "auto", "f32"):
        clf = BirdClassifier(path, None, top_k=5, min_confidence=0.0, precision=prec)
        blocks = clf.fused_blocks()
        print(f"v3.0-sized {prec}: {len(blocks)} of 83 blocks fused, {2 * clf.info.macs_per_segment / 1e9:.2f} GFLOP per segment")
        assert len(blocks) >= (76 if prec != "f32" else 60), (prec, len(blocks))
        ctx = clf.create_batch_context(2)
        got, emb = clf.predict_logits(ctx, segs, want_embeddings=True)
        err = float(np.abs(got - ref).max())
        escale = max(1.0, float(np.abs(ref_emb).max()))
        print(f"v3.0-sized {prec}: max |dp| = {err:.3e} (probabilities), embeddings {float(np.abs(emb - ref_emb).max()) / escale:.3e} of their scale")
        assert np.isfinite(got).all() and err <= LOGIT_RTOL and np.abs(emb - ref_emb).max() <= LOGIT_RTOL * escale, (prec, err)
        if prec == "f16x3":
            big = clf.create_batch_context(40)
            gb = clf.predict_logits(big, np.ascontiguousarray(np.tile(segs, (20, 1))))
            assert all((gb[i] == got[i % 2]).all() for i in range(40))
            big.close()
        ctx.close(); clf.close()


@pytest.mark.parametrize("mel32", ["0", "1"])
def test_both_front_end_kernels_match_oracle(full_model, oracle_lib, monkeypatch, mel32):
    """mel_kernel (16-frame tiles) and mel32_kernel (32-frame tiles, Y rows staged along k) are chosen by hop; force each
    on the BirdNET front-end (96 mels, two branches) and hold it to the same logit tolerance and slicing invariance."""
    import torch
    from birda_amd import synth
    from birda_amd.classifier import BirdClassifier
    path, _, m, _ = full_model
    monkeypatch.setenv("BIRDA_HIP_MEL32", mel32)
    clf = BirdClassifier(path, precision="f16x3")
    segs = synth.synth_segments(5, m.sample_count, m.sample_rate, start=300)
    ctx = clf.create_batch_context(8)
    got = clf.predict_logits(ctx, segs)
    ref = oracle_lib.OracleModel(path).forward(segs)
    print(f"BIRDA_HIP_MEL32={mel32}: max|dlogit| = {_logit_close(got, ref):.3e}")
    x = torch.from_numpy(segs[np.arange(300) % 5]).cuda()
    la = torch.empty((300, m.n_classes), device="cuda"); lb = torch.empty_like(la)
    ctx2 = clf.create_batch_context(96)
    big = clf.create_batch_context(300)
    clf.forward_device(big, x.data_ptr(), 300, la.data_ptr()); big.synchronize()
    for _ in range(3):
        clf.forward_device(ctx2, x.data_ptr(), 300, lb.data_ptr()); ctx2.synchronize()
        assert torch.equal(la, lb)
    assert np.array_equal(la[:5].cpu().numpy(), got)
    ctx.close(); ctx2.close(); big.close(); clf.close()


def test_front_end_on_quiet_audio_with_a_dc_offset(full_model, oracle_lib):
    """Field recordings are not full-scale: a segment of 1e-4 amplitude riding on a DC offset of 0.3, one that is 1e-5 of noise and
    a loud one, in ONE batch.  The per-segment min / max normalisation happens BEFORE the f16 hi / lo split of the folded frames
    (which is why it stays in front of the GEMM: moved behind it -- VERDICT r4 next #4 -- the DC term Gf^T 1, 3 000 x the signal here,
    would be subtracted from a 22-bit product), so every one of them must come out as close to the oracle as full-scale audio does."""
    from birda_amd.classifier import BirdClassifier
    path, _, m, _ = full_model
    rng = np.random.default_rng(77)
    t = np.arange(m.sample_count) / m.sample_rate
    tone = np.sin(2 * np.pi * 1234.0 * t) + 0.5 * np.sin(2 * np.pi * 6100.0 * t)
    segs = np.stack([0.3 + 1e-4 * (0.5 * tone + rng.standard_normal(m.sample_count)),
                     -0.05 + 1e-5 * rng.standard_normal(m.sample_count),
                     np.clip(0.4 * tone + 0.1 * rng.standard_normal(m.sample_count), -1, 1),
                     1e-4 * tone]).astype(np.float32)
    ref = oracle_lib.OracleModel(path).forward(segs)
    scale = max(1.0, float(np.abs(ref).max()))
    for prec in ("f16x3", "f32"):
        clf = BirdClassifier(path, precision=prec)
        ctx = clf.create_batch_context(4)
        got = clf.predict_logits(ctx, segs)
        err = np.abs(got - ref).max(axis=1)
        print(f"{prec}: per-segment max|dlogit| = {err}, scale {scale:.2f}")
        assert np.isfinite(got).all() and err.max() <= LOGIT_RTOL * scale, (prec, err)
        ctx.close(); clf.close()


def test_model_converted_from_onnx_runs_identically(model_dir, tmp_path):
    """model -> ONNX bytes -> birda_amd.convert -> BHM1: the library must plan the same fused blocks and return
    bit-identical logits for the converted file (same weights, layer table rebuilt from the ONNX graph)."""
    from birda_amd import convert, modelfile as mf, onnx_io as ox, synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = model_dir["mini_b0"]
    m2 = convert.model_from_graph(ox.load(ox.dump(convert.graph_from_model(m))), m)
    p2 = str(tmp_path / "converted.bhm")
    mf.write_model(p2, m2)
    segs = synth.synth_segments(5, m.sample_count, m.sample_rate, start=70)
    out = []
    for p in (path, p2):
        clf = BirdClassifier(p, labels, precision="f16x3")
        ctx = clf.create_batch_context(8)
        out.append((clf.fused_blocks(), clf.predict_logits(ctx, segs)))
        ctx.close(); clf.close()
    assert out[0][0] == out[1][0] and len(out[0][0]) == 16
    assert np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("spelling", ["stft", "fused"])
def test_model_converted_from_an_audio_graph_without_a_manifest(model_dir, tmp_path, spelling):
    """The whole model as an ONNX graph over the AUDIO input (front-end spelled as an STFT node / as one fused Conv), converted
    with no front-end manifest: frame length / step, the mel operator, exponent, affine, flip and the normalisation epsilon are
    read off the graph by probing (birda_amd/frontend_recover.py).  The device must give the original model's logits -- within
    the tolerance of the fitted mel matrix (least squares, float32: ~1e-8 per weight) -- and the oracle's."""
    from oracle import oracle as O
    from birda_amd import convert, modelfile as mf, onnx_io as ox, synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = model_dir["mini_b0"]
    m2 = convert.model_from_graph(ox.load(ox.dump(convert.graph_from_model(m, frontend_spelling=spelling))), None,
                                  sample_rate=m.sample_rate)
    p2 = str(tmp_path / "converted.bhm")
    mf.write_model(p2, m2)
    segs = synth.synth_segments(5, m.sample_count, m.sample_rate, start=90)
    out = []
    for p in (path, p2):
        clf = BirdClassifier(p, labels, precision="f16x3")
        ctx = clf.create_batch_context(8)
        out.append((clf.fused_blocks(), clf.predict_logits(ctx, segs)))
        ctx.close(); clf.close()
    assert out[0][0] == out[1][0] and len(out[0][0]) == 16
    scale = max(1.0, float(np.abs(out[0][1]).max()))
    assert np.abs(out[0][1] - out[1][1]).max() <= 2e-5 * scale
    assert np.abs(O.OracleModel(p2).forward(segs) - out[1][1]).max() <= LOGIT_RTOL * scale


# ---- range filter / species list on the kept top-k --------------------------------------------
def test_device_range_filter_reproduces_the_reference_unit_tests(clf_mini, model_dir):
    """geomodel_filter.rs:126-300 as data (tests/golden/reference_unit_cases.json): each case's predictions are
    planted as logits, pass through the top-k kernel and its filter tail, and must come out as the reference asserts."""
    import json
    from test_oracle_golden import filter_case_tables
    with open(os.path.join(GOLDEN, "reference_unit_cases.json")) as f:
        cases = json.load(f)["filter_predictions"]
    _, _, m, _ = model_dir["mini"]
    try:
        for c in cases:
            species, index, scores, idx, conf = filter_case_tables(c)
            table = np.full(m.n_classes, np.nan, np.float32)
            table[:len(species)] = scores[:len(species)]
            logits = np.full((1, m.n_classes), -20.0, np.float32)            # sigmoid 2e-9: below min_confidence
            for i, p in zip(idx, conf):
                logits[0, i] = np.log(np.float64(p) / (1.0 - np.float64(p)))
            clf_mini.set_range_filter(table, 0.01, c["policy"], c["rerank"])
            got = clf_mini.topk_from_logits(logits)[0].predictions
            assert [species[p.index] for p in got] == [s for s, _ in c["out"]], c["src"]
            assert np.allclose([p.confidence for p in got], [p for _, p in c["out"]], atol=1e-6), c["src"]
    finally:
        clf_mini.clear_filters()


@pytest.mark.parametrize("policy,rerank", [("keep", False), ("drop", False), ("keep", True), ("drop", True)])
def test_device_range_filter_matches_oracle_on_every_entry_point(clf_tiny, model_dir, oracle_lib, policy, rerank):
    import torch
    from birda_amd import synth
    _, _, m, _ = model_dir["birdnet_v24_tiny"]
    rng = np.random.default_rng(11)
    segs = synth.synth_segments(12, m.sample_count, m.sample_rate, start=70)
    ctx = clf_tiny.create_batch_context(12)
    plain = clf_tiny.predict_batch_with_context(ctx, list(segs))
    assert sum(len(r.predictions) for r in plain) >= 12          # the tiny model's sigmoid outputs cluster near 0.5
    table = rng.random(m.n_classes).astype(np.float32)
    table[rng.random(m.n_classes) < 0.3] = np.nan                # species without geomodel entry
    table[rng.random(m.n_classes) < 0.3] = 0.001                 # out of range
    try:
        clf_tiny.set_range_filter(table, 0.03, policy, rerank)
        got = clf_tiny.predict_batch_with_context(ctx, list(segs))
        one = clf_tiny.predict(segs[0])
        x = torch.from_numpy(segs).cuda()
        lg = torch.empty((12, m.n_classes), device="cuda")
        di = torch.empty((12, 5), dtype=torch.int32, device="cuda"); dc = torch.empty((12, 5), device="cuda")
        clf_tiny.forward_device(ctx, x.data_ptr(), 12, lg.data_ptr(), di.data_ptr(), dc.data_ptr()); ctx.synchronize()
        di, dc = di.cpu().numpy(), dc.cpu().numpy()
        dropped = 0
        for i, (p, g) in enumerate(zip(plain, got)):
            oi, oc = oracle_lib.filter_predictions([q.index for q in p.predictions], [q.confidence for q in p.predictions],
                                                   table, 0.03, policy == "keep", rerank)
            assert [q.index for q in g.predictions] == oi.tolist()
            assert np.array_equal(np.asarray([q.confidence for q in g.predictions], np.float32), oc)   # one f32 multiply: exact
            k = len(oi)
            assert di[i, :k].tolist() == oi.tolist() and (di[i, k:] == -1).all() and (dc[i, k:] == 0).all()
            assert np.array_equal(dc[i, :k], oc)
            dropped += len(p.predictions) - k
        assert [q.index for q in one.predictions] == [q.index for q in got[0].predictions]
        assert dropped > 0
        # species list: ignored while a range filter is set (classifier.rs:587, :617), applied once it is cleared
        keep = rng.random(m.n_classes) < 0.5
        clf_tiny.set_species_list(keep)
        again = clf_tiny.predict_batch_with_context(ctx, list(segs))
        assert [[q.index for q in r.predictions] for r in again] == [[q.index for q in r.predictions] for r in got]
        clf_tiny.clear_filters()
        clf_tiny.set_species_list(keep)
        listed = clf_tiny.predict_batch_with_context(ctx, list(segs))
        for p, g in zip(plain, listed):
            oi, oc = oracle_lib.species_retain([q.index for q in p.predictions], [q.confidence for q in p.predictions], keep)
            assert [q.index for q in g.predictions] == oi.tolist()
            assert np.array_equal(np.asarray([q.confidence for q in g.predictions], np.float32), oc)
    finally:
        clf_tiny.clear_filters()
    after = clf_tiny.predict_batch_with_context(ctx, list(segs))
    assert [[q.index for q in r.predictions] for r in after] == [[q.index for q in r.predictions] for r in plain]
    with pytest.raises(Exception):
        clf_tiny.set_range_filter(table[:-1], 0.03)
    ctx.close()


def test_directory_mode_two_ranks_cover_every_file_once(clf_tiny, model_dir, tmp_path):
    """coordinator.rs:146-190 + SURVEY 8e directory mode: the walk finds the audio files, two ranks split them by
    cumulative duration, each file is processed exactly once and gives the same CSV as a single-rank run."""
    from birda_amd import pipeline, synth
    _, _, m, _ = model_dir["birdnet_v24_tiny"]
    rec = tmp_path / "recordings"; (rec / "night").mkdir(parents=True)
    lengths = {"a.wav": 2, "night/b.WAV": 5, "night/c.wav": 1, "d.wav": 3}
    for k, (rel, n) in enumerate(lengths.items()):
        x = synth.synth_segments(n, m.sample_count, m.sample_rate, start=10 * k).reshape(-1)
        synth.write_wav_pcm16(str(rec / rel), x, m.sample_rate)
    (rec / "night" / "log.txt").write_text("not audio")
    files = pipeline.collect_input_files([str(rec)])
    assert sorted(files) == sorted(str(rec / r) for r in lengths)
    single = tmp_path / "single"; single.mkdir()
    ref = pipeline.process_files(clf_tiny, files, output_dir=str(single), min_confidence=0.05, batch_size=4)
    assert [r.segments for r in ref] == [lengths[os.path.relpath(f, rec)] for f in files]
    seen = []
    for rank in range(2):
        out = tmp_path / f"rank{rank}"; out.mkdir()
        res = pipeline.process_files(clf_tiny, files, rank=rank, world=2, output_dir=str(out), min_confidence=0.05, batch_size=4)
        assert res, "11 segments over two ranks: neither share is empty"
        for r in res:
            name = os.path.basename(r.output_path)
            seen.append(name)
            assert open(r.output_path, "rb").read() == open(single / name, "rb").read()
    assert sorted(seen) == sorted(os.path.basename(r.output_path) for r in ref)
    # the same share with its short files packed into shared forwards (bhh_process_files): the same CSV files
    pk = tmp_path / "packed_rank0"; pk.mkdir()
    res = pipeline.process_files(clf_tiny, files, rank=0, world=2, output_dir=str(pk), min_confidence=0.05, packed=True)
    assert res and all(open(r.output_path, "rb").read() == open(single / os.path.basename(r.output_path), "rb").read() for r in res)
    # resume (should_process, coordinator.rs:96-143): with the outputs in place nothing is left to do unless forced, and a
    # missing output brings exactly its file back
    assert pipeline.process_files(clf_tiny, files, output_dir=str(single), min_confidence=0.05, batch_size=4, force=False) == []
    os.remove(ref[1].output_path)
    again = pipeline.process_files(clf_tiny, files, output_dir=str(single), min_confidence=0.05, batch_size=4, force=False)
    assert [r.output_path for r in again] == [ref[1].output_path]


def _write_wav(path, x, rate, fmt):
    """x: float [frames] or [frames, channels] in [-1, 1); fmt: 's16' | 's24' | 's32' | 'f32' (canonical 44-byte header)."""
    import struct
    x = np.asarray(x, np.float64)
    ch = 1 if x.ndim == 1 else x.shape[1]
    if fmt == "f32":
        data, tag, bits = x.astype("<f4").tobytes(), 3, 32
    elif fmt == "s32":
        data, tag, bits = np.clip(np.round(x * 2147483647.0), -2147483648, 2147483647).astype("<i4").tobytes(), 1, 32
    elif fmt == "s24":
        v = np.clip(np.round(x * 8388607.0), -8388608, 8388607).astype("<i4").reshape(-1)
        data, tag, bits = v.view(np.uint8).reshape(-1, 4)[:, :3].tobytes(), 1, 24
    else:
        data, tag, bits = np.clip(np.round(x * 32767.0), -32768, 32767).astype("<i2").tobytes(), 1, 16
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, tag, ch, rate, rate * ch * bits // 8, ch * bits // 8, bits)
    with open(path, "wb") as f:
        f.write(hdr + b"data" + struct.pack("<I", len(data)) + data)


@pytest.mark.parametrize("low_latency", [False, True])
def test_contexts_on_one_classifier_run_concurrently_with_the_sequential_bits(model_dir, low_latency):
    """One classifier, one batch context per thread (how bhh_process_files keeps three files in flight, host_pipeline.cpp; the
    reference shares its session between the pipeline's threads the same way, src/pipeline/processor.rs): six threads, each with its
    own context and its own stream of calls -- forwards of different sizes, the source-rate route through the shared resampler plan
    cache, PCM16 streams -- started together.  Every result must be bit for bit what the same call gives alone: nothing a context
    touches may be shared unguarded (lazily built plans, the low-latency regime's partial-sum scratch, the launch-once attributes)."""
    import threading
    from birda_amd import synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = model_dir["birdnet_v24_tiny"]
    clf = BirdClassifier(path, labels, top_k=5, min_confidence=0.0, low_latency=low_latency)
    S, rate = m.sample_count, m.sample_rate
    sizes = [1, 3, 17, 40, 5, 33]
    inputs = [synth.synth_segments(n, S, rate, start=100 * k) for k, n in enumerate(sizes)]
    src_rate = 44100
    n_src = int(np.ceil(S * src_rate / rate))
    pcm = [np.round(synth.synth_segments(3, n_src, src_rate, start=900 + k).reshape(-1)[: int(2.5 * n_src)] * 32767.0).astype(np.int16) for k in range(len(sizes))]

    def work(k, ctx, out):
        got = []
        for rep in range(6):
            got.append(clf.predict_logits(ctx, inputs[k]).copy())
            r = clf.predict_batch_with_context(ctx, [inputs[k][i] for i in range(min(sizes[k], 4))])
            got.append(np.array([[p.confidence for p in res.predictions] for res in r], np.float32))
            r2, starts = clf.predict_pcm16(ctx, pcm[k], src_rate)          # (the source-rate route: resampler plan cache, PCM staging)
            got.append(np.array([[p.confidence for p in res.predictions] for res in r2], np.float32))
            got.append(np.array(starts, np.float64))
        out[k] = got

    ctxs = [clf.create_batch_context(max(n, 4)) for n in sizes]
    alone, together = {}, {}
    for k in range(len(sizes)):
        work(k, ctxs[k], alone)
    threads = [threading.Thread(target=work, args=(k, ctxs[k], together)) for k in range(len(sizes))]
    for t in threads: t.start()
    for t in threads: t.join()
    for k in range(len(sizes)):
        assert len(alone[k]) == len(together[k]) and len(alone[k]) >= 24
        for a, b in zip(alone[k], together[k]):
            assert a.shape == b.shape and (a == b).all(), (low_latency, k, float(np.abs(a - b).max()))
    for c in ctxs: c.close()
    # context-less calls share the classifier's internal context behind its mutex: four threads, the sequential bits
    segs = [inputs[3][i] for i in range(6)]
    want = np.array([[p.confidence for p in r.predictions] for r in clf.predict_batch(segs)], np.float32)
    outs = {}

    def plain(k):
        outs[k] = [np.array([[p.confidence for p in r.predictions] for r in clf.predict_batch(segs)], np.float32) for _ in range(5)]

    threads = [threading.Thread(target=plain, args=(k,)) for k in range(4)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert all((o == want).all() for k in range(4) for o in outs[k])
    clf.close()
    # ... and two classifiers built at the same moment (lazily initialised kernel attributes, the plan tables) give what one gives
    built = {}

    def build(k):
        c2 = BirdClassifier(path, labels, top_k=5, min_confidence=0.0, low_latency=low_latency)
        ctx2 = c2.create_batch_context(8)
        built[k] = c2.predict_logits(ctx2, inputs[3][:8]).copy()
        ctx2.close(); c2.close()

    threads = [threading.Thread(target=build, args=(k,)) for k in range(3)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert (built[0] == built[1]).all() and (built[0] == built[2]).all() and (built[0] == alone[3][0][:8]).all()


def test_a_recording_the_resampler_cannot_take_is_refused_at_once(clf_tiny, model_dir, tmp_path):
    """A header may name any sample rate.  47 999 Hz against the model's 48 000 shares no divisor: the block resampler's operator would
    be 36 GB, and computing it kept the process inside the guarded region until the inference watchdog KILLED it -- one bad file ending
    a directory run (round 6, tools/fuzz_wav_decoder.py through bhh_process_file; also 128 Hz and 1.5 GHz headers).  Such a pair is
    refused before anything is computed (BH_ERR_UNSUPPORTED, both front ends, well under a second), and a failed device allocation no
    longer leaves its error code for the next, innocent call to find (api.hip fail())."""
    import time
    from birda_amd import pipeline, synth
    from birda_amd._lib import BirdaHipError
    _, _, m, _ = model_dir["birdnet_v24_tiny"]
    x = synth.synth_segments(3, m.sample_count, m.sample_rate, start=77).reshape(-1)
    good = str(tmp_path / "good.wav")
    _write_wav(good, x, m.sample_rate, "s16")
    out = tmp_path / "out"; out.mkdir()
    for rate in (47999, 128, 1_574_851_774):
        bad = str(tmp_path / f"bad_{rate}.wav")
        _write_wav(bad, x[: m.sample_count], rate, "s16")
        for fe in ("device", "host"):
            t = time.perf_counter()
            with pytest.raises(Exception) as ei:
                pipeline.process_file(clf_tiny, bad, str(out), min_confidence=0.05, front_end=fe)
            assert time.perf_counter() - t < 5.0, (rate, fe)
            assert getattr(ei.value, "code", None) == -6 and "resampler" in str(ei.value), (rate, fe, str(ei.value))
            r = pipeline.process_file(clf_tiny, good, str(out), min_confidence=0.05, front_end=fe)     # the next file is not harmed
            assert r.segments == 3
    # a rate of 0 is a bad argument, not a division by zero (tools/fuzz_args.py: SIGFPE in bh_resample)
    for fr, to in ((0, 48000), (48000, 0), (0, 0)):
        with pytest.raises(BirdaHipError) as ei:
            clf_tiny.resample(x[:1000], fr, to)
        assert ei.value.code == -1, str(ei.value)
    with pytest.raises(BirdaHipError) as ei:
        clf_tiny.predict_pcm16(clf_tiny.create_batch_context(4), np.zeros(3 * m.sample_count, np.int16), 0)
    assert ei.value.code == -1, str(ei.value)
    # a device allocation that fails (a context of ten million segments) is reported, and the next forward does not inherit its code
    with pytest.raises(BirdaHipError) as ei:
        clf_tiny.create_batch_context(10_000_000)
    assert ei.value.code in (-4, -1, -7), str(ei.value)
    r = pipeline.process_file(clf_tiny, good, str(out), min_confidence=0.05, front_end="device")
    assert r.segments == 3


@pytest.mark.parametrize("poison", [float("nan"), float("inf")])
def test_a_non_finite_sample_costs_its_own_segment_and_nothing_else(clf_tiny, model_dir, tmp_path, poison):
    """A float32 WAV may hold NaN or inf.  The reference hands the samples to its runtime, whose logits for that segment are NaN and
    pass no confidence threshold (processor.rs:375); here the segment that holds the sample yields no detection, every other segment
    of the file is what the clean file gives, on both front ends, and the call does not fail (the non-finite counter of the f16
    modes is about finite samples whose activations overflow: tests above)."""
    from birda_amd import pipeline, synth
    _, _, m, _ = model_dir["birdnet_v24_tiny"]
    S = m.sample_count
    x = synth.synth_segments(4, S, m.sample_rate, start=3).reshape(-1).astype(np.float32)
    y = x.copy()
    y[S + 5000] = poison                                   # inside the second segment
    clean, bad = str(tmp_path / "clean.wav"), str(tmp_path / "bad.wav")
    _write_wav(clean, x.astype(np.float64), m.sample_rate, "f32")
    import struct
    data = y.astype("<f4").tobytes()                       # (_write_wav goes through float64 arithmetic: written raw here)
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 3, 1, m.sample_rate, m.sample_rate * 4, 4, 32)
    open(bad, "wb").write(hdr + b"data" + struct.pack("<I", len(data)) + data)
    for fe in ("device", "host"):
        rows = {}
        for name, p in (("clean", clean), ("bad", bad)):
            out = tmp_path / f"{fe}_{name}"
            out.mkdir()
            r = pipeline.process_file(clf_tiny, p, str(out), min_confidence=0.05, front_end=fe)
            assert r.segments == 4
            rows[name] = [ln.split(",")[:5] for ln in open(pipeline.output_path_for(p, str(out), "csv"), encoding="utf-8-sig").read().splitlines()[1:]]
        assert any(rw[0] == "3.0" for rw in rows["clean"])
        assert not any(rw[0] == "3.0" for rw in rows["bad"]), fe
        assert [rw for rw in rows["clean"] if rw[0] != "3.0"] == rows["bad"], fe


@pytest.mark.parametrize("fmt", ["s24", "s32", "f32"])
def test_device_front_end_takes_every_wav_sample_format(clf_tiny, model_dir, tmp_path, fmt):
    """24-bit, 32-bit and float32 WAV streams are uploaded in the file's own layout and scaled on the device exactly as the host
    decoder scales them (decode.rs:353-411; 24-bit widened << 8 as symphonia does): the device front end writes the same bytes
    as the host front end, mono and stereo, at the model rate and through the resampler, also when packed with other files."""
    from birda_amd import pipeline, synth
    _, _, m, _ = model_dir["birdnet_v24_tiny"]
    S = m.sample_count
    files = []
    for k, (rate, ch, nseg) in enumerate([(48000, 1, 3), (48000, 2, 2), (44100, 1, 2)]):
        n = nseg * S * rate // 48000 + 1234
        x = synth.synth_segments(n // S + 2, S, rate, start=31 * k).reshape(-1)[:n]
        if ch == 2:
            x = np.stack([x, 0.25 * x[::-1]], 1)
        f = str(tmp_path / f"{fmt}_{k}.wav")
        _write_wav(f, x, rate, fmt)
        files.append(f)
    outs = {}
    for fe in ("host", "device"):
        d = tmp_path / fe; d.mkdir()
        for f in files:
            r = pipeline.process_file(clf_tiny, f, str(d), min_confidence=0.05, overlap=0.5, front_end=fe)
            assert r.front_end == fe and r.segments >= 2
        outs[fe] = [open(pipeline.output_path_for(f, str(d), "csv"), "rb").read() for f in files]
    assert outs["host"] == outs["device"] and any(len(b) > 200 for b in outs["device"])
    pk = tmp_path / "packed"; pk.mkdir()
    res, status = pipeline.process_files_packed(clf_tiny, files, str(pk), min_confidence=0.05, overlap=0.5)
    assert status == [0, 0, 0]
    assert [open(pipeline.output_path_for(f, str(pk), "csv"), "rb").read() for f in files] == outs["host"]


def test_packed_short_files_match_the_per_file_pipeline(clf_tiny, model_dir, tmp_path):
    """bhh_process_files packs the PCM16 streams of consecutive short recordings into one upload and one forward
    (bh_predict_pcm16_at) and scatters the rows back: byte for byte the CSV / JSON files bhh_process_file writes per file --
    with overlap, trailing partial segments, mono and stereo, two sample rates (packs break where rate or channel count change),
    a file too long to pack, a float32 WAV (host front end) and a file that does not exist in between."""
    from birda_amd import pipeline, synth
    _, _, m, _ = model_dir["birdnet_v24_tiny"]
    rec = tmp_path / "rec"; rec.mkdir()
    S = m.sample_count
    spec = [("a.wav", 2 * S + 777, 48000, 1), ("b.wav", S // 3, 48000, 1), ("c.wav", 5 * S, 48000, 1), ("d.wav", 3 * S + 1, 48000, 2),
            ("e.wav", S + S // 2, 48000, 2), ("f.wav", 2 * S, 44100, 1), ("g.wav", 4 * S - 5, 44100, 1), ("h.wav", 40 * S, 48000, 1),
            ("i.wav", 1, 48000, 1), ("j.wav", 3 * S, 48000, 1)]
    files = []
    for k, (name, n, rate, ch) in enumerate(spec):
        x = synth.synth_segments(n // S + 1, S, rate, start=20 * k).reshape(-1)[:n]
        if ch == 2:
            x = np.stack([x, 0.5 * x[::-1]], 1)
        synth.write_wav_pcm16(str(rec / name), x, rate, channels=ch)
        files.append(str(rec / name))
    f32 = rec / "k_float.wav"                                            # a float32 WAV: the packer leaves it to bhh_process_file
    import struct
    xs = synth.synth_segments(2, S, 48000, start=500).reshape(-1).astype("<f4")
    with open(f32, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", 36 + xs.nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 3, 1, 48000, 48000 * 4, 4, 32) +
                 b"data" + struct.pack("<I", xs.nbytes) + xs.tobytes())
    files.insert(4, str(f32))
    files.insert(7, str(rec / "missing.wav"))
    kw = dict(min_confidence=0.05, overlap=1.0, formats=("csv", "json"))
    single = tmp_path / "single"; single.mkdir()
    want = {}
    for f in files:
        if os.path.exists(f):
            want[f] = pipeline.process_file(clf_tiny, f, str(single), **kw)
    packed = tmp_path / "packed"; packed.mkdir()
    got, status = pipeline.process_files_packed(clf_tiny, files, str(packed), pack_segments=32, **kw)
    assert len(got) == len(files)
    n_packed = 0
    for f, r, st in zip(files, got, status):
        if not os.path.exists(f):
            assert st != 0
            continue
        assert st == 0, (f, st)
        w = want[f]
        assert (r.segments, r.detections) == (w.segments, w.detections), f
        for fmt in ("csv", "json"):
            a = pipeline.output_path_for(f, str(single), fmt)
            b = pipeline.output_path_for(f, str(packed), fmt)
            ta, tb = open(a, "rb").read(), open(b, "rb").read()
            if fmt == "json":                                   # (the document carries its own analysis_date)
                ta, tb = [b"\n".join(l for l in t.split(b"\n") if b"analysis_date" not in l) for t in (ta, tb)]
            assert ta == tb, (f, fmt)
        n_packed += r.effective_batch > r.segments          # its forward held other files' segments too
    assert n_packed >= 5                                    # {a, b, c} and {i, j} share forwards; the others stand between breaks
    assert sum(r.detections for r in got) > 0


def test_one_failing_recording_does_not_fail_its_pack(clf_tiny, model_dir, tmp_path):
    """a pack-level failure used to be copied to every file of the pack.  The reference isolates failures per
    file (lib.rs:1003-1100 counts files_failed and goes on): the pack is re-run file by file, the offending file alone reports
    the error and the others get exactly the outputs the per-file pipeline writes.
    The failing recording: a float32 WAV whose samples are FINITE but span more than the f32 range (+-3e38): max - min overflows
    in the normalisation, its spectrogram and logits are NaN although no sample was, and that is BH_ERR_NONFINITE (-8) for the
    forward that holds it -- alone through bhh_process_file, and for the whole shared forward when it is packed with others."""
    import struct
    from birda_amd import pipeline, synth
    from birda_amd._lib import BirdaHipError
    _, _, m, _ = model_dir["birdnet_v24_tiny"]
    S, rate = m.sample_count, m.sample_rate
    rec = tmp_path / "rec"; rec.mkdir()

    def write_f32(path, x):
        xs = np.asarray(x, "<f4")
        with open(path, "wb") as fh:
            fh.write(b"RIFF" + struct.pack("<I", 36 + xs.nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 3, 1, rate, rate * 4, 4, 32) +
                     b"data" + struct.pack("<I", xs.nbytes) + xs.tobytes())
    files = []
    for k in range(5):
        x = synth.synth_segments(2, S, rate, start=30 * k).reshape(-1)[: S + S // 2]
        if k == 2:
            x = x.copy(); x[::2] = 3.0e38; x[1::2] = -3.0e38          # finite samples, max - min = inf
        files.append(str(rec / f"r{k}.wav"))
        write_f32(files[-1], x)
    single = tmp_path / "single"; single.mkdir()
    want = {}
    for f in files:
        try:
            want[f] = pipeline.process_file(clf_tiny, f, str(single), min_confidence=0.05, front_end="device")
        except BirdaHipError as err:
            want[f] = err.code
    assert [isinstance(want[f], int) for f in files] == [False, False, True, False, False] and want[files[2]] == -8, want
    packed = tmp_path / "packed"; packed.mkdir()
    got, status = pipeline.process_files_packed(clf_tiny, files, str(packed), min_confidence=0.05, pack_segments=32)
    assert status == [0, 0, -8, 0, 0], status
    for f, r in zip(files, got):
        if f == files[2]:
            continue
        assert (r.segments, r.detections) == (want[f].segments, want[f].detections)
        a, b = pipeline.output_path_for(f, str(single), "csv"), pipeline.output_path_for(f, str(packed), "csv")
        assert open(a, "rb").read() == open(b, "rb").read(), f
    assert not os.path.exists(pipeline.output_path_for(files[2], str(packed), "csv"))
    assert sum(r.detections for r in got) > 0
    # ... and the pack really was shared before it failed: without the bad file the same four go through ONE forward
    got2, status2 = pipeline.process_files_packed(clf_tiny, [f for f in files if f != files[2]], str(packed), min_confidence=0.05, pack_segments=32)
    assert status2 == [0, 0, 0, 0] and all(r.effective_batch > r.segments for r in got2)


def test_parked_contexts_are_reused_and_trimmed(model_dir):
    """a destroyed context is parked in its classifier; a create of that size -- or of up to half that size, files of
    slightly different lengths cap their effective batch differently -- gets it back, and bh_classifier_trim releases what is
    parked (and the internal context of bh_predict / bh_predict_batch)."""
    from birda_amd import synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = model_dir["mini"]
    clf = BirdClassifier(path, labels)
    a = clf.create_batch_context(64)
    h_a, bytes_a = a._h.value, a.device_bytes()
    a.close()                                           # parked
    b = clf.create_batch_context(40)                    # 40 <= 64 <= 80: the parked one
    assert b._h.value == h_a and b.device_bytes() == bytes_a
    b.close()
    c = clf.create_batch_context(16)                    # 64 > 2 * 16: a new, small one
    assert c.device_bytes() < bytes_a
    c.close()
    clf.predict(synth.synth_segments(1, m.sample_count, m.sample_rate)[0])     # builds the internal context
    freed = clf.trim()
    assert freed >= bytes_a
    assert clf.trim() == 0
    d = clf.create_batch_context(64)                    # nothing parked any more: a fresh context, same results as ever
    segs = synth.synth_segments(3, m.sample_count, m.sample_rate, start=9)
    assert len(clf.predict_batch_with_context(d, list(segs))) == 3
    d.close(); clf.close()


def test_non_finite_samples_stay_in_their_own_rows(clf_tiny, model_dir):
    """A corrupt decode (NaN / Inf samples) must not leak into the other rows of a batch, hang a kernel or produce
    predictions from NaN logits: rows are independent (processor.rs:363-367)."""
    from birda_amd import synth
    _, _, m, _ = model_dir["birdnet_v24_tiny"]
    segs = synth.synth_segments(6, m.sample_count, m.sample_rate, start=500)
    ctx = clf_tiny.create_batch_context(6)
    clean = clf_tiny.predict_logits(ctx, segs)
    bad = segs.copy()
    bad[1, 1000] = np.nan
    bad[3, 77777] = np.inf
    bad[4, 5] = -np.inf
    got = clf_tiny.predict_logits(ctx, bad)
    for i in (0, 2, 5):
        assert np.array_equal(got[i], clean[i])
    res = clf_tiny.predict_batch_with_context(ctx, list(bad))
    ref = clf_tiny.predict_batch_with_context(ctx, list(segs))
    for i in (0, 2, 5):
        assert [(p.index, p.confidence) for p in res[i].predictions] == [(p.index, p.confidence) for p in ref[i].predictions]
    for i in (1, 3, 4):
        assert all(np.isfinite(p.confidence) for p in res[i].predictions)
    ctx.close()

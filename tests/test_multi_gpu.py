"""BASELINE configs 3 and 5 on the one GPU a test box has (N logical shards on one device), through
the C ABI's bh_multi_* entry points, and config 5's arithmetic (mixed-rate resample -> f16 MFMA model) against the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F16_LOGIT_RTOL = 3e-3     # BH_FLAG_F16 (operands rounded to f16 once): measured 9e-4 of the logit scale on the full model
RATES = (22050, 44100, 48000)


def _rows(results):
    idx = np.full((len(results), 5), -1, np.int64)
    conf = np.zeros((len(results), 5), np.float32)
    for i, r in enumerate(results):
        for k, p in enumerate(r.predictions):
            idx[i, k], conf[i, k] = p.index, p.confidence
    return idx, conf


def test_c3_ten_thousand_segments_as_eight_logical_shards(full_model):
    """Config 3: 10 000 HBM-resident segments, 8 contiguous shards (8 contexts / streams / host threads on ordinal 0),
    packed top-k rows gathered to the host: row for row what ONE un-sharded pass over the list returns."""
    import torch
    from birda_amd import sharding, synth
    from birda_amd.classifier import BirdClassifier
    from birda_amd.multi import MultiClassifier
    path, labels, m, _ = full_model
    n_total, G = 10000, 8
    uniq = synth.synth_segments(48, m.sample_count, m.sample_rate, start=500)
    host = np.stack([uniq[(7 * j) % 48] for j in range(n_total)])
    x = torch.from_numpy(host).cuda()

    clf = BirdClassifier(path, labels, top_k=5, min_confidence=0.02, precision="f16x3")
    ctx = clf.create_batch_context(1000)
    logits = torch.empty((1000, m.n_classes), device="cuda")
    ti = torch.empty((n_total, 5), dtype=torch.int32, device="cuda")
    tc = torch.empty((n_total, 5), device="cuda")
    for b0 in range(0, n_total, 1000):
        clf.forward_device(ctx, x.data_ptr() + b0 * m.sample_count * 4, 1000, logits.data_ptr(), ti.data_ptr() + b0 * 20, tc.data_ptr() + b0 * 20)
    ctx.synchronize()
    want_i, want_c = ti.cpu().numpy().astype(np.int64), tc.cpu().numpy()
    ctx.close(); clf.close()
    assert (want_i[:, 0] >= 0).mean() > 0.5                     # the threshold leaves real predictions to compare

    mc = MultiClassifier(path, labels, devices=[0] * G, top_k=5, min_confidence=0.02, precision="f16x3", max_batch=625)
    assert mc.n_shards == G and mc.shard_devices() == [0] * G
    assert mc.gather_backend().startswith("host")               # shards share a device: one RCCL rank per device only
    bounds = [sharding.shard_range(n_total, g, G) for g in range(G)]
    assert [hi - lo for lo, hi in bounds] == [1250] * G
    ptrs = [x.data_ptr() + lo * m.sample_count * 4 for lo, _ in bounds]
    for _ in range(2):                                          # twice: buffers are reused
        got_i, got_c = _rows(mc.forward_device(ptrs, [hi - lo for lo, hi in bounds]))
        assert np.array_equal(got_i, want_i)
        assert np.array_equal(got_c.view(np.int32), want_c.view(np.int32))
    # ragged shards (the list does not divide), an empty shard, and host segments in: same rows
    n2 = 1003
    res = mc.predict_batch_contig(host[:n2])
    got_i, got_c = _rows(res)
    assert np.array_equal(got_i, want_i[:n2]) and np.array_equal(got_c.view(np.int32), want_c[:n2].view(np.int32))
    counts = [130, 0, 125, 125, 125, 125, 125, 125]
    offs = np.cumsum([0] + counts)
    got_i, _ = _rows(mc.forward_device([x.data_ptr() + int(o) * m.sample_count * 4 for o in offs[:-1]], counts))
    assert np.array_equal(got_i, want_i[: offs[-1]])
    mc.close()

    # one shard per DISTINCT device: the gather goes through an RCCL communicator (of size 1 on this box) when librccl loads
    mc1 = MultiClassifier(path, labels, devices=[0], top_k=5, min_confidence=0.02, precision="f16x3", max_batch=1000)
    backend = mc1.gather_backend()
    assert backend == "rccl" or backend.startswith("host (")
    got_i, got_c = _rows(mc1.forward_device([x.data_ptr()], [2500]))
    assert np.array_equal(got_i, want_i[:2500]) and np.array_equal(got_c.view(np.int32), want_c[:2500].view(np.int32))
    print("single-device gather backend:", backend)
    mc1.close()


def _mixed_rate_segments(m, n, start=0):
    from birda_amd import pipeline, synth
    segs, rates = [], []
    for i in range(n):
        r = RATES[i % 3]
        segs.append(synth.synth_segments(1, pipeline.source_samples(m.sample_count, r, m.sample_rate), r, start=start + i)[0])
        rates.append(r)
    return segs, rates


def test_c5_mixed_rate_resample_into_the_f16_model_matches_the_oracle(full_model, oracle_lib):
    """Config 5 as written: the SURVEY 8d signal synthesised at 22.05 / 44.1 / 48 kHz round-robin -> device polyphase
    resampler (split-f16 MFMA) -> the FULL v2.4-shaped model with f16 MFMA operands; logits against
    oracle(rubato restatement -> resize -> forward), tolerance 3e-3 of the logit scale."""
    import torch
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = full_model
    n = 12
    segs, rates = _mixed_rate_segments(m, n, start=900)
    ref_in = np.zeros((n, m.sample_count), np.float32)
    for i, (s, r) in enumerate(zip(segs, rates)):
        y = s if r == m.sample_rate else oracle_lib.resample(s, r, m.sample_rate)
        ref_in[i, : min(len(y), m.sample_count)] = y[: m.sample_count]      # samples.resize(segment_samples, 0.0), processor.rs:87
    ref = oracle_lib.OracleModel(path).forward(ref_in)
    scale = max(1.0, float(np.abs(ref).max()))

    clf = BirdClassifier(path, labels, precision="f16")
    assert len(clf.fused_blocks()) == 16
    ctx = clf.create_batch_context(n)
    x48 = torch.zeros((n, m.sample_count), device="cuda")
    for r in RATES:
        ids = [i for i in range(n) if rates[i] == r]
        src = torch.from_numpy(np.stack([segs[i] for i in ids])).cuda()
        out = torch.empty((len(ids), m.sample_count), device="cuda")
        clf.resample_device(ctx, src.data_ptr(), src.shape[1], src.shape[1], r, m.sample_rate, out.data_ptr(), m.sample_count, m.sample_count, len(ids))
        ctx.synchronize()
        x48[torch.tensor(ids, device="cuda")] = out
    logits = torch.empty((n, m.n_classes), device="cuda")
    clf.forward_device(ctx, x48.data_ptr(), n, logits.data_ptr())
    ctx.synchronize()
    got = logits.cpu().numpy()
    err = float(np.abs(got - ref).max())
    print(f"C5 max|dlogit| = {err:.3e} of scale {scale:.2f} ({err / scale:.2e})")
    assert np.isfinite(got).all() and err <= F16_LOGIT_RTOL * scale
    assert (got.argmax(1) == ref.argmax(1)).all()
    ctx.close(); clf.close()


def test_c5_mixed_rate_list_through_logical_shards(full_model):
    """Config 5's sharding: a mixed-rate list cut into 4 shards by SOURCE samples, each shard resampling and
    classifying its (rate) groups; results in list order, identical to one classifier doing the same list."""
    from birda_amd import sharding
    from birda_amd.classifier import BirdClassifier
    from birda_amd.multi import MultiClassifier
    path, labels, m, _ = full_model
    n = 90
    segs, rates = _mixed_rate_segments(m, n, start=40)
    clf = BirdClassifier(path, labels, top_k=5, min_confidence=0.02, precision="f16")
    import ctypes
    from birda_amd._lib import BhResult, check
    want = []
    for r in RATES:
        ids = [i for i in range(n) if rates[i] == r]
        keep = [np.ascontiguousarray(segs[i]) for i in ids]
        ptrs = (ctypes.c_void_p * len(ids))(*[k.ctypes.data for k in keep])
        res = (BhResult * len(ids))()
        check(clf._L.bh_predict_batch_source_rate(clf._h, None, ptrs, len(ids), keep[0].size, r, res))
        want += list(zip(ids, clf._results(res, len(ids))))
    want = [r for _, r in sorted(want, key=lambda t: t[0])]
    clf.close()

    mc = MultiClassifier(path, labels, devices=[0, 0, 0, 0], top_k=5, min_confidence=0.02, precision="f16", max_batch=16)
    got, bounds = mc.predict_batch_source_rate(segs, rates)
    assert bounds == sharding.shard_ranges_weighted([len(s) for s in segs], 4)
    loads = [sum(len(s) for s in segs[bounds[g]:bounds[g + 1]]) for g in range(4)]
    assert max(loads) - min(loads) <= 2 * 144000
    wi, wc = _rows(want)
    gi, gc = _rows(got)
    assert np.array_equal(gi, wi) and np.array_equal(gc.view(np.int32), wc.view(np.int32))
    assert (wi[:, 0] >= 0).any()
    mc.close()


def test_auto_precision_through_logical_shards(tmp_path):
    """BH_FLAG_AUTO behind bh_multi_*: rows beyond the f16 range are re-run on the f32 kernels inside the shards' synchronise --
    after the packed rows were already on their way to the host -- so the gathered rows must be the repaired ones: what ONE
    auto classifier returns for the list, row for row, from the device-resident and the host entry point."""
    import torch
    from birda_amd import modelfile as mf, synth
    from birda_amd.classifier import BirdClassifier
    from birda_amd.multi import MultiClassifier
    from test_parity_gpu import _rescale_trunk
    m = synth.build_model("mini_b0")
    m2, _ = _rescale_trunk(m, [2.0 ** 14])               # most rows leave the f16 range (test_parity_gpu.py)
    path, labels = str(tmp_path / "overflow.bhm"), str(tmp_path / "labels.txt")
    mf.write_model(path, m2)
    synth.write_labels(labels, m.n_classes)
    n = 24
    segs = synth.synth_segments(n, m.sample_count, m.sample_rate, start=40)
    clf = BirdClassifier(path, labels, top_k=5, min_confidence=0.02, precision="auto")
    ctx = clf.create_batch_context(n)
    want_i, want_c = _rows(clf.predict_batch_with_context(ctx, list(segs)))
    assert clf.fallback_segments() > 0 and (want_i[:, 0] >= 0).mean() > 0.5
    ctx.close(); clf.close()
    mc = MultiClassifier(path, labels, devices=[0, 0, 0], top_k=5, min_confidence=0.02, precision="auto", max_batch=5)
    x = torch.from_numpy(segs).cuda()
    ptrs = [x.data_ptr() + lo * m.sample_count * 4 for lo in (0, 8, 16)]
    got_i, got_c = _rows(mc.forward_device(ptrs, [8, 8, 8]))
    assert np.array_equal(got_i, want_i) and np.allclose(got_c, want_c, atol=1e-6)
    got_i, got_c = _rows(mc.predict_batch_contig(segs))
    assert np.array_equal(got_i, want_i) and np.allclose(got_c, want_c, atol=1e-6)
    mc.close()

"""SURVEY 8f-4 on the GPU: the custom (bat) classifier on the backbone's embeddings, bat-mode segmentation in the per-file
pipeline, and BSG post-processing of the kept predictions -- against the numpy restatements in oracle/oracle.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bat(model_dir, tmp_path_factory):
    from birda_amd import modelfile as mf, synth
    path, labels, m, _ = model_dir["mini_b0"]
    d = tmp_path_factory.mktemp("bat")
    cm = synth.build_custom_classifier(m.embedding_dim, 30, (64,))
    cpath = str(d / "bat-eu.bhc")
    mf.write_custom_classifier(cpath, cm)
    clabels = [f"Pipistrellus sp{i}_Pipistrelle {i}" for i in range(30)]
    lpath = str(d / "bat-eu.txt")
    open(lpath, "w").write("\n".join(clabels) + "\n")
    return cm, cpath, lpath, clabels


def test_custom_classifier_on_backbone_embeddings(model_dir, oracle_lib, bat):
    """Two-stage inference (reference processor.rs:319-360): backbone embeddings -> dense stack -> sigmoid -> predictions.
    The embeddings never leave the device; logits against oracle(backbone) -> numpy dense stack."""
    from birda_amd import synth
    from birda_amd.classifier import BirdClassifier, CustomClassifier
    cm, cpath, lpath, clabels = bat
    path, labels, m, _ = model_dir["mini_b0"]
    segs = synth.synth_segments(7, m.sample_count, m.sample_rate, start=3)
    ref_logits, ref_emb = oracle_lib.OracleModel(path).forward(segs, want_embeddings=True)
    want = oracle_lib.custom_classifier_forward(cm, ref_emb)
    for prec in ("f32", "f16x3"):
        clf = BirdClassifier(path, labels, precision=prec)
        cc = CustomClassifier(cpath, lpath)
        assert cc.num_classes() == 30 and cc.input_dim() == m.embedding_dim and cc.label(3) == clabels[3]
        ctx = clf.create_batch_context(4)                       # smaller than the batch: two slices
        res, got = clf.predict_batch_two_stage(ctx, cc, list(segs), want_logits=True)
        scale = max(1.0, float(np.abs(want).max()))
        assert np.abs(got - want).max() <= 5e-5 * scale, (prec, float(np.abs(got - want).max()), scale)
        conf = 1.0 / (1.0 + np.exp(-want.astype(np.float64)))
        for i, r in enumerate(res):                             # every class, by confidence (top_k 0 = all classes)
            assert len(r.predictions) == 30
            order = np.argsort(-conf[i], kind="stable")
            assert [p.index for p in r.predictions[:5]] == list(order[:5])
            assert np.allclose([p.confidence for p in r.predictions], conf[i][order], rtol=1e-4, atol=1e-6)
            assert r.predictions[0].species == clabels[order[0]]
        # the host-embedding entry point (CustomClassifier::predict_batch) gives the same predictions
        _, emb = clf.predict_logits(ctx, segs, want_embeddings=True)
        res2 = cc.predict_batch(emb)
        for a, b in zip(res, res2):
            assert [p.index for p in a.predictions] == [p.index for p in b.predictions]
        # a backbone whose embedding width differs is refused
        ctx.close(); cc.close(); clf.close()
    from birda_amd._lib import BirdaHipError
    path_t, labels_t, mt, _ = model_dir["mini"]
    clf = BirdClassifier(path_t, labels_t)
    cc = CustomClassifier(cpath, lpath)
    ctx = clf.create_batch_context(2)
    with pytest.raises(BirdaHipError) as e:
        clf.predict_batch_two_stage(ctx, cc, list(synth.synth_segments(2, mt.sample_count, mt.sample_rate)))
    assert "requires" in str(e.value) and "embeddings" in str(e.value)
    ctx.close(); cc.close(); clf.close()


def test_bat_mode_file_pipeline(full_model, bat, tmp_path):
    """process_file in bat mode (processor.rs:464-475, 502-508): a 256 kHz recording is NOT resampled, cut into 144 000-sample
    segments (0.5625 s) overlapping by a quarter, and the custom classifier's labels come out."""
    from birda_amd import modelfile as mf, pipeline, synth
    from birda_amd.classifier import BirdClassifier, CustomClassifier
    path, labels, m, _ = full_model
    cm = synth.build_custom_classifier(m.embedding_dim, 12, (), seed=5)
    cpath = str(tmp_path / "bat.bhc"); mf.write_custom_classifier(cpath, cm)
    clabels = [f"Myotis sp{i}_Mouse-eared bat {i}" for i in range(12)]
    lpath = str(tmp_path / "bat.txt"); open(lpath, "w").write("\n".join(clabels) + "\n")
    rate, n = 256000, int(256000 * 2.0)                         # 2 s at 256 kHz
    t = np.arange(n) / rate
    x = np.clip(0.2 * np.random.default_rng(4).standard_normal(n) + 0.4 * np.sin(2 * np.pi * 45000 * t), -1, 1)
    wav = str(tmp_path / "bats.wav")
    synth.write_wav_pcm16(wav, x, rate)
    clf = BirdClassifier(path, labels, precision="f16x3")
    cc = CustomClassifier(cpath, lpath, top_k=3)
    res = pipeline.process_file(clf, wav, str(tmp_path), min_confidence=0.0, custom_classifier=cc, batch_size=4)
    starts = clf.segment_starts(n, 144000, 36000)               # next_segment over the stream, source-rate samples
    assert res.front_end == "host" and res.segments == len(starts) == 6   # four full windows, the partial one, and the overlap remainder (decode.rs:186-196)
    rows = open(res.output_path, encoding="utf-8-sig").read().splitlines()[1:]
    assert len(rows) == 3 * len(starts)                         # top_k 3 per segment, min_confidence 0
    seg_dur = np.float32(144000) / np.float32(rate)
    for i, s0 in enumerate(starts):
        st = np.float32(s0) / np.float32(rate)
        for row in rows[3 * i: 3 * i + 3]:
            f = row.split(",")
            assert f[0] == f"{st:.1f}" and f[1] == f"{np.float32(st + seg_dur):.1f}" and f[2].startswith("Myotis sp")
    # the same segments through the two-stage entry point directly
    dec = pipeline.StreamingDecoder(wav)
    segs = []
    while True:
        nx = dec.next_segment(144000, 36000)
        if nx is None:
            break
        segs.append(nx[0].copy())
    dec.close()
    ctx = clf.create_batch_context(8)
    direct = clf.predict_batch_two_stage(ctx, cc, segs)
    got = [(r.split(",")[2] + "_" + r.split(",")[3], float(r.split(",")[4])) for r in rows]
    want = [(p.species, p.confidence) for r in direct for p in r.predictions]
    assert [g[0] for g in got] == [w[0] for w in want]
    assert np.allclose([g[1] for g in got], [w[1] for w in want], atol=5.1e-5)
    ctx.close(); cc.close(); clf.close()


def test_bsg_postprocessing_on_the_kept_predictions(model_dir, oracle_lib):
    """BSG calibration (+ SDM prior) runs in the top-k kernel's tail (classifier.rs:508-545): compare with the numpy
    restatement applied to the un-processed predictions of the same classifier."""
    from birda_amd import synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, _ = model_dir["birdnet_v24_tiny"]
    rng = np.random.default_rng(9)
    segs = synth.synth_segments(9, m.sample_count, m.sample_rate, start=60)
    clf = BirdClassifier(path, labels, top_k=5, min_confidence=0.0)
    ctx = clf.create_batch_context(9)
    plain = clf.predict_batch_with_context(ctx, list(segs))
    a = rng.normal(0.0, 1.0, m.n_classes).astype(np.float32)
    b = rng.uniform(0.5, 2.0, m.n_classes).astype(np.float32)
    prior = rng.uniform(0.0, 1.0, m.n_classes).astype(np.float32)
    for pr in (None, prior):
        clf.set_bsg(a, b, pr)
        got = clf.predict_batch_with_context(ctx, list(segs))
        changed = False
        for r0, r1 in zip(plain, got):
            idx, conf = oracle_lib.bsg_postprocess(np.array([p.index for p in r0.predictions]), [p.confidence for p in r0.predictions], a, b, pr)
            assert [p.index for p in r1.predictions] == idx
            assert np.allclose([p.confidence for p in r1.predictions], conf, rtol=2e-5, atol=1e-7)
            changed |= idx != [p.index for p in r0.predictions]
        assert changed                                           # the calibration really re-orders some segment
    clf.clear_bsg()
    again = clf.predict_batch_with_context(ctx, list(segs))
    assert [[p.index for p in r.predictions] for r in again] == [[p.index for p in r.predictions] for r in plain]
    ctx.close(); clf.close()

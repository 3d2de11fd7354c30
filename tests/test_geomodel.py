"""The reference's own geomodel fixture on the device (VERDICT r2 missing #1).

`tests/golden/reference_fixtures/fixture-geomodel.onnx` + `fixture-geomodel-labels.txt` are the two DATA files the reference's
tests hold for this path (/root/reference/tests/fixtures/; Gemm(3 -> 5) + Sigmoid, 264 bytes), committed as they are.  They
are the only model artefact under /root/reference and the first .onnx file this repo's readers see that this repo did not
write.  W and B below are the values the reference publishes for that file (tests/fixtures/make_fixture_geomodel.py:20-28),
transcribed as numbers: the expected scores are sigmoid(x W + B) in float64, fp32 tolerance 1e-6.

What this pins on reference-held numbers: both ONNX readers (birda_amd/onnx_io.py and the library's own protobuf walk,
birda_amd/csrc/onnx_dense.hpp), Gemm + Sigmoid on the device GEMM, the label-count validation, and the chain
scores -> bhh_project_scores -> bh_classifier_set_range_filter -> top-k tail with the assertions of
/root/reference/tests/geomodel_range_filter.rs:90-300.  What it does not pin: the conv stack."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN

FIXTURES = os.path.join(GOLDEN, "reference_fixtures")
ONNX = os.path.join(FIXTURES, "fixture-geomodel.onnx")
LABELS = os.path.join(FIXTURES, "fixture-geomodel-labels.txt")

# make_fixture_geomodel.py:20-28 (values, not code)
W = np.array([[0.010, -0.020, 0.030, 0.001, 0.050],
              [0.005, 0.010, -0.015, 0.002, 0.020],
              [0.100, 0.050, -0.200, 0.010, 0.150]], np.float32)
B = np.array([0.5, -3.0, 0.2, -9.0, 1.0], np.float32)
# geomodel_range_filter.rs:35-39
TEST_LAT, TEST_LON, TEST_MONTH, TEST_DAY = 60.1699, 24.9384, 6, 15
SCORE_TOL = 1e-6


def fixture_labels():
    return [l.strip() for l in open(LABELS, encoding="utf-8").read().splitlines() if l.strip()]


def expected_scores(lat, lon, week):
    x = np.array([np.float32(lat), np.float32(lon), np.float32(week)], np.float64)
    return 1.0 / (1.0 + np.exp(-(x @ W.astype(np.float64) + B.astype(np.float64))))


# ---- CPU: the readers, the label validation, the date arithmetic -----------------------------------------------
def test_python_onnx_reader_reads_the_reference_fixture():
    from birda_amd import convert, modelfile as mf, onnx_io as ox
    g = ox.load(open(ONNX, "rb").read())
    assert [n.op_type for n in g.nodes] == ["Gemm", "Sigmoid"] and g.opset == 17
    assert [vi.name for vi in g.inputs] == ["input"] and g.inputs[0].shape == ["batch", 3]
    assert [vi.name for vi in g.outputs] == ["probabilities"] and g.outputs[0].shape == ["batch", 5]
    assert np.array_equal(g.initializers["W"], W) and np.array_equal(g.initializers["B"], B)     # bit for bit
    m = convert.dense_stack_from_graph(g)
    assert (m.input_dim, m.n_classes, m.output_activation, len(m.layers)) == (3, 5, mf.OUT_NONE, 1)
    assert m.layers[0].act == mf.ACT_SIGMOID and np.array_equal(m.layers[0].w, W) and np.array_equal(m.layers[0].b, B)
    assert len(fixture_labels()) == 5 == m.n_classes


def test_library_onnx_reader_validates_labels_before_it_needs_a_device(tmp_path):
    """RangeFilter::from_config validates the label count against the model's output width (range_filter.rs:13-18;
    geomodel_range_filter.rs:104-124: a 1-label set must not build against a 5-output model).  The library reads the
    .onnx itself, so the check runs -- and is testable -- without a GPU."""
    import torch
    from birda_amd import _lib
    L = _lib.load()
    h = C.c_void_p()
    one = tmp_path / "classifier_labels.txt"
    one.write_text("Parus major_Great Tit\n")
    assert L.bh_range_filter_create(ONNX.encode(), str(one).encode(), 0, 0.0, C.byref(h)) == -5          # BH_ERR_LABELS
    assert b"label count 1 does not match the model's output width 5" in L.bh_last_error()
    if not torch.cuda.is_available():
        assert L.bh_range_filter_create(ONNX.encode(), LABELS.encode(), 0, 0.0, C.byref(h)) == -3         # parsed, labels fine: no device
    assert L.bh_range_filter_create(ONNX.encode(), None, 0, 0.0, C.byref(h)) == -1
    assert L.bh_range_filter_create(ONNX.encode(), LABELS.encode(), 0, 1.5, C.byref(h)) == -1
    assert L.bh_range_filter_create(str(tmp_path / "missing.onnx").encode(), LABELS.encode(), 0, 0.0, C.byref(h)) == -2
    # not a dense stack: refused by operator name
    from birda_amd import onnx_io as ox
    g = ox.Graph(inputs=[ox.ValueInfo("x", ox.FLOAT, ["n", 3])], outputs=[ox.ValueInfo("y", ox.FLOAT, ["n", 3])])
    g.nodes = [ox.Node("Tanh", ["x"], ["y"])]
    bad = tmp_path / "tanh.onnx"
    bad.write_bytes(ox.dump(g))
    assert L.bh_range_filter_create(str(bad).encode(), LABELS.encode(), 0, 0.0, C.byref(h)) == -2
    assert b"operator 'Tanh' is not part of a dense stack" in L.bh_last_error()
    # a classifier-shaped input (4 inputs) is not a geomodel
    g = ox.Graph(inputs=[ox.ValueInfo("x", ox.FLOAT, ["n", 4])], outputs=[ox.ValueInfo("y", ox.FLOAT, ["n", 5])])
    g.initializers = {"w": np.zeros((4, 5), np.float32), "b": np.zeros(5, np.float32)}
    g.nodes = [ox.Node("Gemm", ["x", "w", "b"], ["z"]), ox.Node("Sigmoid", ["z"], ["y"])]
    four = tmp_path / "four.onnx"
    four.write_bytes(ox.dump(g))
    assert L.bh_range_filter_create(str(four).encode(), LABELS.encode(), 0, 0.0, C.byref(h)) == -6
    assert b"(latitude, longitude, week)" in L.bh_last_error()


def test_date_arithmetic_matches_the_reference_unit_tests():
    """utils/date.rs:129-177 (the reference's own expectations), plus the week convention the geomodel query uses."""
    from birda_amd import _lib
    L = _lib.load()
    for (mo, d), wk in {(1, 1): 1, (12, 31): 48, (6, 15): 22, (7, 1): 24}.items():        # test_date_to_week_*
        assert L.bhh_date_to_week(mo, d) == wk
    for wk, day in {1: 1, 24: 175, 48: 358}.items():                                        # test_week_to_start_day_*
        assert L.bhh_week_to_start_day(wk) == day
    for doy, want in {1: (1, 1), 365: (12, 31), 166: (6, 15), 400: (12, 31)}.items():       # test_day_of_year_to_date_*
        mo, d = C.c_uint32(), C.c_uint32()
        L.bhh_day_of_year_to_date(doy, C.byref(mo), C.byref(d))
        assert (mo.value, d.value) == want
    # [EXT] bh_birdnet_week (four weeks per month) is the convention birda's week -> start day -> (month, day) round trip
    # (config/range_filter.rs:120-123) inverts; its own 7.6-day date_to_week does not (include/birda_hip.h)
    back4, back76 = 0, 0
    for wk in range(1, 49):
        mo, d = C.c_uint32(), C.c_uint32()
        L.bhh_day_of_year_to_date(L.bhh_week_to_start_day(wk), C.byref(mo), C.byref(d))
        back4 += L.bh_birdnet_week(mo.value, d.value) == wk
        back76 += L.bhh_date_to_week(mo.value, d.value) == wk
    assert back4 == 48 and back76 == 10      # every `--week w` reaches the model as week w (the clamped form lost week 5)
    assert L.bh_birdnet_week(TEST_MONTH, TEST_DAY) == 23 and L.bh_birdnet_week(1, 1) == 1 and L.bh_birdnet_week(12, 31) == 48
    # days 29-31: the next month's first week, no clamp inside the month; the year's last days stay in week 48
    assert [L.bh_birdnet_week(1, d) for d in (28, 29, 30, 31)] == [4, 5, 5, 5] and L.bh_birdnet_week(2, 1) == 5
    assert L.bh_birdnet_week(11, 30) == 45 and L.bh_birdnet_week(12, 28) == 48 and L.bh_birdnet_week(12, 29) == 48


# ---- GPU: the fixture through the device GEMM, and the reference's end-to-end assertions ------------------------
@pytest.fixture(scope="module")
def fixture_filter():
    from birda_amd.classifier import RangeFilter
    rf = RangeFilter(ONNX, LABELS, threshold=0.0)
    yield rf
    rf.close()


@pytest.fixture(scope="module")
def five_class_classifier(tmp_path_factory):
    """a classifier whose label set is the geomodel's five species (plus one with 'Dog_Dog' and localized names below)"""
    from birda_amd import modelfile as mf, synth
    from birda_amd.classifier import BirdClassifier
    d = tmp_path_factory.mktemp("geo")
    out = {}
    for name, labels in (("same", fixture_labels()),
                         ("localized", ["Parus major_Talitiainen", "Cyanistes caeruleus_Sinitiainen", "Dog_Dog"]),
                         ("dog", ["Parus major_Great Tit", "Dog_Dog"])):
        m = synth.build_model("mini", n_classes=len(labels))
        p, lp = str(d / f"{name}.bhm"), str(d / f"{name}.txt")
        mf.write_model(p, m)
        open(lp, "w", encoding="utf-8").write("\n".join(labels) + "\n")
        out[name] = (BirdClassifier(p, lp, top_k=5, min_confidence=0.1), labels)
    yield out
    for c, _ in out.values():
        c.close()


def _planted(clf, labels, species, confidence):
    """one PredictionResult holding `species` at `confidence`: planted as a logit, through the real top-k kernel + filter tail"""
    logits = np.full((1, len(labels)), -20.0, np.float32)
    logits[0, labels.index(species)] = np.log(np.float64(confidence) / (1.0 - np.float64(confidence)))
    return [(labels[p.index], p.confidence) for p in clf.topk_from_logits(logits)[0].predictions]


def _project(geomodel_labels, scores, classifier_labels, threshold=0.0):
    from birda_amd import pipeline
    return pipeline.project_scores(geomodel_labels, list(zip(geomodel_labels, [float(s) for s in scores])), classifier_labels, threshold)


@pytest.mark.gpu
def test_fixture_scores_are_sigmoid_of_the_published_gemm(fixture_filter, tmp_path):
    rf = fixture_filter
    labels = fixture_labels()
    assert rf.num_species() == 5 and rf.labels() == labels
    scores, kept = rf.predict(TEST_LAT, TEST_LON, TEST_MONTH, TEST_DAY)
    assert scores.shape == (len(labels),) and kept.tolist() == [0, 1, 2, 3, 4]      # test_zero_threshold_returns_every_class
    want = expected_scores(TEST_LAT, TEST_LON, 23)
    assert np.abs(scores - want).max() <= SCORE_TOL, (scores, want)
    for week in (1.0, 22.0, 48.0):
        for lat, lon in ((TEST_LAT, TEST_LON), (-51.6, -69.2), (0.0, 0.0), (89.9, -179.9)):
            got, _ = rf.predict_week(lat, lon, week)
            assert np.abs(got - expected_scores(lat, lon, week)).max() <= SCORE_TOL
    # the BHC1 route (Python reader -> container) is the same model: bit-identical scores
    from birda_amd import convert
    from birda_amd.classifier import RangeFilter
    bhc = str(tmp_path / "geo.bhc")
    convert.convert_geomodel_file(ONNX, bhc)
    rf2 = RangeFilter(bhc, LABELS)
    assert np.array_equal(rf2.predict(TEST_LAT, TEST_LON, TEST_MONTH, TEST_DAY)[0], scores)
    rf2.close()
    # a threshold leaves low scorers out of the kept list (scores still cover every species)
    rf3 = RangeFilter(ONNX, LABELS, threshold=0.5)
    s3, kept3 = rf3.predict(TEST_LAT, TEST_LON, TEST_MONTH, TEST_DAY)
    assert np.array_equal(s3, scores) and kept3.tolist() == [i for i in range(5) if scores[i] >= 0.5] and 0 < len(kept3) < 5
    rf3.close()


@pytest.mark.gpu
def test_classifier_labels_are_rejected_as_geomodel_labels(tmp_path):
    """geomodel_range_filter.rs:104-124, with a device present"""
    from birda_amd._lib import BirdaHipError
    from birda_amd.classifier import RangeFilter
    one = tmp_path / "one.txt"
    one.write_text("Parus major_Great Tit\n")
    with pytest.raises(BirdaHipError) as e:
        RangeFilter(ONNX, str(one))
    assert e.value.code == -5


@pytest.mark.gpu
def test_a_different_location_produces_different_scores(fixture_filter):
    """geomodel_range_filter.rs:281-300"""
    helsinki, _ = fixture_filter.predict(TEST_LAT, TEST_LON, TEST_MONTH, TEST_DAY)
    patagonia, _ = fixture_filter.predict(-51.6, -69.2, TEST_MONTH, TEST_DAY)
    assert (np.abs(helsinki - patagonia) > 1e-6).any()


@pytest.mark.gpu
def test_scores_project_onto_localized_classifier_labels(fixture_filter, five_class_classifier):
    """geomodel_range_filter.rs:126-157"""
    scores, _ = fixture_filter.predict(TEST_LAT, TEST_LON, TEST_MONTH, TEST_DAY)
    _, labels = five_class_classifier["localized"]
    table, summary = _project(fixture_labels(), scores, labels)
    assert (summary.mapped, summary.unmatched) == (2, 1)
    assert np.isfinite(table[0]) and np.isfinite(table[1]) and np.isnan(table[2])      # score_of(..).is_some() x2, is_none()
    assert table[0] == scores[0] and table[1] == scores[1]


@pytest.mark.gpu
@pytest.mark.parametrize("policy,kept", [("keep", 1), ("drop", 0)])
def test_unmatched_policy_end_to_end(fixture_filter, five_class_classifier, policy, kept):
    """geomodel_range_filter.rs:159-214: 'Dog_Dog' has no geomodel entry"""
    scores, _ = fixture_filter.predict(TEST_LAT, TEST_LON, TEST_MONTH, TEST_DAY)
    clf, labels = five_class_classifier["dog"]
    table, _ = _project(fixture_labels(), scores, labels)
    try:
        clf.set_range_filter(table, 0.01, policy, False)
        got = _planted(clf, labels, "Dog_Dog", 0.7)
        assert len(got) == kept
        if kept:
            assert got[0][0] == "Dog_Dog" and abs(got[0][1] - 0.7) < 1e-6
    finally:
        clf.clear_filters()


@pytest.mark.gpu
def test_out_of_range_species_is_filtered_and_rerank_scales_end_to_end(fixture_filter, five_class_classifier):
    """geomodel_range_filter.rs:216-279: species index 3 (bias -9) scores below 0.01 at the query point and a 0.95 detection of
    it is dropped; 'Parus major_Great Tit' at 0.8 is re-ranked to 0.8 * score within 1e-6."""
    scores, _ = fixture_filter.predict(TEST_LAT, TEST_LON, TEST_MONTH, TEST_DAY)
    clf, labels = five_class_classifier["same"]
    table, summary = _project(fixture_labels(), scores, labels)
    assert summary.mapped == 5 and np.array_equal(table, scores)
    unlikely = "Turdus merula_Common Blackbird"
    assert labels.index(unlikely) == 3 and table[3] < 0.01
    try:
        clf.set_range_filter(table, 0.01, "keep", False)
        assert _planted(clf, labels, unlikely, 0.95) == []
        assert [s for s, _ in _planted(clf, labels, "Parus major_Great Tit", 0.8)] == ["Parus major_Great Tit"]
        clf.set_range_filter(table, 0.01, "keep", True)
        got = _planted(clf, labels, "Parus major_Great Tit", 0.8)
        assert len(got) == 1
        assert abs(got[0][1] - 0.8 * float(table[0])) < 1e-6
    finally:
        clf.clear_filters()


@pytest.mark.gpu
def test_deeper_dense_stacks_match_numpy_through_both_readers(tmp_path):
    """A three-layer stack written with this repo's ONNX writer (MatMul + Add, Relu, Gemm with transB / alpha / beta, Sigmoid;
    hidden width 16, 12 012-wide output like the published geomodel): the library's reader, the Python reader -> BHC1 route and
    float64 numpy agree."""
    from birda_amd import convert, onnx_io as ox
    from birda_amd.classifier import RangeFilter
    rng = np.random.default_rng(9)
    n_out = 12012
    w0, b0 = rng.standard_normal((3, 16)).astype(np.float32) * 0.05, rng.standard_normal(16).astype(np.float32)
    w1, b1 = rng.standard_normal((n_out, 16)).astype(np.float32), rng.standard_normal(n_out).astype(np.float32)
    g = ox.Graph(inputs=[ox.ValueInfo("x", ox.FLOAT, ["n", 3])], outputs=[ox.ValueInfo("y", ox.FLOAT, ["n", n_out])])
    g.initializers = {"w0": w0, "b0": b0, "w1": w1, "b1": b1}
    g.nodes = [ox.Node("MatMul", ["x", "w0"], ["h0"]), ox.Node("Add", ["h0", "b0"], ["h1"]), ox.Node("Relu", ["h1"], ["h2"]),
               ox.Node("Gemm", ["h2", "w1", "b1"], ["h3"], {"transB": 1, "alpha": 0.5, "beta": 2.0}), ox.Node("Sigmoid", ["h3"], ["y"])]
    path = str(tmp_path / "stack.onnx")
    open(path, "wb").write(ox.dump(g))
    labels = str(tmp_path / "labels.txt")
    open(labels, "w").write("\n".join(f"Genus species{i}_Name {i}" for i in range(n_out)) + "\n")
    x = np.array([np.float32(TEST_LAT), np.float32(TEST_LON), 23.0], np.float64)
    h = np.maximum(x @ w0.astype(np.float64) + b0, 0.0)
    want = 1.0 / (1.0 + np.exp(-(0.5 * (h @ w1.astype(np.float64).T) + 2.0 * b1)))
    rf = RangeFilter(path, labels, threshold=0.03)
    got, kept = rf.predict(TEST_LAT, TEST_LON, TEST_MONTH, TEST_DAY)
    assert got.shape == (n_out,) and np.abs(got - want).max() <= 2e-6
    assert kept.tolist() == np.nonzero(got >= np.float32(0.03))[0].tolist()
    bhc = str(tmp_path / "stack.bhc")
    convert.convert_geomodel_file(path, bhc)
    rf2 = RangeFilter(bhc, labels)
    assert np.array_equal(rf2.predict(TEST_LAT, TEST_LON, TEST_MONTH, TEST_DAY)[0], got)
    rf.close(); rf2.close()

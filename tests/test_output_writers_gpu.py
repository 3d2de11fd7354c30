"""Every result format behind a DEVICE run (Raven / Audacity / Kaleidoscope / JSON / Parquet and the NDJSON
reporter had CPU tests against the reference's unit-test expectations only -- tests/test_output_writers.py -- and no gpu test put a
writer other than CSV behind `bhh_process_file`).

Here a WAV goes through `bhh_process_file` on the MI355X with all six formats selected (reference write_output,
src/pipeline/processor.rs:819-873, one writer per OutputFormat: src/output/{csv,raven,audacity,kaleidoscope,json,parquet}.rs), and a
second time under a JSON-lines reporter (src/output/reporter.rs:170-420, the `detections` event of processor.rs:739-769).  What must
come out is what the ORACLE's detections give: the plain-C oracle segments, batches, classifies and thresholds the same samples
(oracle/birda_oracle.c bo_process_stream), its CSV rows are parsed back into (label, confidence, start, end), and those are handed
to the host writers one format at a time.  File for file the device run must match -- text fields exactly, confidences within the
1e-4 a 4-decimal CSV column resolves (two fp32 implementations differ by ~1e-6 of a logit).
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FORMATS = ("csv", "raven", "audacity", "kaleidoscope", "json", "parquet")
CONF_ATOL = 1.01e-4


def _oracle_detections(oracle_lib, model_path, names, mono, rate, overlap, min_conf, wav):
    om = oracle_lib.OracleModel(model_path)
    csv, st = om.process_stream(names, mono, rate, overlap, min_conf, 5, 8, True, wav)
    lines = csv.decode("utf-8-sig").splitlines()
    assert lines[0] == "Start (s),End (s),Scientific name,Common name,Confidence,File"
    by_name = {}
    for lab in names:
        sci, _, com = lab.partition("_")
        by_name[(sci, com if _ else sci)] = lab
    import csv as csvmod
    dets = []
    for row in csvmod.reader(lines[1:]):
        start, end, sci, com, conf, path = row
        dets.append((by_name[(sci, com)], float(conf), float(start), float(end), path))
    return dets, st


def _fields_close(a, b, sep):
    fa, fb = a.split(sep), b.split(sep)
    assert len(fa) == len(fb), (a, b)
    for x, y in zip(fa, fb):
        if x == y:
            continue
        assert abs(float(x) - float(y)) <= CONF_ATOL, (a, b)


def test_every_format_written_from_a_device_run_is_what_the_oracles_detections_give(model_dir, oracle_lib, tmp_path):
    from birda_amd import pipeline, synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, names = model_dir["birdnet_v24_tiny"]
    overlap, min_conf = 1.0, 0.02
    x = synth.synth_segments(9, m.sample_count, m.sample_rate, start=3).reshape(-1)[: int(25.5 * m.sample_rate)]
    wav = str(tmp_path / "rec field 01.wav")
    synth.write_wav_pcm16(wav, x, m.sample_rate)
    pcm = np.clip(np.round(x.astype(np.float64) * 32767.0), -32768, 32767).astype(np.int16)
    mono = np.zeros(pcm.size, np.float32)
    oracle_lib.lib().bo_pcm16_to_mono(pcm.ctypes.data, pcm.size, 1, mono)
    dets, st = _oracle_detections(oracle_lib, path, names, mono, m.sample_rate, overlap, min_conf, wav)
    assert len(dets) >= 8 and st.n_segments >= 12, (len(dets), st.n_segments)

    clf = BirdClassifier(path, labels, top_k=5, min_confidence=min_conf)
    out_dev, out_ref = tmp_path / "device", tmp_path / "oracle"
    out_dev.mkdir(); out_ref.mkdir()
    res = pipeline.process_file(clf, wav, str(out_dev), min_confidence=min_conf, overlap=overlap, formats=FORMATS, model_name="birdnet-v24-tiny",
                                lat=60.17, lon=24.94, week=22)
    assert res.front_end == "device" and res.segments == st.n_segments and res.detections == len(dets)
    assert res.formats_written == pipeline.format_mask(FORMATS)

    # the same six files from the oracle's detections through the host writers
    for fmt in FORMATS:
        p = pipeline.output_path_for(wav, str(out_ref), fmt)
        w = pipeline.OutputWriter(fmt, p, source_file=os.path.basename(wav), model="birdnet-v24-tiny", min_confidence=min_conf, overlap=overlap,
                                  audio_duration=res.audio_duration_secs, lat=60.17, lon=24.94, week=22)
        w.write_header()
        for d in dets:
            w.write_detection(*d)
        w.finalize()
        got_path = pipeline.output_path_for(wav, str(out_dev), fmt)
        assert os.path.basename(got_path) == os.path.basename(p) and os.path.exists(got_path), fmt
        if fmt == "parquet":
            import pyarrow.parquet as pq
            tg, tw = pq.read_table(got_path), pq.read_table(p)
            assert tg.schema.names == tw.schema.names and tg.num_rows == tw.num_rows == len(dets)
            for name in tg.schema.names:
                a, b = tg.column(name).to_pylist(), tw.column(name).to_pylist()
                if a != b:
                    assert all(abs(float(u) - float(v)) <= CONF_ATOL for u, v in zip(a, b)), name
        elif fmt == "json":
            dg, dw = json.load(open(got_path, encoding="utf-8")), json.load(open(p, encoding="utf-8"))
            assert list(dg) == list(dw) and dg["source_file"] == dw["source_file"] and dg["model"] == dw["model"]
            assert dg["settings"] == dw["settings"] and dg["summary"]["total_detections"] == len(dets)
            assert dg["summary"]["unique_species"] == dw["summary"]["unique_species"]
            assert abs(dg["summary"]["audio_duration_seconds"] - dw["summary"]["audio_duration_seconds"]) < 1e-3
            for a, b in zip(dg["detections"], dw["detections"]):
                assert {k: v for k, v in a.items() if k != "confidence"} == {k: v for k, v in b.items() if k != "confidence"}
                assert abs(a["confidence"] - b["confidence"]) <= CONF_ATOL
        else:
            g, w_ = open(got_path, "rb").read().decode("utf-8"), open(p, "rb").read().decode("utf-8")
            gl, wl = g.split("\n"), w_.split("\n")
            assert len(gl) == len(wl) and gl[0] == wl[0], fmt
            sep = "\t" if fmt in ("raven", "audacity") else ","
            for a, b in zip(gl[1:], wl[1:]):
                if a != b:
                    _fields_close(a, b, sep)

    # ... and under the JSON-lines reporter: the `detections` event carries the same list (no files: the reporter owns the output)
    nd = str(tmp_path / "events.ndjson")
    rep = pipeline.ProgressReporter("ndjson", nd)
    only = tmp_path / "reporter_only"
    only.mkdir()
    r2 = pipeline.process_file(clf, wav, str(only), min_confidence=min_conf, overlap=overlap, formats=FORMATS, reporter=rep)
    rep.close()
    assert r2.detections == len(dets) and r2.formats_written == 0 and not os.listdir(only)
    events = [json.loads(l) for l in open(nd, encoding="utf-8").read().splitlines() if l.strip()]
    ev = [e for e in events if e.get("event") == "detections"]
    assert len(ev) == 1, [e.get("event") for e in events]
    payload = ev[0].get("payload", ev[0])
    listed = payload["detections"]
    assert payload["file"] == wav and len(listed) == len(dets)
    for a, (lab, conf, start, end, _) in zip(listed, dets):
        sci, _, com = lab.partition("_")
        assert a["scientific_name"] == sci and a["common_name"] == (com or sci)
        assert abs(a["start_time"] - start) < 1e-6 and abs(a["end_time"] - end) < 1e-6 and abs(a["confidence"] - conf) <= CONF_ATOL
    clf.close()


@pytest.mark.parametrize("rate", [96000, 192000, 384000])
def test_high_rate_recordings_give_the_oracles_detections(rate, model_dir, oracle_lib, tmp_path):
    """96 / 192 / 384 kHz recordings (recorders that sample far above the model's 48 kHz) through `bhh_process_file`, device and host
    front end: the detections are the oracle's, whose resampler restates rubato's block FFT (reference src/audio/resample.rs:10-91).
    Round 6: decimation by more than 1.5 takes the block form of the device resampler (resample.hip) -- 96 kHz had come out 1e-4 from
    the oracle on the polyphase form (rubato's block is not shift-invariant there), 192 kHz and beyond were refused."""
    from birda_amd import pipeline, synth
    from birda_amd.classifier import BirdClassifier
    path, labels, m, names = model_dir["birdnet_v24_tiny"]
    seconds = 7.3
    n = int(seconds * rate)
    t = np.arange(n) / rate
    rng = np.random.default_rng(rate)
    x = np.clip(0.25 * np.sin(2 * np.pi * 2310.0 * t) + 0.2 * np.sin(2 * np.pi * 7100.0 * t * (1 + 0.02 * t)) + 0.05 * rng.standard_normal(n), -1, 1)
    wav = str(tmp_path / f"rec_{rate}.wav")
    synth.write_wav_pcm16(wav, x.astype(np.float32), rate)
    pcm = np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)
    mono = np.zeros(pcm.size, np.float32)
    oracle_lib.lib().bo_pcm16_to_mono(pcm.ctypes.data, pcm.size, 1, mono)
    dets, st = _oracle_detections(oracle_lib, path, names, mono, rate, 0.0, 0.02, wav)
    assert st.n_segments == 3 and len(dets) >= 3
    clf = BirdClassifier(path, labels, top_k=5, min_confidence=0.02)
    for fe in ("device", "host"):
        out = tmp_path / fe
        out.mkdir()
        res = pipeline.process_file(clf, wav, str(out), min_confidence=0.02, overlap=0.0, front_end=fe)
        assert res.segments == st.n_segments and res.detections == len(dets), (fe, res.segments, res.detections, len(dets))
        import csv as csvmod
        rows = list(csvmod.reader(open(pipeline.output_path_for(wav, str(out), "csv"), encoding="utf-8-sig").read().splitlines()))[1:]
        assert len(rows) == len(dets)
        for row, (lab, conf, start, end, _) in zip(rows, dets):
            sci, _, com = lab.partition("_")
            assert row[2] == sci and abs(float(row[4]) - conf) <= CONF_ATOL and abs(float(row[0]) - start) < 1e-3, (fe, row, lab, conf)
    clf.close()

"""Result writers, output naming, float formatting and the NDJSON / JSON progress reporter of the host pipeline
(SURVEY 8f-3), pinned on the reference's own unit-test expectations (tests/golden/reference_unit_cases.json, each
case cites the reference test) and on independent Python restatements of the Rust formatters.  CPU only."""
import json
import math
import os
import struct
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def cases():
    with open(os.path.join(GOLDEN, "reference_unit_cases.json"), encoding="utf-8") as f:
        return json.load(f)


def _write(fmt, path, dets, **kw):
    from birda_amd import pipeline
    w = pipeline.OutputWriter(fmt, path, **kw)
    w.write_header()
    for d in dets:
        w.write_detection(*d)
    w.finalize()
    return open(path, "rb").read()


def test_raven_writer_reference_cases(cases, tmp_path):
    from birda_amd import pipeline
    c = cases["raven"]
    b = c["basic"]
    text = _write("raven", str(tmp_path / "r.txt"), [(b["label"], b["conf"], b["start"], b["end"], b["path"])]).decode()
    for needle in b["contains"]:
        assert needle in text
    lines = text.split("\n")
    assert lines[0] == c["header"] and lines[1] == b["row"] and lines[2] == ""
    for sc in c["species_code"]:
        assert pipeline.species_code(sc["in"]) == sc["out"]
    # selection ids count up; spaces in the common name become underscores; multi-byte names keep whole characters
    text = _write("raven", str(tmp_path / "r2.txt"), [("A b_Ääkkönen Ölü Üç", 0.5, 3.0, 6.0, "f.wav"), ("X y_Single", 0.25, 6.0, 9.0, "f.wav")]).decode()
    rows = text.split("\n")[1:3]
    assert rows[0].split("\t")[0] == "1" and rows[1].split("\t")[0] == "2"
    assert rows[0].split("\t")[7] == "Ääkkönen_Ölü_Üç" and rows[0].split("\t")[8] == "ääküç"      # three CHARACTERS each, lower-cased
    assert rows[1].split("\t")[8] == "sing" and rows[1].split("\t")[11] == "6.0"


def test_audacity_writer_reference_cases(cases, tmp_path):
    c = cases["audacity"]
    b = c["basic"]
    text = _write("audacity", str(tmp_path / "a.txt"), [(b["label"], b["conf"], b["start"], b["end"], b["path"])]).decode()
    for needle in b["contains"]:
        assert needle in text
    assert text == "0.0\t3.0\tHouse Sparrow\t0.8542\n"
    assert _write("audacity", str(tmp_path / "a0.txt"), []).decode() == c["no_header"]["contents"]
    # underscores inside the common name become ", " (audacity.rs:29)
    assert _write("audacity", str(tmp_path / "a1.txt"), [("S n_Night_Heron", 0.12345, 1.25, 4.25, "x")]).decode() == "1.2\t4.2\tNight, Heron\t0.1235\n"


def test_kaleidoscope_writer_reference_cases(cases, tmp_path):
    c = cases["kaleidoscope"]
    b = c["basic"]
    text = _write("kaleidoscope", str(tmp_path / "k.csv"), [(b["label"], b["conf"], b["start"], b["end"], b["path"])]).decode()
    for needle in b["contains"]:
        assert needle in text
    assert text.split("\n")[:2] == [c["header"], b["row"]]
    # Path::parent / file_name corner cases (kaleidoscope.rs:40-57): bare name, one directory, root
    rows = _write("kaleidoscope", str(tmp_path / "k2.csv"), [("a_B", 0.5, 3.0, 6.5, "audio.wav"), ("a_B", 0.5, 0.0, 3.0, "dir/audio.wav"),
                                                            ("a_B", 0.5, 0.0, 3.0, "/audio.wav"), ("a_B", 0.5, 0.0, 3.0, "/x/y/z/audio.wav")]).decode().split("\n")
    assert rows[1] == ",,audio.wav,3.0,3.5,B,0.5000"
    assert rows[2] == ",dir,audio.wav,0.0,3.0,B,0.5000"
    assert rows[3] == ",,audio.wav,0.0,3.0,B,0.5000"
    assert rows[4] == "/x/y,z,audio.wav,0.0,3.0,B,0.5000"


def test_csv_writer_columns_and_bom(tmp_path):
    raw = _write("csv", str(tmp_path / "c.csv"), [("Passer domesticus_House Sparrow", 0.8542, 0.0, 3.0, "/p/a,b.wav")], csv_bom=True,
                 csv_columns=["lat", "lon", "week", "model"])
    assert raw[:3] == b"\xef\xbb\xbf"
    lines = raw[3:].decode().split("\n")
    assert lines[0] == "Start (s),End (s),Scientific name,Common name,Confidence,File,lat,lon,week,model"   # csv.rs:41-52
    # DetectionMetadata is all None in the pipeline (types.rs:69-78): the extra cells are empty (csv.rs:68-113)
    assert lines[1] == '0.0,3.0,Passer domesticus,House Sparrow,0.8542,"/p/a,b.wav",,,,'
    raw = _write("csv", str(tmp_path / "c2.csv"), [], csv_bom=False)
    assert raw == b"Start (s),End (s),Scientific name,Common name,Confidence,File\n"


def test_json_writer_reference_cases(cases, tmp_path):
    for name in ("basic", "unique"):
        c = cases["json_writer"][name]
        raw = _write("json", str(tmp_path / f"{name}.json"), [tuple(d) for d in c["detections"]], source_file=c["source_file"],
                     model=c["model"], min_confidence=c["min_confidence"], overlap=c["overlap"], audio_duration=c["audio_duration"],
                     lat=c.get("lat"), lon=c.get("lon"), week=c.get("week")).decode()
        doc = json.loads(raw)
        e = c["expect"]
        assert list(doc) == ["source_file", "analysis_date", "model", "settings", "detections", "summary"]   # struct field order
        assert doc["summary"]["total_detections"] == e["total_detections"] and doc["summary"]["unique_species"] == e["unique_species"]
        if name == "basic":
            assert doc["source_file"] == e["source_file"] and doc["model"] == e["model"] and len(doc["detections"]) == e["n_detections"]
            assert doc["detections"][0]["scientific_name"] == e["first_scientific_name"]
            assert abs(doc["summary"]["audio_duration_seconds"] - e["audio_duration_seconds"]) < 1e-3
            assert "lat" not in doc["settings"] and "week" not in doc["settings"]     # skip_serializing_if = Option::is_none
            # serde_json::to_writer_pretty: two-space indent, f32 through ryu (0.1f32 -> 0.1, 60.0 -> 60.0)
            assert '  "settings": {\n    "min_confidence": 0.1,\n    "overlap": 0.0\n  },' in raw
            assert '      "confidence": 0.95\n    }\n  ],' in raw and raw.endswith('    "audio_duration_seconds": 60.0\n  }\n}')
        else:
            assert doc["settings"]["lat"] == e["lat"] and doc["settings"]["lon"] == -73.0 and doc["settings"]["week"] == 24
            assert list(doc["detections"][1]) == ["start_time", "end_time", "scientific_name", "common_name", "confidence"]
        assert doc["analysis_date"].endswith("Z") and "T" in doc["analysis_date"]
    raw = _write("json", str(tmp_path / "empty.json"), [], source_file='we"ird\\\n.wav', model="m").decode()
    assert '"detections": [],' in raw and json.loads(raw)["source_file"] == 'we"ird\\\n.wav'


def _rust_display(v, f32):
    """core::fmt Display for floats: the shortest digit string that round-trips, positional (never an exponent)."""
    if f32:
        v = float(np.float32(v))
        r = np.format_float_positional(np.float32(v), unique=True, trim="-")
    else:
        r = np.format_float_positional(np.float64(v), unique=True, trim="-")
    return r


def _ryu_pretty(v, f32):
    """serde_json's float text (ryu 'pretty'): digits + exponent laid out by the position of the decimal point."""
    if v == 0:
        return "-0.0" if math.copysign(1, v) < 0 else "0.0"
    s = np.format_float_scientific(np.float32(v) if f32 else np.float64(v), unique=True, trim="-", exp_digits=1)
    mant, ex = s.split("e")
    neg = mant.startswith("-")
    digits = mant.lstrip("-").replace(".", "")
    kk = int(ex) + 1
    n = len(digits)
    k = kk - n
    lim, low = (13, -6) if f32 else (16, -5)
    if 0 <= k and kk <= lim:
        out = digits + "0" * k + ".0"
    elif 0 < kk <= lim:
        out = digits[:kk] + "." + digits[kk:]
    elif low < kk <= 0:
        out = "0." + "0" * (-kk) + digits
    else:
        out = digits[0] + ("." + digits[1:] if n > 1 else "") + "e" + str(kk - 1)
    return ("-" if neg else "") + out


def test_float_formatting_matches_rust_layouts():
    from birda_amd import _lib, pipeline
    rng = np.random.default_rng(5)
    vals = [0.0, 1.0, 0.1, 0.95, 3.0, 60.0, 45.0, -73.0, 0.8542, 1e-7, 1.5e-5, 123456.789, 1e16, 1e13, 1e12, 2.5e-6, 9.999999e-6, 1e-5,
            16777216.0, 3.4e38, 1.17549435e-38, 0.30000001192092896, 1 / 3]
    vals += list(np.exp(rng.uniform(-30, 30, 200)) * rng.choice([-1, 1], 200))
    for v in vals:
        assert pipeline.format_float(_lib.FLOAT_DISPLAY_F32, v) == _rust_display(v, True), v
        assert pipeline.format_float(_lib.FLOAT_DISPLAY_F64, v) == _rust_display(v, False), v
        assert pipeline.format_float(_lib.FLOAT_JSON_F32, v) == _ryu_pretty(float(np.float32(v)), True), v
        assert pipeline.format_float(_lib.FLOAT_JSON_F64, v) == _ryu_pretty(v, False), v
        for kind, f32 in ((_lib.FLOAT_JSON_F32, True), (_lib.FLOAT_JSON_F64, False)):      # what serde_json reads back is the same number
            back = json.loads(pipeline.format_float(kind, v))
            assert (np.float32(back) == np.float32(v)) if f32 else (back == v)
    # spot values with known Rust / serde_json text
    assert pipeline.format_float(_lib.FLOAT_JSON_F32, 0.1) == "0.1" and pipeline.format_float(_lib.FLOAT_JSON_F64, 0.1) == "0.1"
    assert pipeline.format_float(_lib.FLOAT_JSON_F32, 1e-7) == "1e-7" and pipeline.format_float(_lib.FLOAT_JSON_F64, 1e16) == "1e16"
    assert pipeline.format_float(_lib.FLOAT_DISPLAY_F64, 45.0) == "45" and pipeline.format_float(_lib.FLOAT_DISPLAY_F32, 1e-7) == "0.0000001"
    assert pipeline.format_float(_lib.FLOAT_JSON_F32, float("nan")) == "null" and pipeline.format_float(_lib.FLOAT_DISPLAY_F32, float("inf")) == "inf"


def test_output_path_for_reference_cases(cases):
    from birda_amd import pipeline
    for c in cases["output_path"]:
        assert pipeline.output_path_for(c["input"], c["dir"], c["format"]) == c["path"], c["src"]
    assert pipeline.output_path_for("audio.wav", None, "csv") == "audio.BirdNET.results.csv"     # parent() == Some("") joins to the bare name
    assert pipeline.output_path_for(".hidden", "/o", "csv") == "/o/.hidden.BirdNET.results.csv"   # file_stem of a dot file is the whole name
    with pytest.raises(ValueError):
        pipeline.format_mask(["xml"])
    assert pipeline.format_mask(["csv", "table", "JSON"]) == 1 | 2 | 16                           # "table" is an alias of raven (types.rs:360)


def test_parquet_writer_reads_back_with_pyarrow(tmp_path):
    pq = pytest.importorskip("pyarrow.parquet")
    import pyarrow as pa
    dets = [("Passer domesticus_House Sparrow", 0.8542, 0.0, 3.0, "/path/to/audio.wav"),
            ("Turdus merula_Eurasian Blackbird", 0.25, 3.0, 6.0, "/path/to/audio.wav"),
            ("Nonevent", 0.125, 6.0, 9.0, "rel.wav")] * 700                                      # past the reference's 1000-row batches
    path = str(tmp_path / "d.parquet")
    _write("parquet", path, dets, csv_columns=["lat", "week", "model", "bogus", "min_conf"])
    t = pq.read_table(path)
    # build_schema (parquet.rs:141-171): six required core columns, nullable metadata columns, unknown names skipped
    assert t.schema.names == ["start_s", "end_s", "scientific_name", "common_name", "confidence", "file", "lat", "week", "model", "min_conf"]
    want = {"start_s": pa.float32(), "end_s": pa.float32(), "scientific_name": pa.string(), "common_name": pa.string(),
            "confidence": pa.float32(), "file": pa.string(), "lat": pa.float64(), "week": pa.uint8(), "model": pa.string(), "min_conf": pa.float32()}
    for f in t.schema:
        assert f.type == want[f.name], f
        assert f.nullable == (f.name in ("lat", "week", "model", "min_conf")), f
    assert t.num_rows == len(dets)
    cols = t.to_pydict()
    assert cols["scientific_name"][:3] == ["Passer domesticus", "Turdus merula", "Nonevent"] and cols["common_name"][2] == "Nonevent"
    assert cols["file"][:3] == ["audio.wav", "audio.wav", "rel.wav"]                               # the file NAME (parquet.rs:219-231)
    assert np.array_equal(np.asarray(cols["confidence"], np.float32), np.asarray([d[1] for d in dets], np.float32))
    assert np.array_equal(np.asarray(cols["end_s"], np.float32), np.asarray([d[3] for d in dets], np.float32))
    assert all(v is None for v in cols["lat"]) and all(v is None for v in cols["week"]) and all(v is None for v in cols["model"])
    md = pq.ParquetFile(path).metadata
    assert md.format_version == "2.6" or md.format_version.startswith("2")                          # WriterVersion::PARQUET_2_0
    assert md.num_row_groups == 1 and md.row_group(0).column(0).compression == "SNAPPY"             # parquet.rs:44-47
    # no detections: a valid file with the schema and no rows
    path0 = str(tmp_path / "d0.parquet")
    _write("parquet", path0, [])
    t0 = pq.read_table(path0)
    assert t0.num_rows == 0 and t0.schema.names[:6] == ["start_s", "end_s", "scientific_name", "common_name", "confidence", "file"]


def test_reporter_reference_cases_and_event_shapes(cases, tmp_path):
    from birda_amd import pipeline
    out = str(tmp_path / "events.ndjson")
    r = pipeline.ProgressReporter("ndjson", out)
    c = cases["reporter"]["ndjson"]
    r.pipeline_started(c["total_files"], c["model"], c["min_confidence"], c["requested"], c["actual"])
    r.file_started("a.wav", 0, 10, 30.0)
    r.detections("test.wav", [("Parus major_Great Tit", 0.95, 0.0, 3.0)])
    r.file_completed_success("a.wav", 3, 120)
    r.file_completed_failure("b.wav", "audio_open", "cannot open")
    r.file_skipped("c.wav")
    r.file_skipped("d.wav", locked=True)
    r.error("inference_failed", True, "boom", "reduce the batch size")
    r.pipeline_completed(1, 1, 2, 3, 10, 1500, 20.0)
    r.close()
    text = open(out, encoding="utf-8").read()
    for needle in c["contains"] + cases["reporter"]["detections"]["contains"]:
        assert needle in text
    ev = [json.loads(l) for l in text.splitlines()]
    assert [e["event"] for e in ev] == ["pipeline_started", "file_started", "detections", "file_completed", "file_completed",
                                        "file_completed", "file_completed", "error", "pipeline_completed"]
    for e in ev:
        assert list(e) == ["spec_version", "timestamp", "event", "payload"] and e["spec_version"] == "1.1"   # json_envelope.rs:10-25
    assert ev[0]["payload"] == {"total_files": 5, "model": "test-model", "min_confidence": 0.1,
                                "execution_provider": {"requested": "cpu", "actual": "CPU"}}        # None fields are skipped
    assert ev[1]["payload"] == {"file": "a.wav", "index": 0, "estimated_segments": 10, "duration_seconds": 30.0}
    assert ev[2]["payload"]["detections"][0] == {"species": "Parus major_Great Tit", "common_name": "Great Tit", "scientific_name": "Parus major",
                                                 "confidence": 0.95, "start_time": 0.0, "end_time": 3.0}
    assert ev[3]["payload"] == {"file": "a.wav", "status": "processed", "detections": 3, "duration_ms": 120}
    assert ev[4]["payload"] == {"file": "b.wav", "status": "failed", "error": {"code": "audio_open", "message": "cannot open"}}
    assert ev[5]["payload"] == {"file": "c.wav", "status": "skipped"} and ev[6]["payload"]["status"] == "locked"
    assert ev[7]["payload"] == {"code": "inference_failed", "severity": "fatal", "message": "boom", "suggestion": "reduce the batch size"}
    assert ev[8]["payload"] == {"status": "partial_success", "files_processed": 1, "files_failed": 1, "files_skipped": 2,
                                "total_detections": 3, "total_segments": 10, "duration_ms": 1500, "realtime_factor": 20.0}
    # JSON mode: nothing until pipeline_completed, then one array (reporter.rs:224-243)
    out2 = str(tmp_path / "events.json")
    r = pipeline.ProgressReporter("json", out2)
    r.pipeline_started(1, "m", 0.25, "gpu", "HIP", None, {"geomodel_version": "3.0.2", "species_in_range": 120, "total_species": 6522,
                                                          "mapped_species": 6000, "unmatched_species": 522, "unmatched_policy": "keep", "threshold": 0.03})
    assert os.path.getsize(out2) == 0
    r.pipeline_completed(1, 0, 0, 0, 0, 5, 1.0)
    r.close()
    arr = json.loads(open(out2).read())
    assert [e["event"] for e in arr] == ["pipeline_started", "pipeline_completed"] and arr[1]["payload"]["status"] == "success"
    assert arr[0]["payload"]["range_filter"]["unmatched_policy"] == "keep" and arr[0]["payload"]["execution_provider"]["actual"] == "HIP"
    # BSG models: the detections event carries BsgMetadata (json_envelope.rs:352-378; processor.rs:741-768), Option fields skipped
    out3 = str(tmp_path / "bsg.ndjson")
    r = pipeline.ProgressReporter("ndjson", out3)
    det = [("Parus major_Great Tit", 0.5, 0.0, 3.0)]
    r.detections("a.wav", det)                                                        # no BSG processor: no "bsg" key
    r.detections("a.wav", det, pipeline.bsg_metadata())                               # calibration only
    r.detections("a.wav", det, pipeline.bsg_metadata(60.25, 24.5))                    # location, no day: SDM not applied
    r.detections("a.wav", det, pipeline.bsg_metadata(60.25, 24.5, 166))               # SDM applied
    r.close()
    pl = [json.loads(l)["payload"] for l in open(out3, encoding="utf-8").read().splitlines()]
    assert "bsg" not in pl[0] and list(pl[1]) == ["file", "detections", "bsg"]
    assert pl[1]["bsg"] == {"calibration_applied": True, "sdm_applied": False}
    assert pl[2]["bsg"] == {"calibration_applied": True, "sdm_applied": False, "latitude": 60.25, "longitude": 24.5}
    assert pl[3]["bsg"] == {"calibration_applied": True, "sdm_applied": True, "latitude": 60.25, "longitude": 24.5, "day_of_year": 166}


def test_progress_throttler_reference_cases(cases, tmp_path):
    from birda_amd import pipeline
    for c in cases["throttler"]:
        r = pipeline.ProgressReporter("ndjson", str(tmp_path / "t.ndjson"))
        for pct, want in c["steps"]:
            if pct == "reset":
                r.file_started("f.wav", 0, 100)          # file_started resets the throttler (reporter.rs:296)
            else:
                assert r.file_progress("f.wav", int(pct), 100, pct) == want, (c["src"], pct)
        r.close()
    r = pipeline.ProgressReporter("ndjson", str(tmp_path / "t2.ndjson"))
    assert r.file_progress("f", 0, 0, 0.0) and not r.file_progress("f", 3, 100, 3.0)
    time.sleep(0.55)                                     # 500 ms since the last event lets a small change through
    assert r.file_progress("f", 4, 100, 4.0)
    r.close()


def test_watchdog_is_one_thread_for_any_number_of_batches():
    """One process-wide watchdog thread (the reference's thread-per-batch does not survive sub-millisecond batches), and a
    guard that is not cancelled in time still terminates the process with exit code 1 (watchdog.rs:37-50)."""
    from birda_amd import _lib
    L = _lib.load()

    def n_threads():
        return int([l for l in open("/proc/self/status") if l.startswith("Threads:")][0].split()[1])
    g = L.bhh_watchdog_start(60_000, 8)
    L.bhh_watchdog_cancel(g)
    base = n_threads()
    guards = []
    for _ in range(20000):
        guards.append(L.bhh_watchdog_start(3_600_000, 256))
        if len(guards) == 64:
            for g in guards:
                L.bhh_watchdog_cancel(g)
            guards = []
    assert n_threads() == base
    code = ("import sys; sys.path.insert(0, %r)\nfrom birda_amd import _lib\nL = _lib.load()\n"
            "a = L.bhh_watchdog_start(60000, 4); b = L.bhh_watchdog_start(300, 16); L.bhh_watchdog_cancel(a)\n"
            "import time; time.sleep(5); print('survived')\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "survived" not in p.stdout
    assert "FATAL: Inference timeout after 0s (batch size: 16)" in p.stderr and "birda -b 8 <input>" in p.stderr


def test_default_batch_size_reference_cases_and_provider_arm(cases):
    from birda_amd import _lib
    L = _lib.load()
    model = {"birdnet_v24": 0, "perch_v2": 1, "birdnet_v30": 2, "bsg_finland": 3}
    for c in cases["default_batch_size"]:
        for name, want in c["models"].items():
            assert L.bh_default_batch_size(model[name], c["provider"].encode()) == want, (c["src"], name)
    for name in model:                                   # this backend's arm; within MIN..MAX_BATCH_SIZE (constants.rs:44,55)
        assert L.bh_default_batch_size(model[name], b"HIP") == 512 == L.bh_default_batch_size(model[name], None)
    st = _lib.BhProviderStatus()
    assert L.bh_select_provider(b"CPU", -1, st) == 0 and (st.requested, st.actual, st.fallback_reason) == (b"cpu", b"CPU", b"")
    assert L.bh_select_provider(b"tensorrt", -1, st) == -1          # not this backend's arm
    import torch
    if not torch.cuda.is_available():
        assert L.bh_select_provider(b"auto", -1, st) == 0
        assert (st.requested, st.actual, st.fallback_reason, st.device) == (b"auto", b"CPU", b"No GPU providers available", -1)
        assert L.bh_select_provider(b"gpu", -1, st) == 0 and st.actual == b"CPU" and st.fallback_reason == b"No GPU providers available"
        assert L.bh_select_provider(b"hip", -1, st) == -3 and L.bh_select_provider(b"rocm", 0, st) == -3   # explicit provider unavailable: an error
        assert b"no HIP device" in L.bh_last_error()


def test_malformed_wav_headers_fail_cleanly(tmp_path):
    """An fmt chunk that claims 4 GiB, a truncated header, an empty data chunk: status codes, no exception across the ABI."""
    from birda_amd import pipeline
    from birda_amd._lib import BirdaHipError
    bad = tmp_path / "huge_fmt.wav"
    bad.write_bytes(b"RIFF" + struct.pack("<I", 36) + b"WAVEfmt " + struct.pack("<I", 0xFFFFFFF0) + b"\x01\x00\x01\x00" + b"\x00" * 12)
    with pytest.raises(BirdaHipError):
        pipeline.StreamingDecoder(str(bad))
    trunc = tmp_path / "trunc.wav"
    trunc.write_bytes(b"RIFF\x00\x00")
    with pytest.raises(BirdaHipError):
        pipeline.StreamingDecoder(str(trunc))
    fmt = struct.pack("<IHHIIHH", 16, 1, 1, 48000, 96000, 2, 16)
    empty = tmp_path / "empty.wav"
    empty.write_bytes(b"RIFF" + struct.pack("<I", 36) + b"WAVEfmt " + fmt + b"data" + struct.pack("<I", 0) + b"\x01\x02" * 100)
    d = pipeline.StreamingDecoder(str(empty))          # a zero-length data chunk is an empty stream, not "to end of file"
    assert d.duration_hint() == 0.0 and d.next_segment(1000, 0) is None
    d.close()
    ext = tmp_path / "ext.wav"                         # a 40-byte WAVE_FORMAT_EXTENSIBLE fmt chunk followed by padding bytes
    fmt_ext = struct.pack("<IHHIIHH", 48, 0xFFFE, 1, 48000, 96000, 2, 16) + struct.pack("<HHI", 22, 16, 4) + struct.pack("<H", 1) + b"\x00" * 14 + b"\x00" * 8
    ext.write_bytes(b"RIFF" + struct.pack("<I", 100) + b"WAVEfmt " + fmt_ext + b"data" + struct.pack("<I", 8) + struct.pack("<4h", 16384, -16384, 0, 32767))
    d = pipeline.StreamingDecoder(str(ext))
    seg, start = d.next_segment(4, 0)
    assert start == 0 and np.allclose(seg, [0.5, -0.5, 0.0, 32767 / 32768])
    d.close()


def test_should_process_is_the_reference_resume_rule(tmp_path):
    """should_process (pipeline/coordinator.rs:96-143) without the lock-file arm: a file is skipped only when EVERY requested
    format's output exists and force is off; an empty format list always processes (#339 in the reference's comment)."""
    from birda_amd import _lib, pipeline
    wav = str(tmp_path / "rec" / "a.wav")
    os.makedirs(os.path.dirname(wav))
    open(wav, "wb").close()
    out = str(tmp_path / "out")
    os.makedirs(out)
    assert pipeline.should_process(wav, out, ("csv", "json"))
    open(pipeline.output_path_for(wav, out, "csv"), "w").close()
    assert pipeline.should_process(wav, out, ("csv", "json"))            # json still missing
    assert not pipeline.should_process(wav, out, ("csv",))
    open(pipeline.output_path_for(wav, out, "json"), "w").close()
    assert not pipeline.should_process(wav, out, ("csv", "json"))
    assert pipeline.should_process(wav, out, ("csv", "json"), force=True)
    assert _lib.load().bhh_should_process(wav.encode(), out.encode(), 0, 0) == 1       # nothing asked for: process
    # outputs beside the input when no output directory is given (output_path_for)
    assert pipeline.should_process(wav, None, ("raven",))
    open(pipeline.output_path_for(wav, None, "raven"), "w").close()
    assert not pipeline.should_process(wav, None, ("raven",))

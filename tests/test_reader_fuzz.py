"""A model file is untrusted input (the reference hands whatever `models install` downloaded to its runtime,
src/inference/classifier.rs:269-283): whatever the bytes, the library's readers must RETURN -- an error code and a message --, never
crash or hang.  A bounded byte-level fuzz of both readers on the CPU (tools/fuzz_onnx_reader.py: the `.onnx` route with its protobuf
walk, conv-stack reader and probing front-end recovery; the flat BHM1 container through the host-side planner), child processes so
that a crash is seen as a signal.  (Round 6: 6 000 mutants of the longer runs, no crash; ADVICE r5 had found two by reading.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mutated_model_files_are_refused_or_read_never_a_crash():
    env = dict(os.environ, FUZZ_BHM="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_onnx_reader.py"), "240", "3"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-600:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("240 mutants, 0 bad batches"), r.stdout[-800:]
    # the mutants reach past the first byte: some are read, some refused as malformed, some as unsupported
    assert "'0':" in last and "'-2':" in last, last


def test_mutated_recordings_are_decoded_or_refused_never_a_crash():
    """The host WAV decoder (bhh_decoder_open / _next_segment: reference src/audio/decode.rs:54-411) on mutated files of every sample
    format: header fields overwritten with 0 / 1 / 2^31 / 2^32 - 1, truncations, insertions (tools/fuzz_wav_decoder.py; the same mutants
    through bhh_process_file on the GPU box: tests/test_parity_gpu.py::test_a_recording_the_resampler_cannot_take_is_refused_at_once)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_wav_decoder.py"), "300", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-600:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("300 mutants, 0 bad batches"), r.stdout[-800:]
    assert "'0':" in last and "'-2':" in last, last


def test_a_front_end_branch_that_points_outside_the_file_is_refused(tmp_path):
    """validate_model (birda_amd/csrc/model.hpp) held every LAYER to the blob and to its producer and nothing of the front-end's
    BRANCHES: a branch record whose mel_w_off said 4.5e18 reached the operator build, which read the blob there (SIGBUS in
    bh_classifier_create; tools/fuzz_create.py on the GPU box, round 6).  Branch records rewritten one field at a time: each is
    refused as malformed by name -- through the host-side planner, which loads and validates the container without a GPU."""
    import ctypes as C
    import struct
    sys.path.insert(0, ROOT)
    from birda_amd import _lib, modelfile as mf, synth
    L = _lib.load()
    m = synth.build_model("mini_se")
    base = str(tmp_path / "base.bhm")
    mf.write_model(base, m)
    raw = open(base, "rb").read()
    cfgs, layers = (C.c_int32 * 64)(), (C.c_int32 * 64)()
    assert L.bh_plan_fused_blocks(base.encode(), 1, cfgs, layers, 64) >= 0
    off = 256                                  # the first branch record (BRANCH_FMT "<IIIIIIfffffIQ": 64 bytes behind the header)
    rec = list(struct.unpack_from(mf.BRANCH_FMT, raw, off))
    fields = {"frame_length": 0, "frame_step": 1, "fft_length": 2, "n_bins": 3, "n_mels": 4, "n_frames": 5, "mag_scale": 8, "mel_w_off": 12}
    cases = [("mel_w_off", 4496763963580612608, "mel matrix outside blob"), ("mel_w_off", len(m.blob) - 5, "mel matrix outside blob"),
             ("n_frames", rec[5] + 40, "front-end"), ("frame_step", 0, "front-end"), ("frame_step", rec[1] * 3, "front-end"),
             ("frame_length", rec[0] * 64, "front-end"), ("n_bins", rec[3] + 1, "front-end"), ("n_mels", rec[4] + 16, "front-end"),
             ("fft_length", rec[2] * 2, "front-end"), ("mag_scale", float("nan"), "not finite")]
    for name, value, why in cases:
        r = list(rec)
        r[fields[name]] = value
        bad = str(tmp_path / f"bad_{name}.bhm")
        open(bad, "wb").write(raw[:off] + struct.pack(mf.BRANCH_FMT, *r) + raw[off + struct.calcsize(mf.BRANCH_FMT):])
        rc = L.bh_plan_fused_blocks(bad.encode(), 1, cfgs, layers, 64)
        msg = L.bh_last_error().decode()
        assert rc == -2 and why in msg, (name, value, rc, msg)


def test_text_out_functions_never_write_past_the_capacity_they_are_given():
    """The host ABI's text-out functions (include/birda_host.h: `char *out, size_t cap` -> the length needed) with labels, paths and
    floats of every kind and capacities from 0 up: 64 canary bytes behind the buffer must survive every call."""
    import ctypes as C
    import random
    import struct
    sys.path.insert(0, ROOT)
    from birda_amd import _lib
    L = _lib.load()
    rng = random.Random(3)
    alphabet = ["a", "Z", " ", "_", ",", '"', "\n", "\t", "\x01", "\u00e9", "\u9ce5", "\U0001F600", "/", ".", "\\", "-"]
    canary = b"\xA5" * 64

    def word(n=20):
        return "".join(rng.choice(alphabet) for _ in range(rng.randrange(0, n)))

    def call(fn, args, hint=512):
        for cap in (0, 1, 2, 5, 17, 64, hint):
            buf = C.create_string_buffer(cap + 64)
            buf.raw = b"\x5a" * cap + canary
            fn(*args, buf, cap)
            assert buf.raw[cap:cap + 64] == canary, (fn.__name__, args, cap)

    for _ in range(400):
        lab, path = (word() + "_" + word()).encode(), ("/d/" + word()).encode()
        call(L.bhh_csv_row, (lab, C.c_float(rng.uniform(-1e9, 1e9)), C.c_float(rng.uniform(0, 1e6)),
                             C.c_float(rng.choice([0.5, float("nan"), float("inf"), 1e-30, -0.0, 3.4e38])), path))
        call(L.bhh_csv_header, (rng.randrange(2),))
        call(L.bhh_output_path_for, (("/x/" + word(40) + rng.choice([".wav", ".WAV", "", ".flac", "."])).encode(),
                                     rng.choice([None, b"", ("/o/" + word()).encode()]), rng.choice([0, 1, 2, 4, 8, 16, 32, 64, 3, 0xFFFFFFFF])))
        call(L.bhh_species_code, (word(30).encode(),))
        v = rng.choice([0.0, -0.0, 1.0, 0.1, 1e-7, 1e21, 1e-310, float("nan"), float("inf"), -float("inf"), rng.uniform(-1e6, 1e6),
                        struct.unpack("<d", struct.pack("<Q", rng.getrandbits(64)))[0]])
        call(L.bhh_format_float, (rng.randrange(-1, 6), C.c_double(v)))

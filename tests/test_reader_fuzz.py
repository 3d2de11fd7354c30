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

"""The host-side pipeline (WAV parsing, writers, reporter, watchdog) built with -fsanitize=address,undefined and driven over
malformed and hostile inputs.  CPU only; GPU AddressSanitizer is not available on the pool."""
import os
import subprocess

from conftest import ROOT


def test_host_pipeline_under_asan_and_ubsan(tmp_path):
    csrc = os.path.join(ROOT, "birda_amd", "csrc")
    exe = str(tmp_path / "host_sanitize_driver")
    srcs = [os.path.join(ROOT, "tests", "native", "host_sanitize_driver.cpp")] + [os.path.join(csrc, f) for f in
                                                                                  ("host_pipeline.cpp", "host_output.cpp", "host_parquet.cpp")]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-pthread", "-o", exe] + srcs
    subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600)
    work = tmp_path / "work"
    work.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    # two well-formed containers for the loader fuzz (BHM1 model, BHC1 custom classifier)
    from birda_amd import modelfile as mf, synth
    model_path, custom_path = str(work / "mini.bhm"), str(work / "custom.bhc")
    mf.write_model(model_path, synth.build_model("mini"))
    mf.write_custom_classifier(custom_path, synth.build_custom_classifier(64, 10, hidden=(32,)))
    p = subprocess.run([exe, str(work), model_path, custom_path], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-6000:]
    assert "host sanitizer driver: ok" in p.stdout

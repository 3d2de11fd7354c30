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
    # ... and two dense-stack ONNX files for the library's own protobuf walk: the reference's geomodel fixture (committed as
    # data) and a three-layer stack written by this repo's ONNX writer (MatMul + Add, Relu, Gemm with transB / alpha / beta)
    import numpy as np
    from birda_amd import onnx_io as ox
    rng = np.random.default_rng(5)
    g = ox.Graph(inputs=[ox.ValueInfo("x", ox.FLOAT, ["n", 3])], outputs=[ox.ValueInfo("y", ox.FLOAT, ["n", 7])])
    g.initializers = {"w0": rng.standard_normal((3, 16)).astype(np.float32), "b0": rng.standard_normal(16).astype(np.float32),
                      "w1": rng.standard_normal((7, 16)).astype(np.float32), "b1": rng.standard_normal(7).astype(np.float32)}
    g.nodes = [ox.Node("MatMul", ["x", "w0"], ["h0"]), ox.Node("Add", ["h0", "b0"], ["h1"]), ox.Node("Relu", ["h1"], ["h2"]),
               ox.Node("Gemm", ["h2", "w1", "b1"], ["h3"], {"transB": 1, "alpha": 0.5, "beta": 2.0}), ox.Node("Sigmoid", ["h3"], ["y"])]
    stack_path = str(work / "stack.onnx")
    open(stack_path, "wb").write(ox.dump(g))
    # ... and two conv-stack graphs for the classifier's own .onnx reader (onnx_conv.hpp): the exporter-style hand-written one
    # (BatchNormalization, SAME padding, Clip, Sigmoid x Mul, residual, ReduceMean, MatMul + Add, Gemm, Softmax) and a
    # squeeze-excite stack behind an audio-input front-end
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_convert import hand_written_graph
    from birda_amd import convert
    conv_paths = [str(work / "hand.onnx"), str(work / "se_audio.onnx"), str(work / "stft_audio.onnx"), str(work / "raw_and_float.onnx")]
    open(conv_paths[0], "wb").write(ox.dump(hand_written_graph()[0]))
    open(conv_paths[1], "wb").write(ox.dump(convert.graph_from_model(synth.build_model("mini_se"), frontend_spelling="fused")))
    # (round 5: the front-end of an audio-input graph is read by the library's own evaluator -- a second spelling puts the STFT,
    #  Gather, Pow / Exp and MatMul operators under the sanitizers too)
    open(conv_paths[2], "wb").write(ox.dump(convert.graph_from_model(synth.build_model("mini"), frontend_spelling="stft")))
    # ... and the file of ADVICE r4 (high): every float32 initializer carries raw_data AND one float_data element
    import struct
    ser = ox._ser_tensor
    ox._ser_tensor = lambda name, arr: ser(name, arr) + (ox._key(4, 5) + struct.pack("<f", 1.0) if np.asarray(arr).dtype == np.float32 and np.asarray(arr).size >= 8 else b"")
    try:
        open(conv_paths[3], "wb").write(ox.dump(convert.graph_from_model(synth.build_model("mini"))))
    finally:
        ox._ser_tensor = ser
    env["BIRDA_FUZZ_CONV_ONNX"] = ":".join(conv_paths)
    fixture = os.path.join(ROOT, "tests", "golden", "reference_fixtures", "fixture-geomodel.onnx")
    p = subprocess.run([exe, str(work), model_path, custom_path, fixture, stack_path], capture_output=True, text=True, timeout=1500, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-6000:]
    assert "host sanitizer driver: ok" in p.stdout
    assert p.stdout.count("\nonnx fuzz") + p.stdout.startswith("onnx fuzz") == 2 and p.stdout.count("conv onnx fuzz") == 4

"""CPU-side checks of the product: the C-ABI library loads and exports every symbol the
headers declare, fails loudly without a device, and the C++ host logic (decoder/segmenter,
batch sizing, CSV, watchdog) agrees with the oracle and the reference's unit-test cases."""
import ctypes as C
import json
import os
import re
import struct
import sys
import time

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def L():
    from birda_amd import _lib
    return _lib.load()


@pytest.fixture(scope="module")
def cases():
    with open(os.path.join(GOLDEN, "reference_unit_cases.json")) as f:
        return json.load(f)


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"BH_API[^;(]*?\b(bhh?_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(L):
    from birda_amd import _lib
    names = _declared("birda_hip.h") + _declared("birda_host.h") + _declared("birda_hip_debug.h")
    assert len(names) >= 40
    # the diagnostic entry points live in their own header: the one birda binds (and the Rust text generated from it) has none
    assert _declared("birda_hip_debug.h") == ["bh_debug_gated_gemm", "bh_debug_mb_stamps", "bh_debug_read_tensor"]
    assert not [n for n in _declared("birda_hip.h") if "debug" in n]
    assert "bh_debug" not in open(os.path.join(ROOT, "include", "birda_hip_sys.rs")).read()
    bound = {n for n, _, _ in _lib.SYMBOLS + _lib.HOST_SYMBOLS}
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/ but not exported"
        assert n in bound, f"{n} has no ctypes prototype in birda_amd/_lib.py"


def test_no_oracle_in_product_library():
    """The product must not link or embed the checker."""
    blob = open(os.path.join(ROOT, "birda_amd", "libbirda_hip.so"), "rb").read()
    assert b"bo_forward" not in blob and b"birda_oracle" not in blob
    for root, _, files in os.walk(os.path.join(ROOT, "birda_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp")):
                src = open(os.path.join(root, f), errors="replace").read()
                assert "from oracle" not in src and "import oracle" not in src and "libbirda_oracle" not in src, f
                if f == "multi.hip":     # opens RCCL (and nothing else) at run time for the optional device-side result gather
                    assert set(re.findall(r'"(lib[^"]*\.so[^"]*)"', src)) == {"librccl.so.1", "librccl.so"}
                    continue
                if f == "trace.hpp":     # opens a ROCTx library when BIRDA_HIP_ROCTX=1 (optional profiler ranges)
                    assert set(re.findall(r'"(lib[^"]*\.so[^"]*)"', src)) == {"librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "libroctx64.so"}
                    continue
                assert "dlopen" not in src, f


def test_backend_name_and_failing_loudly_without_device(L, model_dir):
    import torch
    assert L.bh_backend_name() == b"HIP (gfx950)"
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from birda_amd._lib import BirdaHipError
    from birda_amd.classifier import BirdClassifier
    with pytest.raises(BirdaHipError) as e:
        BirdClassifier(model_dir["mini"][0])
    assert e.value.code == -3 and "no CPU path" in str(e.value)


def test_classifier_create_rejects_bad_inputs(L, tmp_path, model_dir):
    from birda_amd._lib import BirdaHipError
    from birda_amd.classifier import BirdClassifier
    with pytest.raises(BirdaHipError) as e:
        BirdClassifier(str(tmp_path / "missing.bhm"))
    assert e.value.code == -2
    bad = tmp_path / "bad.bhm"
    bad.write_bytes(b"ONNX" + b"\0" * 600)
    with pytest.raises(BirdaHipError) as e:
        BirdClassifier(str(bad))
    assert e.value.code == -2 and "BHM1" in str(e.value)
    # label count must equal the output width (reference src/inference/mod.rs:34-37)
    short = tmp_path / "short.txt"
    short.write_text("a_b\nc_d\n")
    with pytest.raises(BirdaHipError) as e:
        BirdClassifier(model_dir["mini"][0], str(short))
    assert e.value.code == -5
    with pytest.raises(BirdaHipError):
        BirdClassifier(model_dir["mini"][0], top_k=0)


# ---------------- host logic ----------------
def _write_wav(path, pcm_bytes, rate, channels, bits, fmt_tag=1, extra_chunk=False):
    hdr = b"RIFF" + struct.pack("<I", 36 + len(pcm_bytes) + (12 if extra_chunk else 0)) + b"WAVE"
    if extra_chunk:
        hdr += b"LIST" + struct.pack("<I", 3) + b"abc\0"  # odd-sized chunk + pad byte
    hdr += b"fmt " + struct.pack("<IHHIIHH", 16, fmt_tag, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits)
    hdr += b"data" + struct.pack("<I", len(pcm_bytes))
    with open(path, "wb") as f:
        f.write(hdr + pcm_bytes)


def test_decoder_matches_oracle_segmenter_and_pcm_scaling(tmp_path, oracle_lib, cases):
    from birda_amd.pipeline import StreamingDecoder
    OL = oracle_lib.lib()
    rng = np.random.default_rng(5)
    for (channels, bits, tag, extra) in [(1, 16, 1, False), (2, 16, 1, True), (1, 32, 1, False), (2, 32, 3, False), (1, 24, 1, False)]:
        frames = 30011
        if tag == 3:
            raw = rng.uniform(-1, 1, frames * channels).astype("<f4")
            mono = np.zeros(frames, np.float32)
            OL.bo_f32_to_mono(raw, frames, channels, mono)
            data = raw.tobytes()
        elif bits == 16:
            raw = rng.integers(-32768, 32768, frames * channels).astype("<i2")
            mono = np.zeros(frames, np.float32)
            OL.bo_pcm16_to_mono(raw.ctypes.data, frames, channels, mono)
            data = raw.tobytes()
        elif bits == 32:
            raw = rng.integers(-2**31, 2**31, frames * channels).astype("<i4")
            mono = np.zeros(frames, np.float32)
            OL.bo_pcm32_to_mono(raw.ctypes.data, frames, channels, mono)
            data = raw.tobytes()
        else:  # 24-bit: widened to S32 (<< 8) then the S32 rule
            v = rng.integers(-2**23, 2**23, frames * channels).astype("<i4")
            data = b"".join(int(x).to_bytes(3, "little", signed=True) for x in v)
            wide = (v.astype(np.int64) << 8).astype("<i4")
            mono = np.zeros(frames, np.float32)
            OL.bo_pcm32_to_mono(wide.ctypes.data, frames, channels, mono)
        p = str(tmp_path / f"t_{channels}_{bits}_{tag}.wav")
        _write_wav(p, data, 22050, channels, bits, tag, extra)
        d = StreamingDecoder(p)
        assert d.sample_rate() == 22050
        assert abs(d.duration_hint() - frames / 22050) < 1e-12
        want = oracle_lib.segment_stream(mono, 7000, 1300)
        got = []
        while True:
            r = d.next_segment(7000, 1300)
            if r is None:
                break
            got.append((r[0].copy(), r[1]))
        assert [s for _, s in got] == [s for _, s in want]
        for (a, _), (b, _) in zip(got, want):
            assert np.array_equal(a, b)
        d.close()


def test_decoder_reference_segment_traces(tmp_path, cases):
    from birda_amd import synth
    from birda_amd.pipeline import StreamingDecoder
    from birda_amd._lib import BirdaHipError
    for i, c in enumerate(cases["segmenter"]):
        p = str(tmp_path / f"s{i}.wav")
        x = ((np.arange(c["n_samples"]) % 2000) - 1000) / 1000.0 * 0.9
        synth.write_wav_pcm16(p, x, c["rate"])
        d = StreamingDecoder(p)
        starts = []
        while True:
            r = d.next_segment(c["seg"], c["ovl"])
            if r is None:
                break
            starts.append(r[1])
            assert r[0].size == c["seg"]
        assert starts == c["starts"], c["src"]
    d = StreamingDecoder(p)
    with pytest.raises(BirdaHipError):   # overlap >= segment -> Error::Internal (decode.rs:156-162)
        d.next_segment(100, 100)
    with pytest.raises(BirdaHipError):
        StreamingDecoder(str(tmp_path / "nope.wav"))
    junk = tmp_path / "junk.wav"
    junk.write_bytes(b"not a wav file at all")
    with pytest.raises(BirdaHipError):
        StreamingDecoder(str(junk))


def test_batch_sizing_and_estimates_match_reference_cases(L, cases, oracle_lib):
    from birda_amd import pipeline
    OL = oracle_lib.lib()
    for c in cases["estimate_segment_count"]:
        assert pipeline.estimate_segment_count(c["duration"], c["seg"], c["ovl"]) == c["expect"], c["src"]
    for c in cases["batching"]:
        assert pipeline.effective_batch_size(c["batch_size"], c["estimated"]) == c["effective"]
    assert pipeline.effective_batch_size(8, None) == 8
    for c in cases["source_sizing"]:
        assert pipeline.source_samples(c["target"], c["src_rate"], c["dst_rate"]) == c["expect"]
    # f32 arithmetic of (seconds * rate) as usize, against the oracle over awkward values
    for secs, rate in [(3.0, 48000), (5.0, 32000), (0.25, 48000), (1.5, 44100), (0.1, 22050), (2.9999, 48000), (0.0, 48000)]:
        assert L.bhh_duration_to_samples(secs, rate) == OL.bo_duration_to_samples(secs, rate)


def test_csv_writer_matches_reference_cases_and_oracle(L, cases, oracle_lib):
    from birda_amd import pipeline
    OL = oracle_lib.lib()
    for r in cases["csv"]["rows"]:
        assert pipeline.csv_row(r["label"], r["start"], r["end"], r["conf"], r["path"]).decode() == r["row"] + "\n"
    for c in cases["detection_from_label"]:
        assert pipeline.csv_row(c["label"], 0.0, 3.0, 0.5, "f.wav").decode() == f"0.0,3.0,{c['scientific']},{c['common']},0.5000,f.wav\n"
    for e in cases["csv"]["escape"]:
        assert pipeline.csv_row("a_b", 0.0, 3.0, 0.5, e["in"]).decode() == f"0.0,3.0,a,b,0.5000,{e['out']}\n"
    assert list(pipeline.csv_header(True)[:3]) == cases["csv"]["bom"]
    assert pipeline.csv_header(False).decode() == cases["csv"]["header"] + "\n"
    # rounding of {:.1}/{:.4} on awkward f32 values: byte-identical to the oracle
    buf = C.create_string_buffer(8192)
    rng = np.random.default_rng(3)
    for _ in range(200):
        st, conf = np.float32(rng.uniform(0, 4000)), np.float32(rng.uniform(0, 1))
        lab = "Genus x_Name, \"q\"" if rng.random() < 0.3 else "Plain label"
        n = OL.bo_csv_row(lab.encode(), st, np.float32(st + np.float32(3.0)), conf, b"/a b/c,d.wav", buf)
        assert pipeline.csv_row(lab, float(st), float(np.float32(st + np.float32(3.0))), float(conf), "/a b/c,d.wav") == buf.raw[:n]
    for conf in (0.00005, 0.99995, 0.12345, 0.5, 0.25, 0.1):
        n = OL.bo_csv_row(b"a_b", 1.25, 4.25, np.float32(conf), b"f", buf)
        assert pipeline.csv_row("a_b", 1.25, 4.25, float(np.float32(conf)), "f") == buf.raw[:n]


def test_watchdog_cancelled_when_dropped(L, monkeypatch):
    """reference src/gpu/watchdog.rs:73-84: a dropped guard must not kill the process."""
    g = L.bhh_watchdog_start(300, 32)
    L.bhh_watchdog_cancel(g)
    time.sleep(0.6)
    monkeypatch.delenv("BIRDA_INFERENCE_TIMEOUT", raising=False)
    assert L.bhh_watchdog_timeout_secs() == 10          # processor.rs:196
    for v, want in (("25", 25), ("0", 10), ("3601", 10), ("abc", 10), ("3600", 3600), ("1", 1)):
        monkeypatch.setenv("BIRDA_INFERENCE_TIMEOUT", v)
        assert L.bhh_watchdog_timeout_secs() == want    # processor.rs:205-211


def test_resample_output_length_matches_the_block_rule(L, oracle_lib):
    """whole blocks + ceil(remaining * to / from) (reference src/audio/resample.rs:34-88); no GPU needed."""
    import ctypes as C
    for frm, to in ((44100, 48000), (22050, 48000), (48000, 32000), (44100, 32000), (48000, 48000)):
        for n in (0, 1, 500, 1026, 1029, 1030, 66150, 132300, 144000, 240000):
            out = C.c_size_t()
            assert L.bh_resample_output_len(n, frm, to, C.byref(out)) == 0
            want = len(oracle_lib.resample(np.zeros(n, np.float32), frm, to)) if n else 0
            assert out.value == want, (frm, to, n)


def test_segment_starts_match_reference_traces_and_the_oracle_segmenter(L, cases, oracle_lib):
    """bh_segment_starts = StreamingDecoder::next_segment's start positions (decode.rs:150-202), including
    the trailing tail segment an overlap leaves behind; pure host arithmetic, no GPU needed."""
    def starts(n, seg, ovl):
        cnt = L.bh_segment_starts(n, seg, ovl, None, 0)
        buf = (C.c_uint64 * max(cnt, 1))()
        assert L.bh_segment_starts(n, seg, ovl, buf, cnt) == cnt
        return [int(buf[i]) for i in range(cnt)]
    for c in cases["segmenter"]:
        assert starts(c["n_samples"], c["seg"], c["ovl"]) == c["starts"], c["src"]
    rng = np.random.default_rng(5)
    for _ in range(40):
        seg = int(rng.integers(2, 60)); ovl = int(rng.integers(0, seg)); n = int(rng.integers(0, 400))
        want = [s for _, s in oracle_lib.segment_stream(np.zeros(n, np.float32), seg, ovl, packet=int(rng.integers(1, 50)))]
        assert starts(n, seg, ovl) == want, (n, seg, ovl)
    assert starts(100, 10, 10) == [] and starts(100, 10, 12) == []     # overlap >= segment: Error::Internal upstream


def test_device_code_has_no_half_swapped_packed_f32_ops(tmp_path):
    """Regression guard for a measured hazard: hipcc's SLP-vectorised
    `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` (operand halves swapped) made the split-f16 mel kernel
    nondeterministic on gfx950.  No packed-f32 instruction of the library may select operand halves
    with `op_sel:` (the `op_sel_hi:` broadcast forms are fine); sums that the compiler would pack that
    way are written with `bh_add_unpacked`."""
    import re
    import subprocess
    # the device code that SHIPS: every gfx950 code object bundled in libbirda_hip.so's .hip_fatbin section, disassembled
    # (recompiling the sources to assembly took three minutes of the CPU suite once the fused kernel had 255 instantiations)
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm llvm tools not available")
    lib = os.path.join(ROOT, "birda_amd", "libbirda_hip.so")
    fat = str(tmp_path / "fat.bin")
    subprocess.run([tools[0], "--dump-section", ".hip_fatbin=" + fat, lib], check=True, capture_output=True, timeout=120)
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), data)]
    assert len(starts) >= 5, "one bundle per .hip translation unit with kernels"
    n_packed = 0
    for k, a in enumerate(starts):
        piece = str(tmp_path / f"bundle{k}.bin")
        open(piece, "wb").write(data[a:starts[k + 1] if k + 1 < len(starts) else len(data)])
        code = str(tmp_path / f"code{k}.o")
        subprocess.run([tools[1], "--unbundle", "--type=o", "--input=" + piece, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--output=" + code], check=True, capture_output=True, timeout=120)
        text = subprocess.run([tools[2], "-d", code], check=True, capture_output=True, text=True, timeout=600).stdout
        n_packed += len(re.findall(r"v_pk_(?:add|mul|fma)_f32\b", text))
        bad = [l.strip() for l in text.splitlines() if re.search(r"v_pk_(add|mul|fma)_f32\b.*\bop_sel:\[", l)]
        assert not bad, (k, bad[:3])
    assert n_packed > 1000      # the disassembly really is the library's device code (the GELU alone is packed f32)


def test_shipped_hot_kernels_fit_their_register_budget():
    """The kernels the planner picks for the BirdNET-shaped model, read from the code objects in libbirda_hip.so
    (tools/kernel_resources.py: metadata notes, no GPU): the early blocks -- vector-issue-bound, every scratch access is issue
    slots lost -- must not spill at all and must keep the registers of 4 / 3 waves per SIMD; the late blocks must be the 8-wave
    workgroups (512 threads) at two waves per SIMD without scratch, the Perch-sized stack's wide 5x5 blocks too; the front-end kernel two workgroups per
    CU without scratch.  A source change that silently costs one of these shows up here before it shows up in a profile."""
    import subprocess, sys
    llvm = "/opt/rocm/lib/llvm/bin"
    if not all(os.path.exists(os.path.join(llvm, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")):
        pytest.skip("ROCm llvm tools not available")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"),
                          os.path.join(ROOT, "birda_amd", "libbirda_hip.so")], check=True, capture_output=True, text=True, timeout=600).stdout
    res = {}
    for line in out.splitlines():
        m = re.match(r"(\S.*?)\s+vgpr\s+(\d+) agpr\s+(\d+) sgpr\s+\d+ scratch\s+(\d+) threads (\d+)", line)
        if m:
            res[m.group(1).replace(" ", "")] = tuple(int(v) for v in m.groups()[1:])
    def k(args, se=0):      # (the last template argument: 1 = pass A of a squeeze-excite block, round 5)
        return res["mbconv_kernel<" + args + "," + str(se) + ">"]
    # stem, 16->96->24, 24->144->24 (f16x3, GELU): no scratch, 4 / 4 / 3 waves per SIMD
    for args, max_regs in (("3,1,16,1,3,1,4,1,2,1,4,1,1,4,2,3,0,4,0", 128), ("3,2,16,1,5,1,4,1,1,2,4,0,1,4,0,3,0,4,0", 128),
                           ("3,1,16,1,3,1,4,1,2,2,4,1,1,3,0,3,0,4,0", 168)):
        vgpr, agpr, scratch, threads = k(args)
        assert scratch == 0 and vgpr + agpr <= max_regs and threads == 256, (args, k(args))
    # the nine late blocks' five instantiations: 8 waves, 256 registers, NO scratch (round 4: the column tasks' sums pinned per
    # window column, the 192 -> 1152 -> 320 block without the expand phase's column split) ...
    for args in ("3,1,32,3,3,2,4,2,3,3,5,3,1,2,0,3,0,4,6", "5,1,32,3,3,2,4,2,3,4,5,3,1,2,0,3,0,4,6", "5,1,32,4,3,2,4,2,3,4,5,3,1,2,0,3,0,4,6",
                 "5,1,32,6,2,2,2,4,3,3,4,2,2,2,0,3,0,4,3", "3,1,32,6,1,1,2,4,3,5,4,2,2,2,0,3,0,4,3"):
        vgpr, agpr, scratch, threads = k(args)
        assert threads == 512 and vgpr + agpr <= 256 and scratch == 0, (args, k(args))
    # ... and the Perch-sized stack's 5x5 late blocks (136 and 232 channels: five and eight k steps of resident A fragments), which
    # spilled 70-130 registers until round 4 (swish copies; profiles/r4_k_perch_blocks.txt)
    for args in ("5,1,32,3,2,1,4,2,4,5,5,3,1,2,0,3,0,3,8", "5,1,32,5,2,1,8,1,2,9,5,3,1,2,0,3,0,3,8", "5,1,32,8,1,1,4,2,2,8,4,2,2,2,0,3,0,3,4",
                 "3,1,32,12,1,2,2,4,2,6,4,2,1,2,0,3,0,3,4"):
        vgpr, agpr, scratch, threads = k(args)
        assert threads == 512 and vgpr + agpr <= 256 and scratch == 0, (args, k(args))
        # ... and their squeeze-excite pass-A twins (round 5: the gated Perch-sized plan runs on these) cost no register and no scratch
        # more (a first version that stored D from the depthwise tasks themselves cost 40-120 registers and spilled)
        v2, a2, s2, t2 = k(args, se=1)
        assert t2 == 512 and s2 == 0 and v2 + a2 <= 256, (args, k(args, se=1))
    # (round 6: the software-pipelined loop -- BH_MEL_PIPE 2, the operator held half a step at a time -- parks three thread-invariant
    #  dwords in scratch and reloads them once per work item, at the loop's exit; measured 4.2 % faster than the spill-free plain loop,
    #  profiles/r6_g_mel_pipe_tuning.txt.  Round 4's pipelined form needed 148 bytes and reloaded them inside the MFMA tail.)
    vgpr, agpr, scratch, threads = res["mel_kernel<6,3,1>"]
    assert scratch <= 16 and vgpr + agpr <= 256


def test_shipped_tile_configurations_are_the_reachable_ones():
    """the product library ships what the planner can reach, nothing else.  tools/plan_models.py walks every
    synthetic model (BirdNET-v2.4-shaped, Perch-sized, Perch-shaped, the small test stacks) in the three precisions and the three
    templated activations through bh_plan_fused_blocks (host logic, no GPU); every MB_ENTRY row of mbconv_cfgs.inc must be picked
    somewhere, and the rejected experiments (MB_XENTRY rows, kernels_mbwave.hip, the persistent variants) must not be in the
    build."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import plan_models
    from birda_amd import _lib, modelfile as mf
    lib = _lib.load()
    buf = C.create_string_buffer(128)
    n_total = 0
    while lib.bh_mb_config_name(n_total, buf, 128) > 0:
        n_total += 1
    assert n_total % 3 == 0
    n_base = n_total // 3
    shipped = set()
    for ci in range(n_total):
        lib.bh_mb_config_name(ci, buf, 128)
        args = [int(v) for v in buf.value.decode().split(",")]
        if args[0] != 0:
            shipped.add(ci)
            assert args[16] == 0, f"configuration {ci}: a persistent-workgroup instantiation in the product build"
    reach, by_model = plan_models.survey(acts=(None, mf.ACT_SWISH, mf.ACT_RELU6))
    assert reach <= shipped
    # Round 6: entries 211 .. are GENERIC -- shaped for a class of blocks (input / output width x depthwise kernel / stride), not for
    # a block of this repo's plans; what holds them is tests/test_plan_coverage.py (every block of 57 plans nobody tiled by hand runs
    # fused) and tests/test_random_plans_gpu.py.  Below 211 the rule stands: shipped = what the planner reaches on the repo's models.
    GENERIC0 = 211
    unreached = sorted({c % n_base for c in shipped if c % n_base < GENERIC0} - {c % n_base for c in reach})
    assert not unreached, f"shipped but never picked by the planner: {unreached}"
    # the headline models fuse every block they can in the f16 modes
    assert len(by_model["birdnet_v24/default/f16x3"]["fused"]) == 16 and not by_model["birdnet_v24/default/f16x3"]["unfused_triples"]
    assert len(by_model["perch_v2/default/f16x3"]["fused"]) == 26 and len(by_model["perch_v2/default/f16"]["fused"]) == 26
    assert len(by_model["perch_v2/default/f32"]["fused"]) == 26 and not by_model["perch_v2/default/f32"]["unfused_triples"]   # (round 4)
    assert not by_model["perch_v2/default/f16x3"]["unfused_triples"]
    # the small-launch twins ride behind their blocks in the listing: one-segment tiles (133 / 135) and, since round 4, the narrow
    # 8-column tiles of the nine whole-image late blocks (193-202), each with its block's chunk and project-tile layout
    twins = {c % n_base for _, c in by_model["birdnet_v24/default/f16x3"]["twins"]}
    assert {133, 135} <= twins and {193, 195, 197, 199, 201} <= twins, sorted(twins)
    assert not os.path.exists(os.path.join(ROOT, "birda_amd", "csrc", "kernels_mbwave.hip"))
    so = os.path.getsize(os.path.join(ROOT, "birda_amd", "libbirda_hip.so"))
    # (8.2 MiB in round 3; + the f32 twins of the Perch-sized stack and the narrow-tile twins: 9.0; round 5: + every swish entry a second
    #  time as pass A of a squeeze-excite block, and the front-end evaluator: 10.8)
    #  round 6: + 82 generic / late-stage entries x 3 activations, and pass A of a squeeze-excite block for GELU and ReLU6 as well: 22.5)
    assert so < 25 * 2 ** 20, f"libbirda_hip.so grew to {so / 2 ** 20:.1f} MiB"


# ---------------- range filter tables (host logic, include/birda_host.h) ----------------
def test_species_mapping_and_projection_match_reference_cases_and_oracle(L, cases, oracle_lib):
    from birda_amd import pipeline
    for c in cases["scientific_name"]:
        assert pipeline.scientific_name(c["label"]) == c["expect"], c["src"]
    for c in cases["geomodel_projection"]:
        reported = [tuple(r) for r in c["reported"]]
        got, summary = pipeline.project_scores(c["geomodel"], reported, c["classifier"], 0.01)
        want = np.asarray([np.nan if v is None else v for v in c["scores"]], np.float32)
        assert (summary.mapped, summary.unmatched, summary.total) == (c["mapped"], c["unmatched"], len(c["classifier"])), c["src"]
        assert np.array_equal(got, want, equal_nan=True), c["src"]
        ref, mapped = oracle_lib.project_scores(c["geomodel"], reported, c["classifier"])
        assert np.array_equal(got, ref, equal_nan=True) and mapped == summary.mapped, c["src"]
        for thr, n in c.get("in_range", []):
            assert pipeline.project_scores(c["geomodel"], reported, c["classifier"], thr)[1].in_range == n, c["src"]
    # a label set at the real scale (6 522 classes against 12 012 geomodel species), host against oracle
    rng = np.random.default_rng(5)
    genus = [f"G{g:03d}" for g in range(400)]
    pool = [f"{genus[i % 400]} s{i // 400:02d}" for i in range(13000)]
    cls = [f"{pool[i]}_Common {i}" for i in rng.permutation(13000)[:6522]]
    cls[10] = cls[3].split("_")[0] + "_Duplicate"            # collision: the first label wins
    cls[20] = "Dog_Dog"; cls[21] = "Accelerating_and_revving_and_vroom"
    geo = [f"{pool[i].upper() if i % 7 == 0 else pool[i]}_English {i}" for i in rng.permutation(13000)[:12012]]
    reported = [(g, float(rng.random())) for g in geo[::3]]
    got, summary = pipeline.project_scores(geo, reported, cls, 0.03)
    ref, mapped = oracle_lib.project_scores(geo, reported, cls)
    assert np.array_equal(got, ref, equal_nan=True) and summary.mapped == mapped
    assert np.isnan(got[10]) and np.isnan(got[20]) and np.isnan(got[21])
    assert summary.in_range == int((ref >= np.float32(0.03)).sum()) and 0 < summary.mapped < 6522


# ---------------- directory mode (coordinator.rs:146-190) ----------------
def test_collect_input_files_walks_directories_like_the_reference(L, tmp_path):
    from birda_amd import pipeline
    root = tmp_path / "rec"
    (root / "a" / "deep").mkdir(parents=True)
    (root / "b").mkdir()
    made = ["a/one.wav", "a/deep/two.FLAC", "b/three.Mp3", "b/four.m4a", "five.aac"]
    for rel in made + ["a/notes.txt", "b/wav", "a/.wav", "b/clip.wav.bak", "six.ogg"]:
        (root / rel).write_bytes(b"x")
    single = tmp_path / "solo.WAV"; single.write_bytes(b"x")
    other = tmp_path / "readme.md"; other.write_bytes(b"x")
    got = pipeline.collect_input_files([str(root), str(single), str(other), str(tmp_path / "missing")])
    assert sorted(got) == sorted([str(root / r) for r in made] + [str(single)])
    assert got == pipeline.collect_input_files([str(root), str(single)])          # repeatable order
    assert got[-1] == str(single)                                                  # argument order is kept (:149-160)
    assert pipeline.collect_input_files([]) == [] and pipeline.collect_input_files([str(other)]) == []
    for name, want in (("x.wav", 1), ("x.WaV", 1), ("x.flac", 1), ("x.mp3", 1), ("x.m4a", 1), ("x.aac", 1),
                       ("x.ogg", 0), ("wav", 0), (".wav", 0), ("dir.wav/x", 0), ("x.wav.txt", 0)):
        assert L.bhh_is_audio_file(name.encode()) == want, name


def test_directory_mode_assignment_by_duration():
    from birda_amd import sharding
    rng = np.random.default_rng(3)
    for world in (1, 2, 4, 8):
        for n in (0, 1, 5, 40):
            d = (rng.random(n) * 600 + 5).tolist()
            parts = sharding.assign_by_duration(d, world)
            assert len(parts) == world and sum(parts, []) == list(range(n))      # contiguous runs, file order kept
            if n >= 8 * world:
                loads = [sum(d[i] for i in p) for p in parts]
                assert max(loads) - min(loads) <= 2 * max(d)
    assert sharding.assign_by_duration([0.0, 0.0, 0.0, 0.0], 2) == [[0, 1], [2, 3]]
    assert sharding.assign_by_duration([100.0, 1.0, 1.0, 100.0], 2) == [[0, 1], [2, 3]]


def test_environment_names_in_the_shipped_library_are_the_documented_ones():
    """debug / A-B getenv knobs are compiled out of the product build (BH_XENV, kernels.hpp: they exist behind
    `make EXPERIMENTS=1`).  What `strings` finds in the shipped .so must be the documented list."""
    import re
    import subprocess
    so = os.path.join(ROOT, "birda_amd", "libbirda_hip.so")
    names = set(re.findall(r"^BIRDA_[A-Z0-9_]+$", subprocess.run(["strings", "-n", "8", so], capture_output=True, text=True, check=True).stdout, flags=re.M))
    allowed = {"BIRDA_HIP_PRECISION",           # overrides bh_config.flags (birda_hip.h)
               "BIRDA_HIP_COPY_THREADS", "BIRDA_HOST_PIPELINE_DEPTH",   # host-side tuning
               "BIRDA_HOST_PREAD",              # =1: bhh_process_file hands the WAV over by descriptor (bh_predict_pcm_fd_rows) instead of mapped: measured, no gain
               "BIRDA_HIP_SE_GROUP_MB",         # squeeze-excite blocks in groups of segments (measured and left off: profiles/r6_i_se_groups.txt)
               "BIRDA_INFERENCE_TIMEOUT",       # the reference's own variable (processor.rs:194-211)
               "BIRDA_HIP_ROCTX", "BIRDA_HOST_TIMING",                 # tracing / phase times
               # switches the parity tests drive the product library with: debug contexts, forced tile configurations, layer-by-layer
               # execution, either front-end kernel
               "BIRDA_HIP_KEEP_TENSORS", "BIRDA_HIP_KEEP_FUSED", "BIRDA_HIP_MB_CFG", "BIRDA_HIP_MB_PREFER", "BIRDA_HIP_MB_WHY", "BIRDA_HIP_FUSE",
               "BIRDA_HIP_MEL32", "BIRDA_HIP_MEL_F32", "BIRDA_HIP_HEAD_GAP",
               "BIRDA_HIP_FUSE_SE"}             # round 5: =0 runs squeeze-excite blocks layer by layer (::test_squeeze_excite_and_swish_stack_matches_oracle)
    assert names <= allowed, sorted(names - allowed)
    assert "BIRDA_HIP_PRECISION" in names

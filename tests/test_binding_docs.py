"""The three spellings of the C ABI agree: include/*.h (the boundary), the Rust text a maintainer pastes (INTEGRATION.md and the
generated include/birda_hip_sys.rs) and the ctypes mirror the tests run through (birda_amd/_lib.py).

Round 2's INTEGRATION.md declared `BhModelInfo` with 12 fields against the header's 14: `bh_classifier_info` would have written
64 bytes into a 56-byte Rust struct.  Nothing compared the two; this file does, with gcc's own sizeof / offsetof as the judge of
what the header means."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import abi_parse as A  # noqa: E402

HIP_H = os.path.join(ROOT, "include", "birda_hip.h")
HOST_H = os.path.join(ROOT, "include", "birda_host.h")


@pytest.fixture(scope="module")
def header():
    structs, functions, defines = A.parse_c_header(HIP_H)
    hs, hf, _ = A.parse_c_header(HOST_H, defines)
    return structs, functions, hs, hf


def _eq_ctypes(a: str, b: str) -> bool:
    """ctypes has one type for size_t and uint64_t on this platform"""
    norm = lambda s: s.replace("usize", "u64")
    return norm(a) == norm(b)


def test_parser_sees_the_whole_header(header):
    structs, functions, hs, hf = header
    n_api = len(re.findall(r"\bBH_API\b", A.strip_c_comments(open(HIP_H).read()))) - 1     # minus the #define itself
    assert len(functions) == n_api and n_api >= 70
    assert set(structs) == {"bh_config", "bh_model_info", "bh_result", "bh_provider_status", "bh_multi_config"}
    assert len(structs["bh_model_info"]) == 14
    n_host = len(re.findall(r"\bBH_API\b", A.strip_c_comments(open(HOST_H).read())))
    assert len(hf) == n_host


def test_gcc_layout_of_every_struct_matches_the_parsed_fields(header, tmp_path):
    """sizeof / offsetof from the compiler against the natural-alignment layout of the parsed field list: the parser (and with
    it every comparison below) reads the header the way gcc does."""
    structs, _, hs, _ = header
    allst = {**structs, **hs}
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HOST_H}"', "int main(void){"]
    for name, fields in allst.items():
        src.append(f'printf("{name} %zu", sizeof({name}));')
        for f, _cls in fields:
            src.append(f'printf(" %zu", offsetof({name}, {f}));')
        src.append('printf("\\n");')
    src.append("return 0;}")
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    for line in out.strip().splitlines():
        name, size, *offs = line.split()
        want_size, want_offs = A.layout(allst[name])
        assert int(size) == want_size, name
        assert [int(o) for o in offs] == [o for _, o in want_offs], name
    assert A.layout(structs["bh_model_info"])[0] == 64


def test_generated_rust_binding_is_current_and_complete(header):
    structs, functions, _, _ = header
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"], check=True)
    rs, rf = A.parse_rust(open(os.path.join(ROOT, "include", "birda_hip_sys.rs")).read())
    assert set(rf) == set(functions)
    for name, sig in functions.items():
        assert rf[name] == sig, name
    for name, fields in structs.items():
        assert rs[A.rust_struct_name(name)] == fields, name


def test_every_rust_declaration_in_integration_md_matches_the_header(header):
    structs, functions, _, _ = header
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```rust\n(.*?)```", md, flags=re.S)
    assert len(blocks) >= 4
    seen_structs, seen_fns = set(), set()
    by_rust_name = {A.rust_struct_name(k): k for k in structs}
    for b in blocks:
        rs, rf = A.parse_rust(b)
        for name, fields in rs.items():
            assert name in by_rust_name, f"INTEGRATION.md declares #[repr(C)] struct {name}: not in birda_hip.h"
            want = structs[by_rust_name[name]]
            assert [f for f, _ in fields] == [f for f, _ in want], f"{name}: field names / order / count"
            assert [c for _, c in fields] == [c for _, c in want], f"{name}: field widths"
            assert A.layout(fields) == A.layout(want)
            seen_structs.add(name)
        for name, sig in rf.items():
            assert name in functions, f"INTEGRATION.md declares fn {name}: not in birda_hip.h"
            assert sig[1] == functions[name][1], f"{name}: arguments {sig[1]} vs header {functions[name][1]}"
            assert sig[0] == functions[name][0], f"{name}: return type {sig[0]} vs header {functions[name][0]}"
            seen_fns.add(name)
    # the excerpts must keep covering what hip_backend.rs calls first
    assert {"BhConfig", "BhModelInfo", "BhResult", "BhProviderStatus", "BhMultiConfig"} <= seen_structs
    assert {"bh_classifier_create", "bh_classifier_info", "bh_predict_batch", "bh_predict_batch_with_context",
            "bh_batch_context_create", "bh_select_provider", "bh_default_batch_size", "bh_multi_create",
            "bh_predict_batch_contig", "bh_host_alloc"} <= seen_fns


def test_a_stale_rust_struct_is_caught():
    """the defect of round 2, replayed: the checker must see it"""
    structs, _, _ = A.parse_c_header(HIP_H)
    old = """#[repr(C)] #[derive(Default)]
    struct BhModelInfo { sample_rate: u32, segment_duration: f32, sample_count: u32, n_classes: u32,
                         embedding_dim: u32, output_activation: u32, spec_channels: u32, spec_h: u32,
                         spec_w: u32, n_layers: u32, macs_per_segment: u64, mel_flops_per_segment: u64 }"""
    rs, _ = A.parse_rust(old)
    assert A.layout(rs["BhModelInfo"])[0] == 56
    assert rs["BhModelInfo"] != structs["bh_model_info"]
    _, rf = A.parse_rust('extern "C" { fn bh_default_batch_size(model_type: u32, provider_actual: *const c_char) -> u32; }')
    _, functions, _ = A.parse_c_header(HIP_H)
    assert rf["bh_default_batch_size"] != functions["bh_default_batch_size"]


def test_ctypes_mirror_matches_the_header(header):
    structs, functions, hs, hf = header
    from birda_amd import _lib
    mirrors = {"bh_config": _lib.BhConfig, "bh_model_info": _lib.BhModelInfo, "bh_result": _lib.BhResult,
               "bh_provider_status": _lib.BhProviderStatus, "bh_multi_config": _lib.BhMultiConfig,
               "bhh_writer_options": _lib.BhhWriterOptions, "bhh_range_filter_info": _lib.BhhRangeFilterInfo,
               "bhh_processing_config": _lib.BhhProcessingConfig, "bhh_process_result": _lib.BhhProcessResult,
               "bhh_bsg_metadata": _lib.BhhBsgMetadata}
    allst = {**structs, **hs}
    assert set(mirrors) == set(allst)
    for name, cls in mirrors.items():
        want = allst[name]
        got = [(f, A.ctypes_class(t)) for f, t in cls._fields_]
        assert [f for f, _ in got] == [f for f, _ in want], name
        for (f, a), (_, b) in zip(got, want):
            assert _eq_ctypes(a, b), (name, f, a, b)
        assert C.sizeof(cls) == A.layout(want)[0], name
        for f, off in A.layout(want)[1]:
            assert getattr(cls, f).offset == off, (name, f)
    # (the ctypes table also binds the diagnostic entry points of include/birda_hip_debug.h, which the tests call)
    _, dbg_functions, _ = A.parse_c_header(os.path.join(ROOT, "include", "birda_hip_debug.h"))
    dbg_functions = {k: v for k, v in dbg_functions.items() if k.startswith("bh_debug_")}
    assert len(dbg_functions) == 3
    for table, fns in ((_lib.SYMBOLS, {**functions, **dbg_functions}), (_lib.HOST_SYMBOLS, hf)):
        assert {n for n, _, _ in table} == set(fns)
        for name, res, args in table:
            want_ret, want_args = fns[name]
            assert _eq_ctypes(A.ctypes_class(res), want_ret), name
            assert len(args) == len(want_args), name
            for i, (a, b) in enumerate(zip(args, want_args)):
                assert _eq_ctypes(A.ctypes_class(a), b), (name, i, a, b)

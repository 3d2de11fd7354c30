#!/usr/bin/env python3
"""Writes include/birda_hip_sys.rs -- the complete Rust `extern "C"` side of include/birda_hip.h -- from the header itself, so
that the Rust text a maintainer pastes can never drift from the C ABI again (round 2: INTEGRATION.md's BhModelInfo was 8 bytes
short of bh_model_info).  tests/test_binding_docs.py re-runs the generator and compares; it also holds INTEGRATION.md's excerpts
to the header.

    python tools/gen_rust_ffi.py            # rewrite include/birda_hip_sys.rs
    python tools/gen_rust_ffi.py --check    # exit 1 when the file on disk is stale
"""
from __future__ import annotations

import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import abi_parse as A  # noqa: E402

HEADER = os.path.join(ROOT, "include", "birda_hip.h")
OUT = os.path.join(ROOT, "include", "birda_hip_sys.rs")

RUST_OF = {"i8": "c_char", "u8": "u8", "i16": "i16", "u16": "u16", "i32": "i32", "u32": "u32", "i64": "i64", "u64": "u64",
           "usize": "usize", "f32": "f32", "f64": "f64"}


def rust_type(c_type: str, array, defines, opaque) -> str:
    t = " ".join(c_type.split())
    if " ".join(re.sub(r"\bconst\b", " ", t).split()) in defines.get("__fnptr__", ()):
        return " ".join(re.sub(r"\bconst\b", " ", t).split())      # a `pub type` alias emitted below
    stars = t.count("*")
    if stars:
        # walk the declarator right to left: `const float *const *` = pointer to const pointer to const float
        parts = [p.strip() for p in t.split("*")]
        base = parts[0]
        base_const = bool(re.search(r"\bconst\b", base))
        base_name = " ".join(re.sub(r"\b(const|struct)\b", " ", base).split())
        if base_name == "void":
            inner = "c_void"
        elif base_name in opaque:
            inner = A.rust_struct_name(base_name)
        elif base_name in A.C_SCALARS:
            inner = RUST_OF[A.C_SCALARS[base_name]]
        else:
            inner = A.rust_struct_name(base_name)
        out = inner
        consts = [base_const] + [bool(re.search(r"\bconst\b", p)) for p in parts[1:-1]]
        for c in consts:
            out = ("*const " if c else "*mut ") + out
        return out
    cls = A.c_class(t, None, defines)
    r = "c_int" if " ".join(re.sub(r"\bconst\b", " ", t).split()) == "int" else RUST_OF[cls]
    if array is not None:
        n = defines[array] if array in defines else int(array)
        r = f"[{r}; {n}]"
    return r


def generate() -> str:
    raw = open(HEADER).read()
    defines = A._c_defines(raw)
    text = A.strip_c_comments(raw)
    opaque = re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", text)
    fnptr = A.fn_typedefs(text)
    defines["__fnptr__"] = set(fnptr)
    lines = [
        "// birda_hip_sys.rs -- GENERATED from include/birda_hip.h by tools/gen_rust_ffi.py; do not edit.",
        "// The raw `extern \"C\"` surface of libbirda_hip.so for src/inference/hip_backend.rs (INTEGRATION.md section 2).",
        "#![allow(non_camel_case_types, dead_code)]",
        "use std::ffi::{c_char, c_int, c_void};",
        "",
    ]
    for m in re.finditer(r"^[ \t]*#define[ \t]+(BH_\w+)[ \t]+(0x[0-9a-fA-F]+|-?\d+)(u?)[ \t]*$", text, flags=re.M):
        ty = "u32" if m.group(3) else ("i32" if m.group(2).startswith("-") else "usize")   # (a negative value: an index sentinel)
        lines.append(f"pub const {m.group(1)}: {ty} = {m.group(2)};")
    for m in re.finditer(r"typedef\s+enum\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        for e in m.group(1).split(","):
            e = e.strip()
            if e:
                name, val = [s.strip() for s in e.split("=")]
                lines.append(f"pub const {name}: c_int = {val};")
    lines.append("")
    for o in opaque:
        lines.append(f"pub enum {A.rust_struct_name(o)} {{}}   // opaque handle `{o}`")
    lines.append("")
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        lines.append("#[repr(C)]")
        lines.append(f"pub struct {A.rust_struct_name(m.group(2))} {{")
        for t, name, arr in A._split_c_declarators(m.group(1)):
            lines.append(f"    pub {name}: {rust_type(t, arr, defines, opaque)},")
        lines.append("}")
        lines.append("")
    for name, (ret, args) in fnptr.items():
        rargs = []
        for a in args.split(","):
            mm = re.match(r"^(.*?)(\**)\s*(\w+)$", a.strip())
            rargs.append(f"{mm.group(3)}: {rust_type(mm.group(1) + mm.group(2), None, defines, opaque)}")
        rret = "" if ret == "void" else f" -> {rust_type(ret, None, defines, opaque)}"
        lines.append(f'pub type {name} = Option<unsafe extern "C" fn({", ".join(rargs)}){rret}>;')
    lines.append("")
    lines.append('#[link(name = "birda_hip")]')
    lines.append('extern "C" {')
    for m in re.finditer(r"\bBH_API\s+([^;(]*?)(\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        rargs = []
        if args and args != "void":
            for a in args.split(","):
                mm = re.match(r"^(.*?)(\**)\s*(\w+)$", a.strip())
                pname = mm.group(3)
                if pname in ("in", "type", "fn", "mod", "ref", "use", "move", "match", "loop", "box", "self"):
                    pname += "_"
                rargs.append(f"{pname}: {rust_type(mm.group(1) + mm.group(2), None, defines, opaque)}")
        rret = "" if ret == "void" else f" -> {rust_type(ret, None, defines, opaque)}"
        lines.append(f"    pub fn {name}({', '.join(rargs)}){rret};")
    lines.append("}")
    return "\n".join(lines) + "\n"


def main():
    text = generate()
    if "--check" in sys.argv:
        on_disk = open(OUT).read() if os.path.exists(OUT) else ""
        if on_disk != text:
            print(f"{OUT} is stale: run python tools/gen_rust_ffi.py", file=sys.stderr)
            sys.exit(1)
        return
    with open(OUT, "w") as f:
        f.write(text)
    print(f"wrote {OUT} ({text.count(chr(10))} lines)")


if __name__ == "__main__":
    main()

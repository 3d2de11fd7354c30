"""Parsers that reduce the three spellings of the C ABI -- the C headers under include/, the Rust text a maintainer pastes
(include/birda_hip_sys.rs) and the ctypes mirror (birda_amd/_lib.py) -- to one canonical form, so that a test
can compare them field by field and argument by argument.

Canonical ABI classes: "i8" (char), "u8", "i16", "u16", "i32", "u32", "i64", "u64", "usize", "f32", "f64", "ptr", "void", and
arrays as "<class>[<n>]".  Pointers compare as "ptr" whatever they point at (constness and pointee are documentation).
"""
from __future__ import annotations

import re

C_SCALARS = {
    "char": "i8", "signed char": "i8", "unsigned char": "u8", "int8_t": "i8", "uint8_t": "u8", "int16_t": "i16",
    "uint16_t": "u16", "int": "i32", "int32_t": "i32", "unsigned": "u32", "unsigned int": "u32", "uint32_t": "u32",
    "int64_t": "i64", "uint64_t": "u64", "long long": "i64", "unsigned long long": "u64", "size_t": "usize",
    "float": "f32", "double": "f64", "void": "void",
}
RUST_SCALARS = {
    "c_char": "i8", "i8": "i8", "u8": "u8", "i16": "i16", "u16": "u16", "c_int": "i32", "i32": "i32", "c_uint": "u32",
    "u32": "u32", "i64": "i64", "u64": "u64", "usize": "usize", "f32": "f32", "f64": "f64", "c_float": "f32",
    "c_double": "f64", "()": "void",
}
SIZES = {"i8": 1, "u8": 1, "i16": 2, "u16": 2, "i32": 4, "u32": 4, "i64": 8, "u64": 8, "usize": 8, "f32": 4, "f64": 8, "ptr": 8}


def strip_c_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _c_defines(text: str) -> dict:
    out = {}
    for m in re.finditer(r"^[ \t]*#define[ \t]+(\w+)[ \t]+(0x[0-9a-fA-F]+|-?\d+)(u?)[ \t]*$", strip_c_comments(text), flags=re.M):
        out[m.group(1)] = int(m.group(2), 0)
    return out


def c_class(decl_type: str, array: str | None, defines: dict) -> str:
    t = decl_type.strip()
    if "*" in t:
        cls = "ptr"
    else:
        t = re.sub(r"\b(const|volatile|struct|enum)\b", " ", t)
        t = " ".join(t.split())
        if t in defines.get("__fnptr__", ()):     # `typedef void (*name)(...)`: a function pointer
            return "ptr" if array is None else f"ptr[{array}]"
        if t not in C_SCALARS:
            raise ValueError(f"unknown C type {decl_type!r}")
        cls = C_SCALARS[t]
    if array is not None:
        n = defines[array] if array in defines else int(array)
        cls = f"{cls}[{n}]"
    return cls


def fn_typedefs(text: str) -> dict:
    """`typedef void (*name)(args);` -> {name: (return type, "args")} (comment-stripped text)"""
    return {m.group(2): (" ".join(m.group(1).split()), " ".join(m.group(3).split()))
            for m in re.finditer(r"typedef\s+([\w\s\*]+?)\(\s*\*\s*(\w+)\s*\)\s*\(([^;]*?)\)\s*;", text, flags=re.S)}


def _split_c_declarators(body: str):
    """`uint32_t a, b; const char *p; char s[32];` -> [(type, name, array)]"""
    out = []
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        if not stmt:
            continue
        first, *rest = [s.strip() for s in stmt.split(",")]
        m = re.match(r"^(.*?)(\**)\s*(\w+)\s*(?:\[(\w+)\])?$", first)
        base, stars, name, arr = m.group(1).strip(), m.group(2), m.group(3), m.group(4)
        out.append((base + stars, name, arr))
        for r in rest:
            m2 = re.match(r"^(\**)\s*(\w+)\s*(?:\[(\w+)\])?$", r)
            out.append((base + m2.group(1), m2.group(2), m2.group(3)))
    return out


def parse_c_header(path: str, extra_defines: dict | None = None):
    """-> (structs {name: [(field, class)]}, functions {name: (ret class, [arg classes])}, defines)"""
    raw = open(path).read()
    defines = dict(extra_defines or {})
    defines.update(_c_defines(raw))
    text = strip_c_comments(raw)
    defines["__fnptr__"] = set(defines.get("__fnptr__", ())) | set(fn_typedefs(text))
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        structs[m.group(2)] = [(name, c_class(t, arr, defines)) for t, name, arr in _split_c_declarators(m.group(1))]
    functions = {}
    for m in re.finditer(r"\bBH_API\s+([^;(]*?)(\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        arg_classes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"^(.*?)(\**)\s*(\w+)$", a)
                # `const float *const *segments`: everything before the last identifier is the type
                typ = (mm.group(1) + mm.group(2)).strip()
                if not typ:            # unnamed parameter
                    typ = a
                arg_classes.append(c_class(typ, None, defines))
        functions[name] = (c_class(ret, None, defines), arg_classes)
    return structs, functions, defines


RUST_FN_ALIASES: set = set()      # `pub type name = Option<unsafe extern "C" fn(...)>;` seen by parse_rust


def rust_class(t: str) -> str:
    t = " ".join(t.split())
    if t.startswith("*") or t in RUST_FN_ALIASES:
        return "ptr"
    m = re.match(r"^\[\s*(.+?)\s*;\s*(\d+)\s*\]$", t)
    if m:
        return f"{rust_class(m.group(1))}[{int(m.group(2))}]"
    if t.startswith("Option<") or t.startswith("extern") or t.startswith("unsafe extern"):
        return "ptr"
    if t not in RUST_SCALARS:
        raise ValueError(f"unknown Rust type {t!r}")
    return RUST_SCALARS[t]


def _split_top(s: str, sep: str = ","):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "[(<":
            depth += 1
        elif ch in "])>":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def parse_rust(text: str):
    """Every `#[repr(C)] ... struct Name { field: type, ... }` and every `fn name(arg: type, ...) [-> type];` inside an
    `extern "C" { }` block -> (structs {Name: [(field, class)]}, functions {name: (ret, [args])})."""
    text = re.sub(r"//[^\n]*", " ", text)
    for m in re.finditer(r"\btype\s+(\w+)\s*=\s*Option\s*<\s*(?:unsafe\s+)?extern", text):
        RUST_FN_ALIASES.add(m.group(1))
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[[^\]]*\])*\s*(?:pub\s+)?struct\s+(\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for f in _split_top(m.group(2)):
            f = f.strip()
            if not f:
                continue
            name, typ = f.split(":", 1)
            name = name.replace("pub ", "").strip()
            fields.append((name, rust_class(typ)))
        structs[m.group(1)] = fields
    functions = {}
    for blk in re.finditer(r'extern\s+"C"\s*\{(.*?)\n?\}', text, flags=re.S):
        for m in re.finditer(r"\bfn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", blk.group(1), flags=re.S):
            args = []
            for a in _split_top(m.group(2)):
                a = a.strip()
                if a:
                    args.append(rust_class(a.split(":", 1)[1]))
            functions[m.group(1)] = (rust_class(m.group(3)) if m.group(3) else "void", args)
    return structs, functions


def rust_struct_name(c_name: str) -> str:
    """bh_model_info -> BhModelInfo"""
    return "".join(p.capitalize() for p in c_name.split("_"))


def layout(fields):
    """natural-alignment C layout of [(name, class)] -> (size, [(name, offset)])"""
    off, align_max, out = 0, 1, []
    for name, cls in fields:
        m = re.match(r"^(\w+)\[(\d+)\]$", cls)
        base, n = (m.group(1), int(m.group(2))) if m else (cls, 1)
        sz = SIZES[base]
        off = (off + sz - 1) // sz * sz
        out.append((name, off))
        off += sz * n
        align_max = max(align_max, sz)
    return (off + align_max - 1) // align_max * align_max, out


def ctypes_class(t) -> str:
    import ctypes as C
    if t is None:
        return "void"
    if isinstance(t, type) and issubclass(t, C.Array):
        return f"{ctypes_class(t._type_)}[{t._length_}]"
    table = {C.c_char: "i8", C.c_int8: "i8", C.c_uint8: "u8", C.c_int16: "i16", C.c_uint16: "u16", C.c_int32: "i32", C.c_int: "i32",
             C.c_uint32: "u32", C.c_int64: "i64", C.c_uint64: "u64", C.c_size_t: "usize", C.c_float: "f32", C.c_double: "f64"}
    # c_size_t is c_ulong == c_uint64 on this platform: both spellings name one ctypes type
    if t in (C.c_char_p, C.c_void_p) or (isinstance(t, type) and issubclass(t, (C._Pointer, C._CFuncPtr))):
        return "ptr"
    if t in table:
        cls = table[t]
        return cls
    raise ValueError(f"unknown ctypes type {t!r}")

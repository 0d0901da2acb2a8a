#!/usr/bin/env python3
"""include/bp_msm_ntt.h -> include/bp_msm_ntt.rs: the complete `extern "C"` block of the C ABI for the reference's Rust crate
(INTEGRATION.md section 2 shows the subset the shims of sections 3-6 use).  No Rust toolchain exists in this image, so the file is
generated from the header's prototypes by a fixed type map and kept in step by tests/test_abi.py (which regenerates and compares).
    python tools/gen_rust_bindings.py [--check]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bp_msm_ntt.h")
OUT = os.path.join(ROOT, "include", "bp_msm_ntt.rs")

SCALARS = {"int": "c_int", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8", "float": "f32", "char": "c_char",
           "void": "c_void", "bp_ctx": "BpCtx"}


def rust_type(ctype):
    """`const uint8_t*`, `bp_ctx**`, `size_t`, `uint8_t[96]` (array parameters decay to pointers) ..."""
    t = ctype.strip()
    stars = t.count("*")
    t = t.replace("*", " ").strip()
    const = False
    words = []
    for w in t.split():
        if w == "const":
            const = True
        else:
            words.append(w)
    assert len(words) == 1, ctype
    base = SCALARS[words[0]]
    if stars == 0:
        assert base != "c_void", ctype
        return base
    r = base
    for level in range(stars):
        inner_const = const and level == 0
        r = ("*const " if inner_const else "*mut ") + r
    return r


def prototypes(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#"))
    for m in re.finditer(r"(?:^|\n)\s*((?:const\s+)?[A-Za-z_0-9]+(?:\s*\*)*)\s+(bp_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1), m.group(2), " ".join(m.group(3).split())
        args = []
        if params and params != "void":
            for p in params.split(","):
                p = p.strip()
                arr = re.match(r"(.*?)([A-Za-z_0-9]+)\s*\[\s*\d*\s*\]$", p)
                if arr:
                    ctype, pname = arr.group(1) + "*", arr.group(2)
                else:
                    mm = re.match(r"(.*?)([A-Za-z_0-9]+)$", p)
                    ctype, pname = mm.group(1), mm.group(2)
                if pname in ("in", "type", "ref", "box", "fn", "mod", "use", "move", "match", "loop", "impl", "self"):
                    pname += "_"
                args.append((pname, rust_type(ctype)))
        yield name, args, (None if ret.strip() == "void" else rust_type(ret))


def constants(text):
    plain = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for m in re.finditer(r"#define\s+(BP_[A-Z0-9_]+)\s+\(?(-?\d+)u?\)?", plain):
        yield m.group(1), int(m.group(2))
    for body in re.findall(r"enum\s*(?:[A-Za-z_0-9]+\s*)?\{([^}]*)\}", plain, flags=re.S):
        for item in body.split(","):
            mm = re.match(r"\s*(BP_[A-Z0-9_]+)\s*=\s*(-?\d+)\s*$", item)
            if mm:
                yield mm.group(1), int(mm.group(2))


def generate():
    text = open(HEADER).read()
    out = ["// Generated from include/bp_msm_ntt.h by tools/gen_rust_bindings.py -- do not edit; the header's comments are the documentation.",
           "// Drop into the reference crate as src/gpu.rs (`pub mod gpu;` in src/lib.rs) and link with `-l bp_msm_ntt` (INTEGRATION.md section 1).",
           "#![allow(dead_code)]",
           "use std::os::raw::{c_char, c_int, c_void};",
           "#[repr(C)]",
           "pub struct BpCtx {",
           "    _private: [u8; 0],",
           "}"]
    seen = set()
    for name, value in constants(text):
        if name in seen:
            continue
        seen.add(name)
        ty = "c_int" if value < 0 or value < 2**31 else "u64"
        if name in ("BP_SRS_TABLES_OFF",):
            ty = "u32"
        if name.endswith("_BYTES"):
            ty = "usize"
        out.append("pub const %s: %s = %d;" % (name, ty, value))
    out.append('extern "C" {')
    for name, args, ret in prototypes(text):
        sig = "    pub fn %s(%s)" % (name, ", ".join("%s: %s" % a for a in args))
        out.append(sig + (" -> %s;" % ret if ret else ";"))
    out.append("}")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    new = generate()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == new else 1)
    open(OUT, "w").write(new)
    print("wrote", OUT, "(%d functions)" % new.count("pub fn "))

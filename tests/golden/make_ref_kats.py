#!/usr/bin/env python3
"""Extract the literal known-answer vectors held by the reference's own unit tests.

Reads the reference's Rust test modules AS TEXT (no code is copied, compiled or run)
and writes the numeric literals -- inputs and expected outputs -- to ref_kats.json.
Run in the authoring container only (the reference does not exist on the GPU box):

    python tests/golden/make_ref_kats.py /root/reference

Each entry records the file:line of the test it came from.  Hex-limb arrays are kept
in source order; byte arrays (decimal lists) likewise.
"""
import json
import os
import re
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
SRC = os.path.join(ref, "lib/bls12_381/src")

WANT = {
    "fp.rs": ["test_squaring", "test_multiplication", "test_addition", "test_subtraction", "test_negation",
              "test_debug", "test_from_bytes", "test_sqrt", "test_inversion", "test_lexicographic_largest"],
    "scalar.rs": ["test_to_bytes", "test_from_bytes", "test_from_bytes_wide_r2", "test_from_bytes_wide_negative_one",
                  "test_from_bytes_wide_maximum", "test_addition", "test_from_raw", "test_debug"],
    "g1.rs": ["test_doubling", "test_projective_addition", "test_mixed_addition", "test_beta"],
}

hex_arr = re.compile(r"\[\s*((?:0x[0-9a-fA-F_]+\s*,?\s*)+)\]")
dec_arr = re.compile(r"\[\s*((?:\d+\s*,\s*)+\d+\s*,?\s*)\]")
strings = re.compile(r'"(0x[0-9a-f]+)"')

out = {}
for fname, tests in WANT.items():
    text = open(os.path.join(SRC, fname)).read()
    lines = text.split("\n")
    per = {}
    for t in tests:
        m = re.search(r"fn %s\(\)" % t, text)
        if not m:
            continue
        start = m.start()
        nxt = text.find("#[test]", start)
        body = text[start: nxt if nxt > 0 else len(text)]
        line_no = text[:start].count("\n") + 1
        hexes = [[int(x.replace("_", ""), 16) for x in re.findall(r"0x[0-9a-fA-F_]+", g)] for g in hex_arr.findall(body)]
        decs = [[int(x) for x in re.findall(r"\d+", g)] for g in dec_arr.findall(body)]
        decs = [d for d in decs if len(d) in (32, 48, 64, 96)]
        per[t] = {"source": "lib/bls12_381/src/%s:%d" % (fname, line_no),
                  "hex_arrays": [[hex(v) for v in h] for h in hexes],
                  "byte_arrays": decs,
                  "strings": strings.findall(body)}
    # the LARGEST constant used by the scalar tests (scalar.rs:1058-1063)
    if fname == "scalar.rs":
        m = re.search(r"const LARGEST: Scalar = Scalar\(\[(.*?)\]\)", text, re.S)
        per["LARGEST"] = {"source": "lib/bls12_381/src/scalar.rs:%d" % (text[:m.start()].count("\n") + 1),
                          "hex_arrays": [[hex(int(x.replace("_", ""), 16)) for x in re.findall(r"0x[0-9a-fA-F_]+", m.group(1))]],
                          "byte_arrays": [], "strings": []}
    out[fname] = per

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_kats.json")
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst, {k: list(v) for k, v in out.items()})

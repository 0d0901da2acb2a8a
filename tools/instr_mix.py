#!/usr/bin/env python3
"""Static instruction mix of the bucket-accumulation loop (msm_accumulate<2>), from the device assembly of the shipped source.

  python tools/instr_mix.py [--out profiles/r04_msm_accumulate_instr_mix.json]

Compiles baby_plonk_rust_amd/csrc/msm.hip for gfx950 to assembly (device side only, same flags as the Makefile), finds the
kernel, takes its hottest loop = the back-edge span with the most v_mad_u64_u32 (one iteration = one mixed bucket addition
with the software-pipelined gather of the next point) and counts the VALU instructions of one iteration by issue class:
  quarter_rate  v_mad_u64_u32 and the other integer ops that tools/ubench_int.hip measured at ~57 lanes/clk/CU
                (64-bit shifts / adds, 32-bit multiplies, carry adds)
  full_rate     every other VALU instruction (~90 lanes/clk/CU)
bench.py prices `roofline_valu_issue` with these counts instead of a hand-typed constant."""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUARTER = ("v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_lshl_add_u64", "v_lshlrev_b64", "v_lshrrev_b64",
           "v_ashrrev_i64", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_subbrev_co_u32",
           "v_add_co_ci_u32", "v_sub_co_ci_u32")


def kernel_body(asm, name):
    start = None
    lines = asm.splitlines()
    for i, l in enumerate(lines):
        if l.startswith("_ZN2bp") and name in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]) and ":" in l and "@" in l:
            start = i
            break
    if start is None:
        raise SystemExit("kernel %s not found" % name)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[start:end + 1]


def hottest_loop(body):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, l in enumerate(body):
        m = re.match(r"^\s*s_cbranch_\w+\s+(\.LBB[0-9_]+)", l) or re.match(r"^\s*s_branch\s+(\.LBB[0-9_]+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = body[labels[m.group(1)]:i + 1]
            mads = sum(1 for s in span if s.strip().startswith("v_mad_u64_u32"))
            if best is None or mads > best[0]:
                best = (mads, span)
    if best is None:
        raise SystemExit("no loop found")
    return best[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_msm_accumulate_instr_mix.json"))
    ap.add_argument("--asm", default=None, help="reuse an existing device assembly file")
    args = ap.parse_args()
    src = os.path.join(ROOT, "baby_plonk_rust_amd", "csrc", "msm.hip")
    if args.asm:
        asm = open(args.asm).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "msm.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", out],
                                  stderr=subprocess.DEVNULL)
            asm = open(out).read()
    result = {}
    for waves in (2,):
        body = kernel_body(asm, "msm_accumulateILi%dE" % waves)
        loop = hottest_loop(body)
        counts = collections.Counter()
        for l in loop:
            t = l.strip().split()
            if t and t[0].startswith("v_"):
                counts[t[0].replace("_e32", "").replace("_e64", "")] += 1
        valu = sum(counts.values())
        quarter = sum(v for k, v in counts.items() if k in QUARTER)
        vgpr = next((int(m.group(1)) for m in (re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", l) for l in asm.splitlines()) if m), None)
        result["msm_accumulate<%d>" % waves] = {
            "valu": valu, "quarter_rate": quarter, "full_rate": valu - quarter, "v_mad_u64_u32": counts["v_mad_u64_u32"],
            "loop_instructions_total": sum(1 for l in loop if l.strip() and not l.strip().startswith((";", ".")) and not re.match(r"^\.LBB", l)),
            "top": dict(counts.most_common(14)),
            "memory": {"global_load": sum(1 for l in loop if l.strip().startswith(("global_load", "buffer_load"))),
                       "global_store": sum(1 for l in loop if l.strip().startswith(("global_store", "buffer_store")))},
        }
    result["source"] = "hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S baby_plonk_rust_amd/csrc/msm.hip; hottest back-edge span of the kernel"
    result["note"] = ("a static count of the whole span between the loop label and its back edge: it includes whatever the compiler lays out "
                      "inside it -- the bucket-boundary block (flush + next bucket: ~100 v_mov, the stores, the offset loads), which a wave "
                      "runs in about one of eight iterations at 2^20 points.  With that block outside the span (as an earlier build of the same "
                      "loop had it) the count is 5 022 = 4 288 + 734; the roofline priced from this file is therefore an upper bound on the work "
                      "per addition, by ~2 %.  The register-only microbenchmark (tools/ubench_g1add.hip) is the layout-independent reference.")
    result["quarter_rate_classes"] = list(QUARTER)
    with open(args.out, "w") as f:
        json.dump(result, f, indent=1)
    print(json.dumps(result["msm_accumulate<2>"], indent=1))


if __name__ == "__main__":
    main()

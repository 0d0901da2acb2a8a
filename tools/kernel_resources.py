#!/usr/bin/env python3
"""Per-kernel register / spill / scratch / LDS figures of the SHIPPED library's code object (the AMDGPU metadata note of
baby_plonk_rust_amd/libbp_msm_ntt.so, or of the .so given), as a table; --check fails if any kernel spills or uses scratch.

  python tools/kernel_resources.py [--so PATH] [--filter msm_] [--check] [--out profiles/r05_kernel_resources.txt]"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(so, tmp):
    """the gfx950 code objects embedded in the host library: .hip_fatbin holds one clang offload bundle per translation unit"""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    out = []
    for i, a in enumerate(starts):
        piece = os.path.join(tmp, "bundle%d.bin" % i)
        open(piece, "wb").write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        targets = subprocess.check_output([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", "--input=" + piece], text=True).split()
        tgt = [t for t in targets if "gfx950" in t]
        if not tgt:
            continue
        co = os.path.join(tmp, "dev%d.co" % i)
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=" + tgt[0], "--input=" + piece, "--output=" + co])
        out.append(co)
    return out


def kernels(co):
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    out = []
    for block in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        def field(name, default="0"):
            m = re.search(r"\." + name + r":\s*(\S+)", block)
            return m.group(1) if m else default
        sym = field("name", "?")
        try:
            sym = subprocess.check_output([os.path.join(LLVM, "llvm-cxxfilt"), sym], text=True).strip()
        except Exception:
            pass
        out.append({"name": re.sub(r"\(.*", "", sym), "vgpr": int(field("vgpr_count")), "sgpr": int(field("sgpr_count")),
                    "vgpr_spill": int(field("vgpr_spill_count")), "sgpr_spill": int(field("sgpr_spill_count")),
                    "scratch": int(field("private_segment_fixed_size")), "lds": int(field("group_segment_fixed_size")),
                    "max_wg": int(field("max_flat_workgroup_size"))})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=os.path.join(ROOT, "baby_plonk_rust_amd", "libbp_msm_ntt.so"))
    ap.add_argument("--filter", default="")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--out")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        ks = [k for co in code_objects(args.so, tmp) for k in kernels(co) if args.filter in k["name"]]
    ks.sort(key=lambda k: k["name"])
    lines = ["%-72s %5s %5s %7s %7s %8s %7s" % ("kernel (" + os.path.basename(args.so) + ")", "vgpr", "sgpr", "v.spill", "s.spill", "scratchB", "ldsB")]
    for k in ks:
        lines.append("%-72s %5d %5d %7d %7d %8d %7d" % (k["name"][:72], k["vgpr"], k["sgpr"], k["vgpr_spill"], k["sgpr_spill"], k["scratch"], k["lds"]))
    bad = [k["name"] for k in ks if k["vgpr_spill"] or k["scratch"]]
    lines.append("%d kernels; spilling or using scratch: %s" % (len(ks), ", ".join(bad) if bad else "none"))
    text = "\n".join(lines)
    print(text)
    if args.out:
        open(args.out, "w").write(text + "\n")
    if args.check and bad:
        sys.exit(1)


if __name__ == "__main__":
    main()

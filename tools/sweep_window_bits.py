#!/usr/bin/env python3
"""A/B of the fixed-base window width: device time of one MSM per table width at one size (VERDICT r01 next #4a)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import baby_plonk_rust_amd as bp

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--widths", type=int, nargs="*", default=[16, 18, 19, 20, 22])
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
ctx = bp.Context(0)
n = 1 << args.log_n
srs = ctx.srs_generate_progression(n, 12345, 67891)
sc = torch.empty(n * 4, dtype=torch.int64, device="cuda")
ctx.synthetic_scalars_device(sc.data_ptr(), n, 0x5EED)
ref = None
for c in args.widths:
    info = ctx.srs_precompute(srs, c)
    best = None
    for _ in range(args.reps):
        out = ctx.msm_partial(srs, None, device_ptr=sc.data_ptr(), n=n)
        st = ctx.msm_stats()
        if best is None or st["device_ms"] < best["device_ms"]:
            best = st
    ref = ref or out
    assert out[:144] == ref[:144] or bp.sum_partials(out) == bp.sum_partials(ref)
    print(json.dumps({"log_n": args.log_n, "window_bits": c, "windows": info["windows"], "table_GiB": round(info["bytes"] / 2**30, 2),
                      "device_ms": round(best["device_ms"], 3), "accumulate_ms": round(best["accumulate_ms"], 3),
                      "other_ms": round(best["device_ms"] - best["accumulate_ms"], 3), "additions": best["mixed_adds"]}), flush=True)

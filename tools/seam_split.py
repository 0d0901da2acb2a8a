#!/usr/bin/env python3
"""where the literal bucket_msm(&[G1Projective], &[Scalar]) seam spends its time: load (upload + normalise + 28-bit copy), multiply, free"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import baby_plonk_rust_amd as bp
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
n = 1 << args.log_n
ctx = bp.default_context()
h0 = ctx.srs_generate_progression(n, 12345, 6789)
images = ctx.srs_export_projective144(h0)
sc = O.splitmix_scalars(n, 99)
want = ctx.msm(h0, sc)
for i in range(args.reps + 1):
    t0 = time.perf_counter()
    r = ctx.msm_projective144(images, sc)
    t1 = time.perf_counter()
    assert r == want
    if i:
        print("2^%d: one call (pieces, multiply behind the upload) %.2f ms" % (args.log_n, 1e3 * (t1 - t0)), flush=True)
for i in range(args.reps + 1):
    t0 = time.perf_counter()
    h = ctx.srs_load_projective144(images)
    t1 = time.perf_counter()
    r = ctx.msm(h, sc)
    t2 = time.perf_counter()
    ctx.srs_free(h)
    t3 = time.perf_counter()
    assert r == want
    if i:
        print("2^%d: load %.2f ms  multiply %.2f ms (device %.2f)  free %.2f ms  total %.2f ms" % (args.log_n, 1e3 * (t1 - t0), 1e3 * (t2 - t1), ctx.msm_stats()["device_ms"],
                                                                                                  1e3 * (t3 - t2), 1e3 * (t3 - t0)), flush=True)

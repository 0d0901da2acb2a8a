#!/usr/bin/env python3
"""run K MSMs (and optionally NTTs) of a given size on cuda:0 -- a small target for rocprofv3"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import baby_plonk_rust_amd as bp

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--ntt-log-n", type=int, default=0)
ap.add_argument("--tables", type=int, default=-1, help="fixed-base table window bits (0 = auto, -1 = no tables)")
args = ap.parse_args()
ctx = bp.Context(0)
n = 1 << args.log_n
srs = ctx.srs_generate_progression(n, 12345, 67891)
if args.tables >= 0:
    t0 = time.perf_counter()
    info = ctx.srs_precompute(srs, args.tables)
    print("tables: %.1f ms, %s" % (1e3 * (time.perf_counter() - t0), info), flush=True)
sc = torch.empty(n * 4, dtype=torch.int64, device="cuda")
ctx.synthetic_scalars_device(sc.data_ptr(), n, 0x5EED)
for i in range(args.reps):
    t0 = time.perf_counter()
    ctx.msm_partial(srs, None, device_ptr=sc.data_ptr(), n=n)
    st = ctx.msm_stats()
    print("msm 2^%d: wall %.3f ms, device %.3f ms, accumulate %.3f ms, c=%d tables=%s" % (
        args.log_n, 1e3 * (time.perf_counter() - t0), st["device_ms"], st["accumulate_ms"], st["window_bits"], st["tables"]), flush=True)
# PCIe-inclusive: the caller hands pageable host scalars (bp_msm_g1 through the reference-shaped seam)
host = sc.cpu().numpy().view("uint64").reshape(n, 4)
for i in range(args.reps):
    t0 = time.perf_counter()
    ctx.msm_partial(srs, host)
    print("msm 2^%d from host scalars: wall %.3f ms (device part %.3f ms)" % (
        args.log_n, 1e3 * (time.perf_counter() - t0), ctx.msm_stats()["device_ms"]), flush=True)
if args.ntt_log_n:
    nn = 1 << args.ntt_log_n
    v = torch.empty(nn * 4, dtype=torch.int64, device="cuda")
    ctx.synthetic_scalars_device(v.data_ptr(), nn, 0xF4)
    for i in range(args.reps):
        ctx.ntt_device(v.data_ptr(), args.ntt_log_n)
        print("ntt 2^%d: device %.3f ms" % (args.ntt_log_n, ctx.ntt_stats()["device_ms"]), flush=True)

#!/usr/bin/env python3
"""A/B of the commitments of a prover round (three 2^log_n-coefficient polynomials): one at a time, concurrent lanes, one batched
pipeline (BP_COMMIT_BATCH=1), on one device and through a group context of `--members` shards on this card."""
import argparse
import os
os.environ.setdefault("BABY_PLONK_LIBRARY", "exp")        # this tool sets BP_* knobs: only the experiment build reads them (make -C baby_plonk_rust_amd/csrc exp)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import baby_plonk_rust_amd as bp

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--members", type=int, default=0)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
ctx = bp.Context([0] * args.members) if args.members else bp.Context(0)
n = 1 << args.log_n
setup = bp.Setup.generate_srs(n + 6, 0x1234, ctx)
t = torch.empty((3, n + 2, 4), dtype=torch.int64, device="cuda")
for j in range(3):
    ctx.synthetic_scalars_device(t[j].data_ptr(), n + 2, 0x77 + j)
torch.cuda.synchronize()
polys = [bp.DevicePolynomial(t[j], bp.BASIS_MONOMIAL, ctx) for j in range(3)]
want = [bp.commit_device(setup, p) for p in polys]
for mode in ("one_by_one", "lanes_or_slots", "batched"):
    os.environ["BP_COMMIT_BATCH"] = "1" if mode == "batched" else "0"
    best = None
    for _ in range(args.reps):
        t0 = time.perf_counter()
        got = [bp.commit_device(setup, p) for p in polys] if mode == "one_by_one" else bp.commit_many_device(setup, polys)
        dt = time.perf_counter() - t0
        assert got == want
        best = dt if best is None or dt < best else best
    print("2^%d x 3 commitments, %s, %s: %.3f ms" % (args.log_n, "%d members" % args.members if args.members else "one device", mode, 1e3 * best), flush=True)

#!/usr/bin/env python3
"""ms per column of a BATCH of independent 2^log_n-element NTTs in one call (what the prover issues and the north star shards) -- the target
of the pass-split A/B of VERDICT r05 #5:  BABY_PLONK_LIBRARY=exp BP_NTT_SPLIT="20:10,10" python tools/ntt_batch_ab.py --log-n 20 --batch 8"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import baby_plonk_rust_amd as bp

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
ctx = bp.Context(0)
n = 1 << args.log_n
cols = torch.empty((args.batch, n, 4), dtype=torch.int64, device="cuda")
for j in range(args.batch):
    ctx.synthetic_scalars_device(cols[j].data_ptr(), n, 0xBA7C0000 + 31 * j)
one = cols[args.batch // 2].clone()
torch.cuda.synchronize()
ctx.ntt_device(one.data_ptr(), args.log_n)
ctx.ntt_device(cols.data_ptr(), args.log_n, batch=args.batch)
torch.cuda.synchronize()
assert torch.equal(cols[args.batch // 2], one), "batched transform differs from the single one"
best = None
for rep in range(3):
    for _ in range(3):
        ctx.ntt_device_async(cols.data_ptr(), args.log_n, batch=args.batch)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.ntt_device_async(cols.data_ptr(), args.log_n, batch=args.batch)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / (args.steps * args.batch)
    best = dt if best is None or dt < best else best
print("split %s: 2^%d x %d columns: %.4f ms per column (best of 3), kernels of one batch %.3f ms, passes %d" % (
    os.environ.get("BP_NTT_SPLIT", "default"), args.log_n, args.batch, 1e3 * best, ctx.ntt_stats()["device_ms"], ctx.ntt_stats()["passes"]), flush=True)

#!/usr/bin/env python3
"""MSM time at 2^log_n points for skewed scalar distributions (the balanced-chunk accumulate should not care; the long-bucket
fix-up path must not explode), with and without the SRS's tables."""
import argparse, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import baby_plonk_rust_amd as bp
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--window-bits", type=int, default=0, help="table width (0 = auto)")
ap.add_argument("--only", default="", help="one distribution only, with tables only (a target for rocprofv3)")
args = ap.parse_args()
n = 1 << args.log_n
Q = O.Q
rnd = random.Random(1)
ctx = bp.default_context()
h = ctx.srs_generate_progression(n, 12345, 67891)
base = O.splitmix_scalars(n, 7)
kinds = {}
kinds["random"] = base
kinds["all_equal"] = np.repeat(bp.scalar_from_int(rnd.randrange(Q))[None, :], n, axis=0)
small8 = np.zeros((n, 4), dtype=np.uint64); small8[:, 0] = np.random.default_rng(1).integers(0, 256, n)
small16 = np.zeros((n, 4), dtype=np.uint64); small16[:, 0] = np.random.default_rng(2).integers(0, 65536, n)
raw = np.zeros((n, 32), dtype=np.uint8)
def to_mont(vals_u64_col0):
    out = np.zeros((n, 4), dtype=np.uint64)
    b = np.zeros((n, 4), dtype=np.uint64); b[:, 0] = vals_u64_col0
    from baby_plonk_rust_amd import _lib
    assert _lib.load().bp_fr_convert(b.view(np.uint8).ctypes.data, n, 0, 1, out.ctypes.data) == 0
    return out
kinds["small_8bit"] = to_mont(small8[:, 0])
kinds["small_16bit"] = to_mont(small16[:, 0])
two = np.zeros(n, dtype=np.uint64); two[::2] = 1
kinds["zeros_and_ones"] = to_mont(two)
sparse = np.zeros((n, 4), dtype=np.uint64); idx = np.random.default_rng(3).integers(0, n, n // 100); sparse[idx] = base[idx]
kinds["sparse_1pct"] = sparse
if args.only:
    kinds = {args.only: kinds[args.only]}
for tables in ((True,) if args.only else (False, True)):
    if tables:
        print(ctx.srs_precompute(h, args.window_bits))
    for name, sc in kinds.items():
        t = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).cuda()
        torch.cuda.synchronize()
        best = None
        for _ in range(4):
            t0 = time.perf_counter()
            ctx.msm_partial(h, None, device_ptr=t.data_ptr(), n=n)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        st = ctx.msm_stats()
        print("tables=%-5s %-16s wall %.3f ms  device %.3f ms  accumulate %.3f ms" % (tables, name, 1e3 * best, st["device_ms"], st["accumulate_ms"]), flush=True)

#!/usr/bin/env python3
"""Host-to-host time of ONE 2^24-element bp_ntt_fr call (in place on a pageable numpy buffer: upload, transform, download) on a
single-GPU context and on group contexts; on a one-GPU box the group's members share the card and its PCIe link."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import baby_plonk_rust_amd as bp

LOG_N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
one = bp.Context(0)
import torch
t = torch.empty((1 << LOG_N) * 4, dtype=torch.int64, device="cuda:0")
one.synthetic_scalars_device(t.data_ptr(), 1 << LOG_N, 0x24)
x = t.cpu().numpy().view(np.uint64).reshape(-1, 4)
del t
ref = None
for name, ctx in (("one context", one), ("group {0,0}", bp.Context([0, 0])), ("group {0,0,0,0}", bp.Context([0] * 4)), ("group {0} x 8", bp.Context([0] * 8))):
    ts = []
    for rep in range(4):
        a = x.copy()
        t0 = time.perf_counter()
        ctx.check(ctx._lib.bp_ntt_fr(ctx._h, a.ctypes.data, LOG_N, 0, bp.FR_MONT, 1, 1 << LOG_N), "bp_ntt_fr")
        ts.append(time.perf_counter() - t0)
    st = ctx.ntt_stats()
    print("%-18s 2^%d bp_ntt_fr host to host: best %.1f ms, kernels %.2f ms, members %d" % (name, LOG_N, 1e3 * min(ts), st["device_ms"], st["members"]))
    if ref is None:
        ref = a
    else:
        assert (a == ref).all()

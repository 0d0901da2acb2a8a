#!/usr/bin/env python3
"""probe: how much do two MSMs gain from running concurrently on two streams (two contexts, two host threads)?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import baby_plonk_rust_amd as bp

n = 1 << 20
ctxs = [bp.Context(0), bp.Context(0)]
srs = [c.srs_generate_progression(n, 12345, 67891) for c in ctxs]
sc = torch.empty(n * 4, dtype=torch.int64, device="cuda")
ctxs[0].synthetic_scalars_device(sc.data_ptr(), n, 0x5EED)
torch.cuda.synchronize()

def run(i, reps):
    for _ in range(reps):
        ctxs[i].msm_partial(srs[i], None, device_ptr=sc.data_ptr(), n=n)

run(0, 2); run(1, 2)
t0 = time.perf_counter(); run(0, 8); t1 = time.perf_counter()
print("serial: %.3f ms per MSM" % (1e3 * (t1 - t0) / 8))
th = [threading.Thread(target=run, args=(i, 8)) for i in range(2)]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
t1 = time.perf_counter()
print("two streams concurrently: %.3f ms per MSM (16 MSMs in %.2f ms)" % (1e3 * (t1 - t0) / 16, 1e3 * (t1 - t0)))

import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import baby_plonk_rust_amd as bp
from oracle import oracle as O
one = bp.Context(0)
x = O.splitmix_scalars(1 << 24, 0x24)
for name, ctx in (("one context", one), ("group {0,0}", bp.Context([0, 0])), ("group {0,0,0,0}", bp.Context([0] * 4))):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); y = ctx.ntt(x); ts.append(time.perf_counter() - t0)
    print("%-18s 2^24 host-to-host ntt: best %.1f ms, kernels %.2f ms, members %d" % (name, 1e3 * min(ts), ctx.ntt_stats()["device_ms"], ctx.ntt_stats()["members"]))
    if name == "one context": ref = y
    else: assert (y == ref).all()

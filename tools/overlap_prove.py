#!/usr/bin/env python3
"""proofs/s with T concurrent provers on ONE GPU (each its own library context = own HIP stream, SRS tables and circuit):
does overlapping one proof's latency-bound tails with another proof's bulk kernels raise throughput?"""
import argparse
import os
import random
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import baby_plonk_rust_amd as bp
from baby_plonk_rust_amd.synthetic import Q, chained_multiplications

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--threads", type=int, nargs="+", default=[1, 2, 3])
args = ap.parse_args()
n = 1 << args.log_n
cols, pk = chained_multiplications(n, 7)
blinders = [random.Random(5).randrange(1, Q) for _ in range(11)]
wit = [torch.from_numpy(c.view(np.int64)).cuda() for c in cols]
torch.cuda.synchronize()
provers = []
for t in range(max(args.threads)):
    ctx = bp.Context(0)
    setup = bp.Setup.generate_srs(n + 6, 0x1234567, ctx)
    provers.append(bp.Prover(setup, bp.Circuit(pk, ctx)))
    provers[-1].prove_device(wit[0].data_ptr(), wit[1].data_ptr(), wit[2].data_ptr(), None, blinders)      # warm-up
for T in args.threads:
    def work(p):
        for _ in range(args.reps):
            p.prove_device(wit[0].data_ptr(), wit[1].data_ptr(), wit[2].data_ptr(), None, blinders)
    th = [threading.Thread(target=work, args=(provers[i],)) for i in range(T)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    print("threads %d: %.2f proofs/s (%.2f ms per proof per thread)" % (T, T * args.reps / dt, 1e3 * dt / args.reps), flush=True)

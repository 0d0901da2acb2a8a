#!/usr/bin/env python3
"""latency of one proof on a group context of `--members` shards of this card, with round 3 split by coset and without (the
members share one GPU here: this checks that the split costs nothing extra, not what it buys on separate GPUs)"""
import argparse
import os
os.environ.setdefault("BABY_PLONK_LIBRARY", "exp")        # this tool sets BP_* knobs: only the experiment build reads them (make -C baby_plonk_rust_amd/csrc exp)
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import baby_plonk_rust_amd as bp
from baby_plonk_rust_amd.synthetic import Q, chained_multiplications

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--members", type=int, default=4)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
n = 1 << args.log_n
cols, pk = chained_multiplications(n, 7)
blinders = [random.Random(5).randrange(1, Q) for _ in range(11)]
ref = None
for members in (1, args.members):
    ctx = bp.Context([0] * members) if members > 1 else bp.Context(0)
    setup = bp.Setup.generate_srs(n + 6, 0x1234567, ctx)
    prover = bp.Prover(setup, bp.Circuit(pk, ctx))
    wit = [torch.from_numpy(c.view(np.int64)).cuda() for c in cols]
    torch.cuda.synchronize()
    for split, early in ((("1", "1"), ("1", "0"), ("0", "0")) if members > 1 else (("1", "1"),)):
        os.environ["BP_PROVE_COSET_SPLIT"] = split
        os.environ["BP_PROVE_COSET_EARLY"] = early
        best = None
        for i in range(args.reps + 1):
            t0 = time.perf_counter()
            blob = prover.prove_device(wit[0].data_ptr(), wit[1].data_ptr(), wit[2].data_ptr(), None, blinders)
            dt = time.perf_counter() - t0
            if i:
                best = dt if best is None or dt < best else best
        ref = ref or blob
        assert blob == ref
        print("2^%d gates, %d member(s), coset split %s early %s: %.2f ms  rounds %s" % (args.log_n, members, split if members > 1 else "-", early if members > 1 else "-", 1e3 * best,
                                                                                   ["%.2f" % r for r in prover.last_stats()["round_ms"]]), flush=True)
    ctx.close()

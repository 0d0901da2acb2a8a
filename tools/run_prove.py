#!/usr/bin/env python3
"""K native proofs (bp_prove) of a synthetic 2^log_n-gate circuit on cuda:0 -- a small target for rocprofv3"""
import argparse
import os
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
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
n = 1 << args.log_n
ctx = bp.default_context()
cols, pk = chained_multiplications(n, 7)
setup = bp.Setup.generate_srs(n + 6, 0x1234567, ctx)
circuit = bp.Circuit(pk, ctx)
prover = bp.Prover(setup, circuit)
wit = [torch.from_numpy(c.view(np.int64)).cuda() for c in cols]
torch.cuda.synchronize()
blinders = [random.Random(5).randrange(1, Q) for _ in range(11)]
for i in range(args.reps + 1):
    t0 = time.perf_counter()
    blob = prover.prove_device(wit[0].data_ptr(), wit[1].data_ptr(), wit[2].data_ptr(), None, blinders)
    print("prove 2^%d: %.2f ms  rounds %s" % (args.log_n, 1e3 * (time.perf_counter() - t0),
                                              ["%.2f" % r for r in prover.last_stats()["round_ms"]]), flush=True)
import hashlib
print("proof sha256 %s" % hashlib.sha256(blob).hexdigest()[:16], flush=True)

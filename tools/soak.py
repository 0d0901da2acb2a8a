#!/usr/bin/env python3
"""Soak: the same proofs, commitments and transforms over and over for --seconds, on a plain context and on a rehearsal group ({0, 0, 0}: GPU-to-GPU
branches), every result compared with the first one -- an intermittent ordering bug shows up as a differing hash, a leak as growing device memory."""
import argparse
import hashlib
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
ap.add_argument("--seconds", type=float, default=300)
ap.add_argument("--log-n", type=int, default=16)
args = ap.parse_args()
n = 1 << args.log_n
cols, pk = chained_multiplications(n, 3)
blinders = [random.Random(11).randrange(1, Q) for _ in range(11)]
ctxs = {"one": bp.Context(0), "group3": bp.Context([0, 0, 0])}
provers, first = {}, {}
for name, ctx in ctxs.items():
    setup = bp.Setup.generate_srs(n + 6, 0xABCDEF, ctx)
    provers[name] = (bp.Prover(setup, bp.Circuit(pk, ctx)), setup)
wit = [torch.from_numpy(c.view(np.int64)).cuda() for c in cols]
x = torch.from_numpy(np.ascontiguousarray(cols[0]).view(np.int64)).cuda().reshape(-1).clone()
torch.cuda.synchronize()
ptr = [w.data_ptr() for w in wit]
t_end, rounds, mem0 = time.time() + args.seconds, 0, None
while time.time() < t_end:
    for name, (prover, setup) in provers.items():
        ctx = ctxs[name]
        h = hashlib.sha256()
        h.update(prover.prove_device(ptr[0], ptr[1], ptr[2], None, blinders))                       # witness in HBM
        h.update(prover.prove_with_blinding(cols[0], cols[1], cols[2], None, blinders))             # witness from the host
        h.update(ctx.msm(setup.handle, cols[1]))                                                     # one commitment from host scalars
        y = x.clone()
        torch.cuda.synchronize()
        ctx.ntt_device(y.data_ptr(), args.log_n)
        ctx.ntt_device(y.data_ptr(), args.log_n, inverse=True)
        torch.cuda.synchronize()
        h.update(b"1" if torch.equal(x, y) else b"0")
        d = h.hexdigest()
        if name not in first:
            first[name] = d
        assert d == first[name], "round %d: %s differs" % (rounds, name)
    assert first["one"] == first["group3"], "group context differs from the single-device one"
    rounds += 1
    if rounds == 20:
        mem0 = torch.cuda.mem_get_info()[0]
mem1 = torch.cuda.mem_get_info()[0]
print("soak ok: %d rounds of (2 proofs + commitment + NTT round trip) x 2 contexts at 2^%d gates in %.0f s; free device memory after round 20 / at the end: %s / %d MiB"
      % (rounds, args.log_n, args.seconds, "n/a" if mem0 is None else str(mem0 >> 20), mem1 >> 20))

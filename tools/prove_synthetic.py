#!/usr/bin/env python3
"""BASELINE.json configs[4] rehearsal: the reference prover's rounds 1-5 (restated in tests/prover_rounds.py) on a synthetic
n-gate circuit with every polynomial resident in HBM (DevicePolynomial), random SRS (tau known so the proof can be checked
in G1), Merlin challenges.  Prints one JSON line per size: seconds per proof, proofs/s, and whether the proof verifies.
Measurement script -- the prover logic lives in tests/ because the transcript / circuit front-end are out of the product's
scope; what is timed is the product's MSM / NTT / polynomial / grand-product kernels under the reference's call pattern."""
import argparse
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import baby_plonk_rust_amd as bp
from baby_plonk_rust_amd import _lib
from tests import bigint_model as M
from tests import prover_rounds as PR
from tests.test_gpu_prover_rounds import compute_challenges, decode, g1_only_verify, prove_with_blinding

Q = M.Q


def synthetic(n, seed):
    from baby_plonk_rust_amd.synthetic import chained_multiplications
    cols, pk = chained_multiplications(n, seed)
    return cols, pk, np.zeros((n, 4), dtype=np.uint64)


ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, nargs="+", default=[12, 16])
ap.add_argument("--reps", type=int, default=2)
args = ap.parse_args()
for k in args.log_n:
    n, tau = 1 << k, 0x1234567 + k
    t0 = time.perf_counter()
    cols, pk, public = synthetic(n, k)
    setup = bp.Setup.generate_srs(n + 6, tau)
    dev = PR.GpuDeviceBackend(setup)
    blinders = [random.Random(k).randrange(1, Q) for _ in range(11)]
    t_setup = time.perf_counter() - t0
    times = []
    for _ in range(args.reps):
        t1 = time.perf_counter()
        proof, ev, blob = prove_with_blinding(dev, n, cols, pk, public, blinders, logging=False)
        times.append(time.perf_counter() - t1)
    vk = {name: decode(dev.commit(dev.i_ntt_poly(dev.Polynomial(pk[name], dev.LAG)))) for name in pk}
    ok = g1_only_verify(n, tau, {name: decode(v) for name, v in proof.items()}, ev, compute_challenges(proof, ev), vk, [])
    print(json.dumps({"path": "reference call sequence over DevicePolynomial (tests/prover_rounds.py)", "gates": n,
                      "seconds_per_proof": min(times), "proofs_per_s": 1.0 / min(times), "verifies": bool(ok),
                      "proof_bytes": len(blob), "setup_seconds": t_setup,
                      "note": "includes uploading the witness / selector columns; SRS generation and circuit synthesis excluded"}), flush=True)
    # the native prover (bp_prove): circuit columns resident, witness resident, one call per proof
    import torch
    t2 = time.perf_counter()
    circuit = bp.Circuit(pk, setup.ctx)
    t_circuit = time.perf_counter() - t2
    prover = bp.Prover(setup, circuit)
    wit = [torch.from_numpy(c.view(np.int64)).cuda() for c in cols]
    torch.cuda.synchronize()
    ntimes, stats = [], None
    for _ in range(args.reps + 1):
        t1 = time.perf_counter()
        nblob = prover.prove_device(wit[0].data_ptr(), wit[1].data_ptr(), wit[2].data_ptr(), None, blinders)
        ntimes.append(time.perf_counter() - t1)
        stats = prover.last_stats()
    host_times = []
    for _ in range(args.reps):
        t1 = time.perf_counter()
        hblob = prover.prove_with_blinding(cols[0], cols[1], cols[2], None, blinders)
        host_times.append(time.perf_counter() - t1)
    print(json.dumps({"path": "native bp_prove", "gates": n, "seconds_per_proof": min(ntimes), "proofs_per_s": 1.0 / min(ntimes),
                      "same_bytes_as_reference_call_sequence": nblob == blob and hblob == blob, "round_ms": stats["round_ms"],
                      "seconds_per_proof_host_witness": min(host_times), "circuit_load_seconds": t_circuit,
                      "note": "witness columns resident in HBM (host-witness figure includes the 3 x n x 32 B upload)"}), flush=True)
    circuit.free()
    setup.ctx.srs_free(setup.handle)

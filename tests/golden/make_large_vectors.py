#!/usr/bin/env python3
"""Golden hashes at the sizes where the GPU path changes branches -> tests/golden/large_vectors.json  (VERDICT r01 next #3).

bp_prove has size-dependent branches the small fixtures never reach (two- and three-pass NTTs, the 4n coset of 2^14- and
2^16-gate circuits, chunked carry scans of the binomial division, fixed-base tables at their full width).  This script runs
the reference-shaped restatement of src/prover.rs (tests/prover_rounds.py) on the CPU oracle -- c = 4 bucket_msm as written,
the radix-2 twin of the reference DFT, Polynomial operators with the reference's Div semantics -- for synthetic circuits of
2^12, 2^14 and 2^16 gates and records only the sha256 of the 624 proof bytes (plus each commitment's) together with the
recipe that regenerates the inputs; likewise one 2^16-coefficient Polynomial division by x^n - 1 and by x - zeta.
The reference itself is Rust and cannot run here; nothing of it is copied, the fixture is data.

    python tests/golden/make_large_vectors.py          (2^12, 2^14, 2^16 gates + the divisions: three minutes on 8 cores)
    python tests/golden/make_large_vectors.py 18 19    (adds 2^18 and 2^19 gates -- 4n = 2^20 and 2^21, the three-pass NTT: ~40 minutes)
"""
import hashlib
import json
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import oracle as O
from tests import bigint_model as M
from tests import prover_rounds as PR
from tests.test_gpu_prover_rounds import prove_with_blinding, synthetic_circuit

Q = M.Q
sha = lambda b: hashlib.sha256(bytes(b)).hexdigest()


def srs_powers(count, tau):
    """[tau^i G] as projective points (setup.rs:12-31), one scalar multiplication per power like the reference"""
    g, out, s = O.g1_generator(), np.zeros((count, 18), dtype=np.uint64), 1
    for i in range(count):
        out[i] = O.g1_mul(g, O.fr_from_int(s))
        s = s * tau % Q
    return out


def large_proof(log_n):
    n = 1 << log_n
    seed, tau = 7000 + log_n, 0xABCDEF0000 + log_n
    cols, pk, public = synthetic_circuit(n, seed)
    blinders = [random.Random(100 + log_n).randrange(1, Q) for _ in range(11)]
    t0 = time.time()
    cpu = PR.OracleBackend(srs_powers(n + 6, tau))
    cpu.threads = os.cpu_count() or 1
    proof, ev, blob = prove_with_blinding(cpu, n, cols, pk, public, blinders, logging=False)
    print("2^%d gates: %.0f s" % (log_n, time.time() - t0), flush=True)
    return {"log_n": log_n, "circuit": "tests.test_gpu_prover_rounds.synthetic_circuit(2^log_n, seed)", "seed": seed, "tau": tau,
            "srs_powers": n + 6, "blinders": "random.Random(100 + log_n).randrange(1, Q) x 11", "proof_sha256": sha(blob),
            "commitment_sha256": {k: sha(v) for k, v in proof.items()}, "evaluations": {k: "%064x" % v for k, v in ev.items()}}


def large_division():
    """Polynomial / Polynomial (polynomial.rs:314-380) on 2^16 + 11 random coefficients: by x^n - 1 (n = 2^14: the round-3 shape,
    remainder dropped) and by x - zeta (the round-5 shape); inputs are O.splitmix_scalars streams, outputs hashed as the
    [len, 4] little-endian u64 Montgomery limbs the C ABI returns (BP_FR_MONT)"""
    na, n = (1 << 16) + 11, 1 << 14
    a = O.splitmix_scalars(na, 0xD1F1)
    zh = PR.sparse(n + 1, {0: Q - 1, n: 1})
    zeta = 0x1234567890ABCDEF1234567890ABCDEF % Q
    lin = PR.sparse(2, {0: Q - zeta, 1: 1})
    q1 = O.poly_binop("poly_div", a, zh)
    q2 = O.poly_binop("poly_div", a, lin)
    # independent check on a slice: the top coefficients of a / (x^n - 1) are those of a itself
    assert (q1[-5:] == a[-5:]).all() and len(q1) == na - n and len(q2) == na - 1
    return {"dividend": "O.splitmix_scalars(2^16 + 11, 0xD1F1)", "n": n, "zeta": "%x" % zeta,
            "by_xn_minus_1": {"len": len(q1), "sha256": sha(q1.tobytes())}, "by_x_minus_zeta": {"len": len(q2), "sha256": sha(q2.tobytes())}}


if __name__ == "__main__":
    path = os.path.join(HERE, "large_vectors.json")
    if len(sys.argv) > 1:            # python make_large_vectors.py 18 19  -> add (or refresh) proofs of these sizes, keep the rest
        out = json.load(open(path))
        have = {p["log_n"]: p for p in out["proofs"]}
        for k in map(int, sys.argv[1:]):
            have[k] = large_proof(k)
        out["proofs"] = [have[k] for k in sorted(have)]
    else:
        out = {"_made_by": "tests/golden/make_large_vectors.py (CPU oracle restatement of src/prover.rs / src/polynomial.rs)",
               "proofs": [large_proof(k) for k in (12, 14, 16)], "division": large_division()}
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote large_vectors.json")

"""CPU: the restated prover rounds (tests/prover_rounds.py) on the oracle backend produce a proof for the reference's toy
circuit that satisfies the verifier's final equation -- a protocol-level consistency check of the oracle's Polynomial,
DFT and MSM restatements (src/verifier.rs:80-192 in its G1-only form for known tau)."""
import random

import numpy as np

from oracle import oracle as O
from tests import bigint_model as M
from tests import prover_rounds as PR
from tests.test_gpu_prover_rounds import decode, g1_only_verify, run_rounds, toy_circuit

Q = M.Q


def oracle_srs(powers, tau):
    """setup.rs:12-31: sequential cur *= tau"""
    pts, cur, t = [], O.g1_generator(), O.fr_from_int(tau)
    for _ in range(powers):
        pts.append(cur)
        cur = O.g1_mul(cur, t)
    return np.stack(pts)


def test_toy_circuit_proof_verifies_on_the_oracle():
    n, tau = 8, 101
    cols, pk, public = toy_circuit(n)
    rnd = random.Random(7)
    blinders = [rnd.randrange(1, Q) for _ in range(11)]
    ch = {k: rnd.randrange(1, Q) for k in ("beta", "gamma", "alpha", "zeta", "nu", "mu")}
    cpu = PR.OracleBackend(oracle_srs(n + 6, tau))
    st, proof, ev = run_rounds(cpu, n, cols, pk, public, blinders, ch, lambda *a: O.round2_z(*a))
    vk = {k: decode(cpu.commit(cpu.Polynomial(cpu.i_ntt_381(PR.SV(pk[k])), cpu.MONO))) for k in pk}
    pts = {k: decode(v) for k, v in proof.items()}
    assert g1_only_verify(n, tau, pts, ev, ch, vk, [80])
    assert not g1_only_verify(n, tau, pts, ev, dict(ch, nu=(ch["nu"] + 1) % Q), vk, [80])
    # a wrong witness (c != a*b + b) makes the quotient inexact: the prover's own assert (prover.rs:615) fires
    bad_cols = [list(c) for c in cols]
    bad_cols[2][1] = 17
    try:
        run_rounds(cpu, n, bad_cols, pk, public, blinders, ch, lambda *a: O.round2_z(*a))
        raised = False
    except AssertionError:
        raised = True
    assert raised


def test_toy_circuit_full_proof_with_merlin_challenges_on_the_oracle():
    from tests.test_gpu_prover_rounds import compute_challenges, prove_with_blinding
    n, tau = 8, 101
    cols, pk, public = toy_circuit(n)
    blinders = [random.Random(99).randrange(1, Q) for _ in range(11)]
    cpu = PR.OracleBackend(oracle_srs(n + 6, tau))
    proof, ev, blob = prove_with_blinding(cpu, n, cols, pk, public, blinders, lambda *a: O.round2_z(*a))
    assert len(blob) == 624
    vk = {k: decode(cpu.commit(cpu.Polynomial(cpu.i_ntt_381(PR.SV(pk[k])), cpu.MONO))) for k in pk}
    ch = compute_challenges(proof, ev)
    assert g1_only_verify(n, tau, {k: decode(v) for k, v in proof.items()}, ev, ch, vk, [80])
    # golden: the proof bytes for these blinders are pinned (regression guard for every layer below)
    import hashlib
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "toy_proof_blinders_seed99.sha256")
    digest = hashlib.sha256(blob).hexdigest()
    if os.path.exists(path):
        assert open(path).read().strip() == digest
    else:                                       # first run in the authoring container writes the fixture
        open(path, "w").write(digest + "\n")

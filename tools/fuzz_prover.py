#!/usr/bin/env python3
"""Randomised differential run of bp_prove: random satisfiable circuits (random gates over a pool of shared variables,
optional public inputs, random group order 8..128, random tau, random blinders) -- the native proof bytes must equal the
reference's call sequence (tests/prover_rounds.py) run on the CPU oracle, and the G1-only verifier must accept.
    python tools/fuzz_prover.py --seconds 200 --seed 1"""
import argparse
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import baby_plonk_rust_amd as bp
from oracle import oracle as O
from tests import bigint_model as M
from tests import prover_rounds as PR
from tests.test_gpu_prover_rounds import decode, g1_only_verify, prove_with_blinding
from tests.test_native_prover import _challenges, _circuit_from_rows, _split

Q = M.Q
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rnd = random.Random(args.seed)
t_end, done = time.time() + args.seconds, 0
t_note = time.time() + 60
while time.time() < t_end:
    if time.time() > t_note:
        print("... %d circuits" % done, flush=True)
        t_note = time.time() + 60
    n = rnd.choice([8, 8, 16, 32, 64, 128])
    n_pub = rnd.choice([0, 0, 1, 2])
    n_gates = rnd.randrange(1, n - n_pub + 1)
    small = rnd.random() < 0.4
    val = lambda: rnd.randrange(1 << 8) if small else rnd.randrange(Q)
    values, names = {}, []

    def new_var(v):
        name = "v%d" % len(values)
        values[name] = v % Q
        names.append(name)
        return name

    wires, sel = [], dict(ql=[], qr=[], qm=[], qo=[], qc=[])
    publics = []
    for _ in range(n_pub):                        # "x public": row (x, -, -), ql = 1, PI = -x
        x = new_var(val())
        publics.append(values[x])
        wires.append((x, None, None))
        for k, v in (("ql", 1), ("qr", 0), ("qm", 0), ("qo", 0), ("qc", 0)):
            sel[k].append(v)
    for _ in range(n_gates):
        a = rnd.choice(names) if names and rnd.random() < 0.6 else new_var(val())
        b = rnd.choice(names) if rnd.random() < 0.6 else new_var(val())
        ql, qr, qm, qc = (rnd.choice([0, 1, Q - 1, val()]) for _ in range(4))
        c = new_var(ql * values[a] + qr * values[b] + qm * values[a] * values[b] + qc)      # qo = -1
        wires.append((a, b, c))
        for k, v in (("ql", ql), ("qr", qr), ("qm", qm), ("qo", Q - 1), ("qc", qc)):
            sel[k].append(v)
    rows, pk = _circuit_from_rows(wires, sel, n)
    cols = [[values[r[j]] if r[j] else 0 for r in rows] for j in range(3)]
    public = [(-x) % Q for x in publics] + [0] * (n - n_pub)
    tau = rnd.choice([1, 2, rnd.randrange(Q)])
    blinders = [rnd.randrange(Q) for _ in range(11)]
    setup = bp.Setup.generate_srs(n + 6, tau, tables=rnd.random() < 0.7)
    circuit = bp.Circuit({k: PR.SV(v) for k, v in pk.items()})
    blob = bp.Prover(setup, circuit).prove_with_blinding(PR.SV(cols[0]), PR.SV(cols[1]), PR.SV(cols[2]), PR.SV(public), blinders)
    cpu = PR.OracleBackend(O.proj_from_bytes96(setup.powers_of_x()))
    cpu.threads = 8
    want = prove_with_blinding(cpu, n, cols, pk, public, blinders, logging=False)[2]
    case = dict(n=n, n_pub=n_pub, n_gates=n_gates, small=small, tau=tau, seed=args.seed, case=done)
    if blob != want:
        print("PROOF MISMATCH", case)
        sys.exit(1)
    vk = {k: decode(v) for k, v in circuit.commitments(setup).items()}
    if not g1_only_verify(n, tau, *_split(blob), _challenges(blob), vk, publics):
        print("VERIFIER REJECTS", case)
        sys.exit(1)
    circuit.free()
    setup.ctx.srs_free(setup.handle)
    done += 1
print("prover fuzz ok: %d random circuits, seed %d" % (done, args.seed))

#!/usr/bin/env python3
"""Golden vectors of the hot path in the C-ABI wire formats -> tests/golden/path_vectors.json.

Made in the authoring container by the CPU oracle (oracle/, the restatement of the reference's algorithms), with every
group element and every DFT additionally produced by the independent Python big-int model (tests/bigint_model.py) and
required to agree before anything is written.  The reference itself is Rust and cannot run here; these
fixtures are what travels to the GPU box.  Inputs are small and explicit (hex), outputs are explicit bytes.

    python tests/golden/make_path_vectors.py
"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import oracle as O
from tests import bigint_model as M
from tests import prover_rounds as PR

Q = M.Q
out = {"_made_by": "tests/golden/make_path_vectors.py (oracle + big-int model, CPU)"}
le = lambda v: (v % Q).to_bytes(32, "little").hex()

# ---- MSM: the reference's own 1000-point fixture (i*G) as SRS, seeded scalars; 96-byte affine result ----
uncomp = open(os.path.join(HERE, "g1_uncompressed_valid_test_vectors.dat"), "rb").read()
pts = O.proj_from_bytes96(uncomp)
msm = []
for seed, n in ((1, 1), (2, 2), (3, 17), (4, 256), (5, 1000)):
    rnd = random.Random(seed)
    sc = [rnd.randrange(Q) for _ in range(n)]
    got = O.g1_bytes96(O.bucket_msm(pts[:n].copy(), O.fr_array_from_ints(sc), 256, 4, threads=8))
    assert got == M.enc96(M.ec_mul(sum(i * s for i, s in enumerate(sc)) % Q))
    msm.append({"points": "first %d of g1_uncompressed_valid_test_vectors.dat" % n, "n": n, "scalars_le32": [le(s) for s in sc] if n <= 17 else None,
                "scalars_python_random_seed": seed, "result96": got.hex()})
edge = [0, 1, Q - 1, 2**254, 5]
got = O.g1_bytes96(O.bucket_msm(pts[:5].copy(), O.fr_array_from_ints(edge), 256, 4))
assert got == M.enc96(M.ec_mul(sum(i * s for i, s in enumerate(edge)) % Q))
msm.append({"points": "first 5", "n": 5, "scalars_le32": [le(s) for s in edge], "scalars_python_random_seed": None, "result96": got.hex()})
out["msm"] = msm

# ---- DFT (utils.rs:63-81, 106-129): explicit small cases, hashes for 2^7..2^10 ----
ntt = []
for k in range(0, 11):
    n = 1 << k
    rnd = random.Random(100 + k)
    x = [rnd.randrange(Q) for _ in range(n)]
    fwd = O.fr_array_to_ints(O.ntt_fast(O.fr_array_from_ints(x)))
    inv = O.fr_array_to_ints(O.ntt_fast(O.fr_array_from_ints(x), inverse=True))
    if n <= 128:
        assert fwd == O.fr_array_to_ints(O.ntt_381(O.fr_array_from_ints(x))) == M.dft(x)          # faithful O(n^2) restatement and big-int model
        assert inv == M.dft(x, inverse=True)
    enc = lambda v: b"".join(t.to_bytes(32, "little") for t in v)
    e = {"log_n": k, "input_python_random_seed": 100 + k, "forward_sha256": hashlib.sha256(enc(fwd)).hexdigest(),
         "inverse_sha256": hashlib.sha256(enc(inv)).hexdigest()}
    if n <= 8:
        e.update(input_le32=[le(v) for v in x], forward_le32=[le(v) for v in fwd], inverse_le32=[le(v) for v in inv])
    ntt.append(e)
out["ntt"] = ntt

# ---- Polynomial operators (polynomial.rs:189-380) incl. the Div quirk ----
rnd = random.Random(7)
a = [rnd.randrange(Q) for _ in range(9)]
b = [rnd.randrange(Q) for _ in range(5)]
P = lambda v: O.fr_array_from_ints(v)
prod = O.fr_array_to_ints(O.poly_binop("poly_mul_fast", P(a), P(b)))
conv = [0] * 13
for i, u in enumerate(a):
    for j, v in enumerate(b):
        conv[i + j] = (conv[i + j] + u * v) % Q
assert prod == conv
quo = O.fr_array_to_ints(O.poly_binop("poly_div", P(prod), P(b)))
assert quo == a
zn = [Q - 1] + [0] * 7 + [1]                                       # x^8 - 1
f = [rnd.randrange(Q) for _ in range(6)]
fz = [0] * 14
for i, u in enumerate(f):
    for j, v in enumerate(zn):
        fz[i + j] = (fz[i + j] + u * v) % Q
assert O.fr_array_to_ints(O.poly_binop("poly_div", P(fz), P(zn))) == f
quirk_num, quirk_den = [(-1) % Q, 0, 0, 1], [(-1) % Q, 1]         # (x^3 - 1)/(x - 1) = x^2 + x + 1, no zero coefficient
quirk2_num = [0, 0, 0, 0, 1]                                        # x^4 / x^2 = x^2: the reference squeezes the zero coefficients out
q2 = O.fr_array_to_ints(O.poly_binop("poly_div", P(quirk2_num), P([0, 0, 1])))
x0 = rnd.randrange(Q)
out["poly"] = {"a_le32": [le(v) for v in a], "b_le32": [le(v) for v in b], "a_times_b_le32": [le(v) for v in prod],
               "f_le32": [le(v) for v in f], "f_times_x8_minus_1_le32": [le(v) for v in fz],
               "x4_div_x2_le32": [le(v) for v in q2], "eval_point_le32": le(x0),
               "a_at_point_le32": le(sum(v * pow(x0, i, Q) for i, v in enumerate(a)))}
assert O.fr_to_int(O.poly_eval(P(a), O.fr_from_int(x0))) == sum(v * pow(x0, i, Q) for i, v in enumerate(a)) % Q

# ---- the toy circuit of tests/verify_proof_test.rs with fixed blinders: the 624-byte proof and per-round commitments ----
from tests.test_gpu_prover_rounds import prove_with_blinding, toy_circuit
n, tau = 8, 101
cols, pk, public = toy_circuit(n)
blinders = [random.Random(99).randrange(1, Q) for _ in range(11)]
cur, srs = O.g1_generator(), b""
for _ in range(n + 6):
    srs += O.g1_bytes96(cur)
    cur = O.g1_mul(cur, O.fr_from_int(tau))
cpu = PR.OracleBackend(O.proj_from_bytes96(srs))
proof, ev, blob = prove_with_blinding(cpu, n, cols, pk, public, blinders)
assert hashlib.sha256(blob).hexdigest() == open(os.path.join(HERE, "toy_proof_blinders_seed99.sha256")).read().strip()
out["toy_proof"] = {"circuit": "tests/verify_proof_test.rs:16-44: group order 8, SRS = 14 powers of tau = 101, witness a=3 b=4 c=16 d=5 e=80",
                    "blinders_le32": [le(b) for b in blinders], "srs96": srs.hex(),
                    "columns": {k: [le(v) for v in col] for k, col in pk.items()},
                    "wires_a_b_c": [[le(v) for v in c] for c in cols], "public_input_column": [le(v) for v in public],
                    "commitments96": {k: v.hex() for k, v in proof.items()}, "evaluations_le32": {k: le(v) for k, v in ev.items()},
                    "proof624": blob.hex()}
json.dump(out, open(os.path.join(HERE, "path_vectors.json"), "w"), indent=1)
print("wrote path_vectors.json,", os.path.getsize(os.path.join(HERE, "path_vectors.json")), "bytes")

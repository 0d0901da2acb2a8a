"""tests/golden/path_vectors.json -- committed golden vectors of the hot path in the C-ABI wire formats (made on the CPU by
tests/golden/make_path_vectors.py: oracle + independent big-int model).  The CPU test re-derives them with the oracle
(fixture drift), the GPU test reproduces every one of them through the C ABI without touching the oracle."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

HERE = os.path.dirname(__file__)
V = json.load(open(os.path.join(HERE, "golden", "path_vectors.json")))
UNCOMP = open(os.path.join(HERE, "golden", "g1_uncompressed_valid_test_vectors.dat"), "rb").read()
Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ints = lambda hexes: [int.from_bytes(bytes.fromhex(h), "little") for h in hexes]
enc = lambda vals: b"".join(int(v).to_bytes(32, "little") for v in vals)


def msm_scalars(case):
    if case["scalars_le32"] is not None:
        return ints(case["scalars_le32"])
    rnd = random.Random(case["scalars_python_random_seed"])
    return [rnd.randrange(Q) for _ in range(case["n"])]


def test_oracle_reproduces_the_fixture():
    from oracle import oracle as O
    pts = O.proj_from_bytes96(UNCOMP)
    for case in V["msm"]:
        sc = msm_scalars(case)
        assert O.g1_bytes96(O.bucket_msm(pts[:case["n"]].copy(), O.fr_array_from_ints(sc), 256, 4, threads=8)).hex() == case["result96"]
    for case in V["ntt"]:
        rnd = random.Random(case["input_python_random_seed"])
        x = [rnd.randrange(Q) for _ in range(1 << case["log_n"])]
        assert hashlib.sha256(enc(O.fr_array_to_ints(O.ntt_fast(O.fr_array_from_ints(x))))).hexdigest() == case["forward_sha256"]
        assert hashlib.sha256(enc(O.fr_array_to_ints(O.ntt_fast(O.fr_array_from_ints(x), inverse=True)))).hexdigest() == case["inverse_sha256"]
        if "input_le32" in case:
            assert ints(case["input_le32"]) == x
            assert O.fr_array_to_ints(O.ntt_381(O.fr_array_from_ints(x))) == ints(case["forward_le32"])       # the faithful O(n^2) form
            assert O.fr_array_to_ints(O.i_ntt_381(O.fr_array_from_ints(x))) == ints(case["inverse_le32"])
    p = V["poly"]
    P = lambda key: O.fr_array_from_ints(ints(p[key]))
    assert O.fr_array_to_ints(O.poly_binop("poly_mul_fast", P("a_le32"), P("b_le32"))) == ints(p["a_times_b_le32"])
    assert O.fr_array_to_ints(O.poly_binop("poly_div", P("a_times_b_le32"), P("b_le32"))) == ints(p["a_le32"])


@pytest.mark.gpu
def test_library_reproduces_the_fixture():
    import baby_plonk_rust_amd as bp
    ctx = bp.default_context()
    S = bp.scalars_from_ints
    T = bp.scalars_to_ints
    # MSM, with and without the SRS's fixed-base tables, Montgomery and canonical-bytes scalars
    for case in V["msm"]:
        n, sc = case["n"], msm_scalars(case)
        h = ctx.srs_load(UNCOMP[:96 * n])
        assert ctx.msm(h, S(sc)).hex() == case["result96"]
        le = np.frombuffer(enc(sc), dtype=np.uint8).reshape(-1, 32)
        assert ctx.msm(h, le, fmt=bp.FR_BYTES_LE).hex() == case["result96"]
        ctx.srs_precompute(h, 4)
        assert ctx.msm(h, S(sc)).hex() == case["result96"]
        ctx.srs_free(h)
    for case in V["ntt"]:
        rnd = random.Random(case["input_python_random_seed"])
        x = [rnd.randrange(Q) for _ in range(1 << case["log_n"])]
        fwd, inv = T(bp.ntt_381(S(x))), T(bp.i_ntt_381(S(x)))
        assert hashlib.sha256(enc(fwd)).hexdigest() == case["forward_sha256"]
        assert hashlib.sha256(enc(inv)).hexdigest() == case["inverse_sha256"]
        if "forward_le32" in case:
            assert fwd == ints(case["forward_le32"]) and inv == ints(case["inverse_le32"])
    p = V["poly"]
    P = lambda key: bp.Polynomial(S(ints(p[key])), bp.BASIS_MONOMIAL, ctx)
    assert T((P("a_le32") * P("b_le32")).values) == ints(p["a_times_b_le32"])
    assert T((P("a_times_b_le32") / P("b_le32")).values) == ints(p["a_le32"])
    x8m1 = bp.Polynomial(S([Q - 1] + [0] * 7 + [1]), bp.BASIS_MONOMIAL, ctx)
    assert T((P("f_times_x8_minus_1_le32") / x8m1).values) == ints(p["f_le32"])
    x4, x2 = bp.Polynomial(S([0, 0, 0, 0, 1]), bp.BASIS_MONOMIAL, ctx), bp.Polynomial(S([0, 0, 1]), bp.BASIS_MONOMIAL, ctx)
    assert T((x4 / x2).values) == ints(p["x4_div_x2_le32"])                       # the Div quirk (polynomial.rs:371-376)
    assert bp.scalar_to_int(P("a_le32").coeffs_evaluate(bp.scalar_from_int(ints([p["eval_point_le32"]])[0]))) == ints([p["a_at_point_le32"]])[0]
    # the toy proof: SRS and circuit from the fixture's bytes, every commitment, evaluation and the 624 proof bytes
    t = V["toy_proof"]
    setup = bp.Setup.from_points(bytes.fromhex(t["srs96"]), ctx)
    circuit = bp.Circuit({k: S(ints(v)) for k, v in t["columns"].items()}, ctx)
    a, b, c = (S(ints(col)) for col in t["wires_a_b_c"])
    blob = bp.Prover(setup, circuit).prove_with_blinding(a, b, c, S(ints(t["public_input_column"])), ints(t["blinders_le32"]))
    assert blob.hex() == t["proof624"]
    from tests import bigint_model as M
    names = ("a_1", "b_1", "c_1", "z_1", "t_lo_1", "t_mid_1", "t_hi_1", "w_zeta_1", "w_zeta_omega_1")
    for i, k in enumerate(names):                     # the proof's compressed points are the fixture's uncompressed commitments
        pt = bytes.fromhex(t["commitments96"][k])
        dec = None if pt[0] & 0x40 else (int.from_bytes(pt[:48], "big"), int.from_bytes(pt[48:], "big"))
        assert M.enc48(dec) == blob[48 * i: 48 * i + 48]
    for i, k in enumerate(("a_bar", "b_bar", "c_bar", "s1_bar", "s2_bar", "z_omega_bar")):
        assert blob[432 + 32 * i: 464 + 32 * i].hex() == t["evaluations_le32"][k]
    circuit.free()

"""-m gpu: Polynomial operators (src/polynomial.rs:14-380) through the C ABI vs the oracle restatement,
including the reference's own unit-test literals (polynomial.rs:386-521) and its quirks."""
import random

import numpy as np
import pytest

import baby_plonk_rust_amd as bp
from oracle import oracle as O
from tests.gpu_common import Q

pytestmark = pytest.mark.gpu
MONO, LAG = bp.BASIS_MONOMIAL, bp.BASIS_LAGRANGE


@pytest.fixture(scope="module")
def ctx():
    return bp.default_context()


def P(vals, basis=MONO):
    return bp.Polynomial(bp.scalars_from_ints([v % Q for v in vals]), basis)


def ints(p):
    return bp.scalars_to_ints(p.values)


def S(v):
    return bp.scalar_from_int(v % Q)


def test_add_sub_scalar_literals(ctx):
    assert ints(P([1, 2, 3]) + P([4, 5, 6])) == [5, 7, 9]
    assert ints(P([1, 2, 3]) + P([4, 5])) == [5, 7, 3]
    assert ints(P([4, 5, 6]) - P([1, 2])) == [3, 3, 6]
    assert ints(P([1]) - P([1, 2])) == [0, Q - 2]
    assert ints(P([1, 2, 3], LAG) + P([4, 5, 6], LAG)) == [5, 7, 9]
    with pytest.raises(bp.BpError) as e:
        P([1, 2, 3], LAG) + P([4, 5], LAG)
    assert e.value.code == -6
    with pytest.raises(bp.BpError) as e:
        P([1, 2, 3], LAG) + P([4, 5, 6], MONO)
    assert e.value.code == -5
    assert ints(P([1, 2, 3]) * S(2)) == [2, 4, 6]
    assert ints(P([1, 2, 3]) + S(2)) == [3, 2, 3]
    assert ints(P([1, 2, 3], LAG) + S(2)) == [3, 4, 5]
    assert ints(P([1, 2, 3]) - S(2)) == [Q - 1, 2, 3]
    assert ints(P([1, 2, 3], LAG) - S(2)) == [3, 4, 5]           # reference quirk (polynomial.rs:126-128)
    assert bp.scalars_to_ints(P([1, 2, 3], LAG).shift_left(1)) == [2, 3, 1]
    assert ints(P([1, 2]).rlc(P([3, 4]), S(3), S(4))) == [14, 14]                # impl Rlc for Polynomial (utils.rs:170-175)
    assert ints(P([1, 2], LAG).rlc(P([3, 4], LAG), S(3), S(4))) == [14, 18]


def test_mul_literals_and_random(ctx):
    assert ints(P([1, 1]) * P([1, 1])) == [1, 2, 1]             # polynomial.rs:437-451
    rnd = random.Random(31)
    for na, nb in ((1, 1), (3, 2), (5, 8), (9, 9), (100, 29), (1000, 1025), (5000, 3000)):
        a = O.splitmix_scalars(na, rnd.randrange(2**32))
        b = O.splitmix_scalars(nb, rnd.randrange(2**32))
        got = bp.Polynomial(a, MONO) * bp.Polynomial(b, MONO)
        assert len(got) == na + nb - 1
        assert (got.values == O.poly_binop("poly_mul_fast", a, b)).all()
        if na + nb <= 20:
            assert (got.values == O.poly_binop("poly_mul", a, b)).all()      # the reference's literal algorithm
    with pytest.raises(bp.BpError) as e:
        P([1, 2], LAG) * P([1, 2], LAG)                                      # todo!() in the reference
    assert e.value.code == -5


def test_eval(ctx):
    assert bp.scalar_to_int(P([1, 3, 2]).coeffs_evaluate(S(2))) == 15
    rnd = random.Random(32)
    for n in (1, 2, 17, 256, 4097, 100000):
        c = O.splitmix_scalars(n, rnd.randrange(2**32))
        x = O.fr_from_int(rnd.randrange(Q))
        assert (bp.Polynomial(c, MONO).coeffs_evaluate(x) == O.poly_eval(c, x, fast=True)).all()
        if n <= 17:
            assert (bp.Polynomial(c, MONO).coeffs_evaluate(x) == O.poly_eval(c, x)).all()
    with pytest.raises(bp.BpError):
        P([1, 2], LAG).coeffs_evaluate(S(3))


def test_div_literals_quirk_and_random(ctx):
    assert ints(P([-1, -1, -1, 3]) / P([-1, 1])) == [1, 2, 3]
    assert ints(P([-1, -1, -1, 3, 0, 0]) / P([-1, 1, 0])) == [1, 2, 3]
    assert ints(P([1, 0, 1]) / P([1, 1])) == [Q - 1, 1]
    assert ints(P([-1, 0, 0, 0, 1]) / P([-1, 0, 1])) == [1, 1]    # zero quotient coefficient squeezed out
    assert ints(P([5]) / P([1, 1])) == []
    assert ints(P([0, 0]) / P([1, 1])) == []
    with pytest.raises(bp.BpError) as e:
        P([1, 2]) / P([0, 0])
    assert e.value.code == -7
    with pytest.raises(bp.BpError) as e:
        P([1, 2], LAG) / P([1, 1], LAG)                           # polynomial.rs:523-547 test_lagrange_div panics
    assert e.value.code == -5
    rnd = random.Random(33)
    # the prover's divisors: Z_H = x^n - 1 (prover.rs:450), x - zeta (prover.rs:623-638), plus general ones
    for nq, divisor in ((9, [5, 0, 0, 7]), (40, [-1] + [0] * 7 + [1]), (3000, [-1] + [0] * 1023 + [1]),
                        (5000, [-rnd.randrange(Q), 1]), (70, [3, 1, 4, 1, 5]), (300, [rnd.randrange(Q) for _ in range(33)])):
        q = O.splitmix_scalars(nq, rnd.randrange(2**32))
        dv = bp.scalars_from_ints([v % Q for v in divisor])
        prod = O.poly_binop("poly_mul_fast", q, dv)
        got = bp.Polynomial(prod, MONO) / bp.Polynomial(dv, MONO)
        assert (got.values == q).all()
        assert (got.values == O.poly_binop("poly_div", prod, dv)).all()
        # with a remainder: quotient unchanged
        prod2 = prod.copy()
        prod2[0] = bp.scalar_from_int(bp.scalar_to_int(prod2[0]) + 1)
        if len(divisor) > 1:
            assert ((bp.Polynomial(prod2, MONO) / bp.Polynomial(dv, MONO)).values == O.poly_binop("poly_div", prod2, dv)).all()


def test_ntt_roundtrip_methods(ctx):
    c = O.splitmix_scalars(64, 99)
    p = bp.Polynomial(c, MONO)
    assert p.ntt().basis == LAG and (p.ntt().i_ntt().values == c).all()
    with pytest.raises(bp.BpError):
        p.i_ntt()


def test_canonical_byte_format_paths(ctx):
    """BP_FR_BYTES_LE (Scalar::to_bytes) in and out for every host-pointer operator"""
    import ctypes as C
    rnd = random.Random(34)
    lib, h = ctx._lib, ctx._h
    le = lambda vals: np.frombuffer(b"".join((v % Q).to_bytes(32, "little") for v in vals), dtype=np.uint8).copy()
    back = lambda buf, n: [int.from_bytes(bytes(buf[32 * i: 32 * i + 32]), "little") for i in range(n)]
    a, b = [rnd.randrange(Q) for _ in range(7)], [rnd.randrange(Q) for _ in range(4)]
    A, B = le(a), le(b)
    out, n = np.zeros(32 * 16, dtype=np.uint8), C.c_size_t()
    ctx.check(lib.bp_poly_mul(h, A.ctypes.data, 7, B.ctypes.data, 4, MONO, bp.FR_BYTES_LE, out.ctypes.data, C.byref(n)), "mul")
    want = [0] * 10
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            want[i + j] = (want[i + j] + x * y) % Q
    assert n.value == 10 and back(out, 10) == want
    prod = le(want)
    ctx.check(lib.bp_poly_div(h, prod.ctypes.data, 10, B.ctypes.data, 4, MONO, bp.FR_BYTES_LE, out.ctypes.data, C.byref(n)), "div")
    assert n.value == 7 and back(out, 7) == a
    ctx.check(lib.bp_poly_add(h, A.ctypes.data, 7, B.ctypes.data, 4, MONO, bp.FR_BYTES_LE, out.ctypes.data, C.byref(n)), "add")
    assert back(out, 7) == [(x + (b[i] if i < 4 else 0)) % Q for i, x in enumerate(a)]
    ctx.check(lib.bp_poly_sub(h, A.ctypes.data, 7, B.ctypes.data, 4, MONO, bp.FR_BYTES_LE, out.ctypes.data, C.byref(n)), "sub")
    assert back(out, 7) == [(x - (b[i] if i < 4 else 0)) % Q for i, x in enumerate(a)]
    x = le([12345])
    r = np.zeros(32, dtype=np.uint8)
    ctx.check(lib.bp_poly_evaluate(h, A.ctypes.data, 7, MONO, x.ctypes.data, bp.FR_BYTES_LE, r.ctypes.data), "eval")
    assert back(r, 1) == [sum(v * pow(12345, i, Q) for i, v in enumerate(a)) % Q]
    ctx.check(lib.bp_poly_scalar_op(h, A.ctypes.data, 7, LAG, x.ctypes.data, 2, bp.FR_BYTES_LE, out.ctypes.data), "scalar mul")
    assert back(out, 7) == [v * 12345 % Q for v in a]
    # NTT in canonical bytes with a batch
    data = le([rnd.randrange(Q) for _ in range(32)])
    orig = back(data, 32)
    ctx.check(lib.bp_ntt_fr(h, data.ctypes.data, 3, 0, bp.FR_BYTES_LE, 4, 8), "ntt batch")
    from tests import bigint_model as M
    for bidx in range(4):
        assert back(data, 32)[8 * bidx: 8 * bidx + 8] == M.dft(orig[8 * bidx: 8 * bidx + 8])
    # a non-canonical scalar argument is rejected
    bad = np.frombuffer((Q + 1).to_bytes(32, "little"), dtype=np.uint8).copy()
    assert lib.bp_poly_evaluate(h, A.ctypes.data, 7, MONO, bad.ctypes.data, bp.FR_BYTES_LE, r.ctypes.data) == -4


def test_division_2p16_against_the_committed_oracle_hashes():
    """Polynomial / Polynomial (polynomial.rs:314-380) on 2^16 + 11 coefficients by x^n - 1 (n = 2^14, the round-3 shape) and by
    x - zeta (the round-5 shape): sha256 of the quotient limbs recorded by the CPU oracle (tests/golden/make_large_vectors.py);
    host-pointer and HBM-resident entry points"""
    import hashlib
    import json
    import os
    from tests import prover_rounds as PR
    rec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "large_vectors.json")))["division"]
    n, zeta = rec["n"], int(rec["zeta"], 16)
    a = O.splitmix_scalars((1 << 16) + 11, 0xD1F1)
    zh = PR.sparse(n + 1, {0: Q - 1, n: 1})
    lin = PR.sparse(2, {0: Q - zeta, 1: 1})
    for divisor, key in ((zh, "by_xn_minus_1"), (lin, "by_x_minus_zeta")):
        q = bp.Polynomial(a, bp.BASIS_MONOMIAL) / bp.Polynomial(divisor, bp.BASIS_MONOMIAL)
        assert len(q) == rec[key]["len"] and hashlib.sha256(q.values.tobytes()).hexdigest() == rec[key]["sha256"], key
        qd = bp.DevicePolynomial(a, bp.BASIS_MONOMIAL) / bp.DevicePolynomial(divisor, bp.BASIS_MONOMIAL)
        assert hashlib.sha256(np.ascontiguousarray(qd.values).tobytes()).hexdigest() == rec[key]["sha256"], key


def test_device_division_squeezes_zero_coefficients_at_size():
    """the reference's Div drops zero quotient coefficients (polynomial.rs:371-376); on HBM-resident operands the compaction is a
    device scan + scatter (no host round trip): a quotient of 70 001 coefficients with zeros at the ends, across tile edges and in
    a long run comes back with exactly those removed, in order"""
    nq, zeta = 70001, 0x5A5A5A5A1234567 % Q
    q = O.splitmix_scalars(nq, 0xC0FFEE)
    zero_at = [0, 1, 2047, 2048, 2049, 4096, 40000, nq - 2] + list(range(10000, 10300))
    q[zero_at] = 0
    assert not (q[nq - 1] == 0).all()
    lin = np.stack([bp.scalar_from_int(Q - zeta), bp.scalar_from_int(1)])
    qd, bd = bp.DevicePolynomial(q, bp.BASIS_MONOMIAL), bp.DevicePolynomial(lin, bp.BASIS_MONOMIAL)
    a = qd * bd                                             # exact product: a / (x - zeta) == q
    got = (a / bd).values
    keep = np.ones(nq, dtype=bool)
    keep[zero_at] = False
    assert got.shape == (int(keep.sum()), 4) and (got == q[keep]).all()
    host = bp.Polynomial(a.values, bp.BASIS_MONOMIAL) / bp.Polynomial(lin, bp.BASIS_MONOMIAL)
    assert (host.values == got).all()

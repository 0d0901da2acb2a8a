"""Pins the C oracle's Fr / Fp arithmetic against the reference's own known-answer tests
(tests/golden/ref_kats.json, extracted from lib/bls12_381/src/{fp,scalar}.rs) and against an
independent Python big-int model."""
import json
import os
import random

import numpy as np

from oracle import oracle as O
from tests.bigint_model import P, Q

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_kats.json")))


def H(test, fname):
    return [np.array([int(x, 16) for x in a], dtype=np.uint64) for a in KATS[fname][test]["hex_arrays"]]


def B(test, fname):
    return [bytes(a) for a in KATS[fname][test]["byte_arrays"]]


# ------------------------------------------------------------------ Fp (fp.rs:700-979)
def test_fp_squaring_kat():
    a, b = H("test_squaring", "fp.rs")
    assert (O.fp_un("fp_square", a) == b).all()


def test_fp_multiplication_kat():
    a, b, c = H("test_multiplication", "fp.rs")
    assert (O.fp_bin("fp_mul", a, b) == c).all()


def test_fp_addition_subtraction_negation_kat():
    a, b, c = H("test_addition", "fp.rs")
    assert (O.fp_bin("fp_add", a, b) == c).all()
    a, b, c = H("test_subtraction", "fp.rs")
    assert (O.fp_bin("fp_sub", a, b) == c).all()
    a, b = H("test_negation", "fp.rs")
    assert (O.fp_un("fp_neg", a) == b).all()


def test_fp_debug_string_kat():
    (a,) = H("test_debug", "fp.rs")
    assert "0x%096x" % O.fp_to_int(a) == KATS["fp.rs"]["test_debug"]["strings"][0]


def test_fp_from_bytes_kat():
    (a,) = H("test_from_bytes", "fp.rs")
    for _ in range(100):                       # fp.rs:862-868
        a = O.fp_un("fp_square", a)
        back, ok = O.fp_from_bytes(O.fp_to_bytes(a))
        assert ok and (back == a).all()
    minus_one, too_big = B("test_from_bytes", "fp.rs")
    out, ok = O.fp_from_bytes(minus_one)
    assert ok and (out == O.fp_un("fp_neg", O.fp_one())).all()
    assert not O.fp_from_bytes(too_big)[1]
    assert not O.fp_from_bytes(bytes([0xFF] * 48))[1]


def test_fp_sqrt_kat():
    four, two = H("test_sqrt", "fp.rs")
    r, ok = O.fp_sqrt(four)
    assert ok and (O.fp_un("fp_neg", r) == two).all()


def test_fp_inversion_kat():
    a, b = H("test_inversion", "fp.rs")
    r, ok = O.fp_invert(a)
    assert ok and (r == b).all()
    assert not O.fp_invert(O.u64(6))[1]


def test_fp_lexicographic_largest_kat():
    a, b, c = H("test_lexicographic_largest", "fp.rs")
    L = O.fp_lex_largest
    assert not L(O.u64(6)) and not L(O.fp_one()) and not L(a) and L(b) and L(c)


def test_fp_vs_bigint_model():
    rnd = random.Random(1)
    R = pow(2, 384, P)
    for _ in range(200):
        x, y = rnd.randrange(P), rnd.randrange(P)
        a, b = O.fp_from_int(x), O.fp_from_int(y)
        assert O.unlimbs(a) == x * R % P                      # Montgomery representative
        assert O.fp_to_int(O.fp_bin("fp_mul", a, b)) == x * y % P
        assert O.fp_to_int(O.fp_bin("fp_add", a, b)) == (x + y) % P
        assert O.fp_to_int(O.fp_bin("fp_sub", a, b)) == (x - y) % P
        assert O.fp_to_int(O.fp_un("fp_neg", a)) == (-x) % P
    for x in (0, 1, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2):
        a = O.fp_from_int(x)
        assert O.fp_to_int(O.fp_bin("fp_mul", a, a)) == x * x % P
        assert O.fp_to_int(O.fp_bin("fp_add", a, a)) == 2 * x % P


# ------------------------------------------------------------------ Fr (scalar.rs:794-1258)
def test_fr_constants():
    assert O.unlimbs(O.fr_const("FR_MODULUS")) == Q
    root, rinv = O.fr_const("FR_ROOT_OF_UNITY"), O.fr_const("FR_ROOT_OF_UNITY_INV")
    assert O.fr_to_int(O.fr_bin("fr_mul", root, rinv)) == 1                         # scalar.rs:803-806
    e = np.array([1 << 32, 0, 0, 0], dtype=np.uint64)
    assert O.fr_to_int(O.fr_bin("fr_pow", root, e)) == 1                            # scalar.rs:809-812
    assert O.fr_to_int(root) == pow(7, (Q - 1) >> 32, Q)


def test_fr_to_from_bytes_kat():
    zero, one, r2, minus_one = B("test_to_bytes", "scalar.rs")
    R, R2, tb = O.fr_const("FR_R"), O.fr_const("FR_R2"), O.fr_to_bytes

    assert tb(O.u64(4)) == zero and tb(R) == one and tb(R2) == r2
    assert tb(O.fr_un("fr_neg", R)) == minus_one
    vecs = B("test_from_bytes", "scalar.rs")                 # scalar.rs:904-972
    expect_ok = [1, 1, 1, 1, 0, 0, 0, 0]
    expect_val = [O.u64(4), R, R2, None, None, None, None, None]
    for v, ok, val in zip(vecs, expect_ok, expect_val):
        out, got = O.fr_from_bytes(v)
        assert got == bool(ok)
        if val is not None:
            assert (out == val).all()


def test_fr_from_bytes_wide_kat():
    R, R2, R3, wide = O.fr_const("FR_R"), O.fr_const("FR_R2"), O.fr_const("FR_R3"), O.fr_from_bytes_wide

    assert (wide(B("test_from_bytes_wide_r2", "scalar.rs")[0]) == R2).all()
    assert (wide(B("test_from_bytes_wide_negative_one", "scalar.rs")[0]) == O.fr_un("fr_neg", R)).all()
    assert (wide(bytes([0xFF] * 64)) == H("test_from_bytes_wide_maximum", "scalar.rs")[0]).all()

    u512 = O.fr_from_u512

    assert (u512(O.limbs(Q, 4) + [0] * 4) == 0).all()                                # scalar.rs:975-989
    assert (u512([1] + [0] * 7) == R).all()                                          # :992-994
    assert (u512([0] * 4 + [1] + [0] * 3) == R2).all()                               # :997-999
    assert (u512([2**64 - 1] * 8) == O.fr_bin("fr_sub", R3, R)).all()                # :1002-1008


def test_fr_largest_add_neg_sub_kat():
    (largest,) = H("LARGEST", "scalar.rs")
    (twice,) = H("test_addition", "scalar.rs")
    one_raw = np.array([1, 0, 0, 0], dtype=np.uint64)
    assert (O.fr_bin("fr_add", largest, largest) == twice).all()
    assert (O.fr_bin("fr_add", largest, one_raw) == 0).all()
    assert (O.fr_un("fr_neg", largest) == one_raw).all()
    assert (O.fr_un("fr_neg", one_raw) == largest).all()
    assert (O.fr_un("fr_neg", O.u64(4)) == 0).all()
    assert (O.fr_bin("fr_sub", largest, largest) == 0).all()
    assert (O.fr_bin("fr_sub", O.u64(4), largest) == one_raw).all()                  # MODULUS - LARGEST = 1


def test_fr_mul_square_vs_double_and_add():
    """scalar.rs:1112-1168 restated: cur*cur equals double-and-add over cur's bits, 100 values from LARGEST"""
    (largest,) = H("LARGEST", "scalar.rs")
    cur = largest.copy()
    for _ in range(100):
        prod = O.fr_bin("fr_mul", cur, cur)
        assert (O.fr_un("fr_square", cur) == prod).all()
        bits = O.fr_to_int(cur)
        acc = O.u64(4)
        for i in range(255, -1, -1):
            acc = O.fr_bin("fr_add", acc, acc)
            if (bits >> i) & 1:
                acc = O.fr_bin("fr_add", acc, cur)
        assert (acc == prod).all()
        cur = O.fr_bin("fr_add", cur, largest)


def test_fr_inversion_and_pow():
    """scalar.rs:1170-1213 restated"""
    R, R2 = O.fr_const("FR_R"), O.fr_const("FR_R2")
    assert not O.fr_invert(O.u64(4))[1]
    out, ok = O.fr_invert(R)
    assert ok and (out == R).all()
    m1 = O.fr_un("fr_neg", R)
    out, ok = O.fr_invert(m1)
    assert ok and (out == m1).all()
    tmp = R2.copy()
    qm2 = np.array(O.limbs(Q - 2, 4), dtype=np.uint64)
    for _ in range(30):
        out, _ = O.fr_invert(tmp)
        assert (O.fr_bin("fr_mul", out, tmp) == R).all()
        assert (O.fr_bin("fr_pow_vartime", tmp, qm2) == out).all()
        tmp = O.fr_bin("fr_add", tmp, R2)


def test_fr_from_raw_kat():
    (a,) = H("test_from_raw", "scalar.rs")
    R = O.fr_const("FR_R")
    fr = lambda l: O.fr_un("fr_from_raw", np.array(l, dtype=np.uint64))
    assert (fr(a) == fr([2**64 - 1] * 4)).all()
    assert (fr(O.limbs(Q, 4)) == 0).all()
    assert (fr([1, 0, 0, 0]) == R).all()


def test_fr_vs_bigint_model():
    rnd = random.Random(2)
    R = pow(2, 256, Q)
    for _ in range(200):
        x, y = rnd.randrange(Q), rnd.randrange(Q)
        a, b = O.fr_from_int(x), O.fr_from_int(y)
        assert O.unlimbs(a) == x * R % Q
        assert O.fr_to_int(O.fr_bin("fr_mul", a, b)) == x * y % Q
        assert O.fr_to_int(O.fr_bin("fr_add", a, b)) == (x + y) % Q
        assert O.fr_to_int(O.fr_bin("fr_sub", a, b)) == (x - y) % Q
        assert O.fr_to_int(O.fr_un("fr_neg", a)) == (-x) % Q

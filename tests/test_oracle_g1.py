"""Pins the C oracle's G1 group law and wire encodings against the reference's fixtures:
the two 1000-point .dat files (lib/bls12_381/src/tests/mod.rs:3-55), the literal KATs of
g1.rs:1263-1297 / :1372-1417, and an independent Python big-int model."""
import json
import os
import random

import numpy as np

from oracle import oracle as O
from tests import bigint_model as M

HERE = os.path.dirname(__file__)
KATS = json.load(open(os.path.join(HERE, "golden", "ref_kats.json")))["g1.rs"]
UNCOMP = open(os.path.join(HERE, "golden", "g1_uncompressed_valid_test_vectors.dat"), "rb").read()
COMP = open(os.path.join(HERE, "golden", "g1_compressed_valid_test_vectors.dat"), "rb").read()


def H(test):
    return [np.array([int(x, 16) for x in a], dtype=np.uint64) for a in KATS[test]["hex_arrays"]]


def test_wire_vectors_1000_points():
    """tests/mod.rs:3-29: e = identity; 1000 times {encode, compare, decode, e += G}"""
    assert len(UNCOMP) == 96000 and len(COMP) == 48000
    e, g = O.g1_identity(), O.g1_generator()
    for i in range(1000):
        aff = O.g1_to_affine(e)
        enc_u, enc_c = O.g1_to_uncompressed(aff), O.g1_to_compressed(aff)
        assert enc_u == UNCOMP[96 * i: 96 * i + 96], i
        assert enc_c == COMP[48 * i: 48 * i + 48], i
        dec_u, ok_u = O.g1_from_uncompressed(enc_u)
        dec_c, ok_c = O.g1_from_compressed(enc_c)
        assert ok_u and ok_c and (dec_u == aff).all() and (dec_c == aff).all(), i
        e = O.g1_add(e, g)


def test_wire_vectors_match_bigint_model():
    """the fixture itself equals i*G computed by independent affine big-int arithmetic"""
    pt = None
    for i in range(0, 1000):
        if i % 37 == 0 or i < 4:
            assert M.enc96(pt) == UNCOMP[96 * i: 96 * i + 96]
            assert M.enc48(pt) == COMP[48 * i: 48 * i + 48]
        pt = M.ec_add(pt, (M.GX, M.GY))


def test_doubling_kat():
    x, y = H("test_doubling")
    d = O.g1_double(O.g1_generator())
    aff = O.g1_to_affine(d)
    assert (aff[:6] == x).all() and (aff[6:12] == y).all() and aff[12] == 0
    ident2 = O.g1_double(O.g1_identity())
    assert O.lib.g1_is_identity(ident2.ctypes.data) and O.lib.g1_is_on_curve(ident2.ctypes.data)
    assert O.lib.g1_is_on_curve(d.ctypes.data) and not O.lib.g1_is_identity(d.ctypes.data)


def test_projective_addition_kats():
    z1, z2, beta, x, y = H("test_projective_addition")
    assert (z1 == z2).all()
    g, ident = O.g1_generator(), O.g1_identity()
    # identity + rescaled generator (g1.rs:1308-1354)
    b = np.concatenate([O.fp_bin("fp_mul", g[:6], z1), O.fp_bin("fp_mul", g[6:12], z1), z1])
    for c in (O.g1_add(ident, b), O.g1_add(b, ident)):
        assert O.g1_eq(c, g) and not O.lib.g1_is_identity(c.ctypes.data) and O.lib.g1_is_on_curve(c.ctypes.data)
    # 4P + 2P == 6P (g1.rs:1355-1369)
    a4, b2 = O.g1_double(O.g1_double(g)), O.g1_double(g)
    d = g
    for _ in range(5):
        d = O.g1_add(d, g)
    assert O.g1_eq(O.g1_add(a4, b2), d)
    # degenerate case: same y, x scaled by a cube root of unity (g1.rs:1372-1417)
    beta2 = O.fp_un("fp_square", beta)
    bb = np.concatenate([O.fp_bin("fp_mul", a4[:6], beta2), O.fp_un("fp_neg", a4[6:12]), a4[12:]])
    assert O.lib.g1_is_on_curve(bb.ctypes.data)
    c = O.g1_add(a4, bb)
    expect = O.g1_to_affine(np.concatenate([x, y, O.fp_one()]))
    assert (O.g1_to_affine(c) == expect).all()


def test_mixed_addition_kats():
    z1, z2, beta, x, y = H("test_mixed_addition")
    g = O.g1_generator()
    gaff = O.g1_to_affine(g)
    ident_aff = O.u64(13)
    O.lib.g1_affine_identity(ident_aff.ctypes.data)
    assert O.g1_eq(O.g1_add_mixed(O.g1_identity(), gaff), g)
    b = np.concatenate([O.fp_bin("fp_mul", g[:6], z1), O.fp_bin("fp_mul", g[6:12], z1), z1])
    assert O.g1_eq(O.g1_add_mixed(b, ident_aff), g)
    a4, b2 = O.g1_double(O.g1_double(g)), O.g1_double(g)
    d = g
    for _ in range(5):
        d = O.g1_add_mixed(d, gaff)
    assert O.g1_eq(O.g1_add_mixed(a4, O.g1_to_affine(b2)), d)
    beta2 = O.fp_un("fp_square", beta)
    a4aff = O.g1_to_affine(a4)
    bb = np.concatenate([O.fp_bin("fp_mul", a4aff[:6], beta2), O.fp_un("fp_neg", a4aff[6:12]), np.zeros(1, dtype=np.uint64)])
    c = O.g1_add_mixed(a4, bb)
    expect = O.g1_to_affine(np.concatenate([x, y, O.fp_one()]))
    assert (O.g1_to_affine(c) == expect).all()
    # P + P and P + (-P) through the complete mixed formula
    assert O.g1_eq(O.g1_add_mixed(g, gaff), O.g1_double(g))
    assert O.lib.g1_is_identity(O.g1_add_mixed(O.g1_neg(g), gaff).ctypes.data)


def test_scalar_mul_and_batch_normalize():
    """g1.rs:1558-1595 (a*b)G == a(bG) on fresh values; g1.rs:1691-1727 batch_normalize == per-point"""
    rnd = random.Random(5)
    g = O.g1_generator()
    for _ in range(3):
        a, b = rnd.randrange(M.Q), rnd.randrange(M.Q)
        lhs = O.g1_mul(O.g1_mul(g, O.fr_from_int(a)), O.fr_from_int(b))
        rhs = O.g1_mul(g, O.fr_from_int(a * b % M.Q))
        assert O.g1_eq(lhs, rhs)
        assert O.g1_bytes96(rhs) == M.enc96(M.ec_mul(a * b))
    pts = np.stack([O.g1_double(g), O.g1_identity(), O.g1_mul(g, O.fr_from_int(77)), O.g1_identity(), g])
    out = O.u64((5, 13))
    O.lib.g1_batch_normalize(pts.ctypes.data, out.ctypes.data, 5)
    for i in range(5):
        assert (out[i] == O.g1_to_affine(pts[i])).all()


def test_srs_identities():
    """src/setup.rs:46-57: powers_of_x[i] == G * tau^i via sequential *= tau"""
    tau = O.fr_from_int(2)
    cur = O.g1_generator()
    for i in range(8):
        assert O.g1_eq(cur, O.g1_mul(O.g1_generator(), O.fr_from_int(pow(2, i, M.Q))))
        cur = O.g1_mul(cur, tau)

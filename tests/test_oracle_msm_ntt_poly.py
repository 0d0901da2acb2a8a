"""Pins the C oracle's bucket MSM, DFT and Polynomial restatements:
 - the reference's own MSM identities (src/setup.rs:60-116)
 - closed-form answers from the independent big-int model
 - the reference's polynomial unit tests (src/polynomial.rs:386-521), restated with their literals
"""
import random

import numpy as np
import pytest

from oracle import oracle as O
from tests import bigint_model as M

Q = M.Q


def fr(v):
    return O.fr_from_int(v % Q)


def frs(vals):
    return O.fr_array_from_ints([v % Q for v in vals])


def ints(a):
    return O.fr_array_to_ints(np.ascontiguousarray(a).reshape(-1, 4))


def srs(n, tau):
    """setup.rs:12-31 generate_srs: sequential cur *= tau"""
    pts, cur, t = [], O.g1_generator(), fr(tau)
    for _ in range(n):
        pts.append(cur)
        cur = O.g1_mul(cur, t)
    return np.stack(pts)


# ------------------------------------------------------------------ MSM
def test_get_c_bit_chunk_msb_first():
    v = 0x123456789ABCDEF0FEDCBA9876543210_0F1E2D3C4B5A69788796A5B4C3D2E1F0 % Q
    s = fr(v)
    for c in (1, 3, 4, 8, 16):
        for i in range(256 // c):
            want = (v >> (256 - c * (i + 1))) & ((1 << c) - 1)
            assert O.lib.msm_get_c_bit_chunk(s.ctypes.data, i, c) == want


def test_monomial_commit_identities():
    """setup.rs:60-116"""
    g = O.g1_generator()
    s = srs(2, 10)
    got = O.bucket_msm(s, frs([2, 3]))
    assert O.g1_eq(got, O.g1_add(O.g1_mul(g, fr(2)), O.g1_mul(g, fr(30))))
    s = srs(8, 2)
    assert O.g1_eq(O.bucket_msm(s, frs([0, 1])), O.g1_mul(g, fr(2)))
    assert O.g1_eq(O.bucket_msm(s, frs([0, 0, 1])), O.g1_mul(g, fr(4)))
    # (3x^2+2x+1)(x-1) = 3x^3-x^2-x-1 evaluated in the exponent at tau = 2 (pairing-free form of :103-115)
    lhs = O.g1_mul(O.bucket_msm(s, frs([1, 2, 3])), fr(2 - 1))
    rhs = O.bucket_msm(s, frs([-1, -1, -1, 3]))
    assert O.g1_eq(lhs, rhs)


def test_bucket_msm_closed_form_and_zip_truncation():
    rnd = random.Random(7)
    a, d, n = rnd.randrange(Q), rnd.randrange(Q), 40
    aff = O.points_progression(n, a, d)
    pts = O.affine_to_proj(aff)
    sc = [rnd.randrange(Q) for _ in range(n)]
    # edge scalars
    sc[0], sc[1], sc[2], sc[3] = 0, 1, Q - 1, 2**254
    want = sum(s * (a + i * d) for i, s in enumerate(sc)) % Q
    got = O.bucket_msm(pts, frs(sc))
    assert O.g1_bytes96(got) == M.enc96(M.ec_mul(want))
    assert O.g1_bytes96(O.bucket_msm(pts, frs(sc), threads=4)) == M.enc96(M.ec_mul(want))
    # msm.rs:29 zip(): shorter side wins, either way round
    want_short = sum(s * (a + i * d) for i, s in enumerate(sc[:17])) % Q
    assert O.g1_bytes96(O.bucket_msm(pts, frs(sc[:17]))) == M.enc96(M.ec_mul(want_short))
    assert O.g1_bytes96(O.bucket_msm(pts[:17], frs(sc))) == M.enc96(M.ec_mul(want_short))
    # window size does not change the point
    assert O.g1_eq(O.bucket_msm(pts, frs(sc), 256, 8), got)
    # all-equal points (tau = 1 SRS, prover.rs:684): every bucket add is a doubling
    same = np.stack([O.g1_generator()] * 9)
    assert O.g1_bytes96(O.bucket_msm(same, frs(sc[:9]))) == M.enc96(M.ec_mul(sum(sc[:9])))
    # empty input -> identity
    assert O.g1_bytes96(O.bucket_msm(pts[:0], frs([]))) == M.enc96(None)


def test_progression_points_match_model():
    rnd = random.Random(8)
    a, d = rnd.randrange(Q), rnd.randrange(Q)
    aff = O.points_progression(5, a, d)
    buf = bytes(O.points_to_bytes96(aff))
    for i in range(5):
        assert buf[96 * i: 96 * i + 96] == M.enc96(M.ec_mul(a + i * d))


def test_splitmix_scalars_match_model():
    got = ints(O.splitmix_scalars(6, 0x5EED0010))
    assert got == [M.splitmix_scalar(i, 0x5EED0010) for i in range(6)]


# ------------------------------------------------------------------ DFT
def test_root_of_unity():
    """utils.rs:239-242 and the omega constants listed in BASELINE.md section 6"""
    r = O.u64(4)
    O.lib.ntt_root_of_unity(r.ctypes.data, 4)
    w4 = O.fr_to_int(r)
    assert pow(w4, 4, Q) == 1 and pow(w4, 2, Q) != 1
    for n, want in ((8, 0x345766F603FA66E78C0625CD70D77CE2B38B21C28713B7007228FD3397743F7A),
                    (1 << 16, 0x2155379D12180CAA88F39A78F1AEB57867A665AE1FCADC91D7118F85CD96B8AD),
                    (1 << 20, 0x03E1C54BCB947035A57A6E07CB98DE4A2F69E02D265E09D9FECE7E0E39898D4B),
                    (1 << 24, 0x291CF6D68823E6876E0BCD91EE76273072CF6A8029B7D7BC92CF4DEB77BD779C)):
        O.lib.ntt_root_of_unity(r.ctypes.data, n)
        assert O.fr_to_int(r) == want == M.omega(n)
    roots = O.u64((8, 4))
    O.lib.ntt_roots_of_unity(roots.ctypes.data, 8)
    assert ints(roots) == [pow(M.omega(8), i, Q) for i in range(8)]


@pytest.mark.parametrize("n", [1, 2, 4, 8, 16, 64])
def test_dft_faithful_vs_model_and_fast(n):
    rnd = random.Random(n)
    vals = [rnd.randrange(Q) for _ in range(n)]
    a = frs(vals)
    fwd = O.ntt_381(a)
    assert ints(fwd) == M.dft(vals)
    assert (O.ntt_fast(a) == fwd).all()
    inv = O.i_ntt_381(a)
    assert ints(inv) == M.dft(vals, inverse=True)
    assert (O.ntt_fast(a, inverse=True) == inv).all()
    assert (O.i_ntt_381(fwd) == a).all()


def test_dft_simple_vectors_and_errors():
    one_hot = frs([1, 0, 0, 0, 0, 0, 0, 0])
    assert ints(O.ntt_381(one_hot)) == [1] * 8                       # DFT of delta_0
    assert ints(O.ntt_381(frs([3, 3]))) == [6, 0]                    # setup.rs:128-135 input
    with pytest.raises(AssertionError):
        O.ntt_381(frs([1, 2, 3]))
    with pytest.raises(AssertionError):
        O.i_ntt_381(frs([1, 2, 3, 4, 5, 6]))


@pytest.mark.parametrize("logn", [8, 12])
def test_ntt_fast_larger_sizes(logn):
    n = 1 << logn
    a = O.splitmix_scalars(n, 0xF40000 + logn)
    f = O.ntt_fast(a)
    assert (O.ntt_fast(f, inverse=True) == a).all()
    assert (O.ntt_fast(a, threads=4) == f).all()
    vals, w = ints(a), M.omega(n)
    for x in (0, 1, n // 2 + 3, n - 1):
        want = sum(v * pow(w, x * y, Q) for y, v in enumerate(vals)) % Q
        assert O.fr_to_int(f[x]) == want
    if logn == 8:
        assert (O.ntt_381(a) == f).all()


# ------------------------------------------------------------------ Polynomial (polynomial.rs tests :386-521)
def test_poly_add_sub_scalar_mul_literals():
    assert ints(O.poly_binop("poly_add", frs([1, 2, 3]), frs([4, 5, 6]), 1)) == [5, 7, 9]
    assert ints(O.poly_binop("poly_add", frs([1, 2, 3]), frs([4, 5]), 1)) == [5, 7, 3]
    assert ints(O.poly_binop("poly_sub", frs([4, 5, 6]), frs([1, 2]), 1)) == [3, 3, 6]
    assert ints(O.poly_binop("poly_sub", frs([1]), frs([1, 2]), 1)) == [0, Q - 2]
    with pytest.raises(AssertionError):
        O.poly_binop("poly_add", frs([1, 2, 3]), frs([4, 5]), 0)     # Lagrange needs equal lengths
    out, a, s = O.u64((3, 4)), frs([1, 2, 3]), fr(2)
    O.lib.poly_mul_scalar(out.ctypes.data, a.ctypes.data, 3, s.ctypes.data)
    assert ints(out) == [2, 4, 6]
    O.lib.poly_add_scalar(out.ctypes.data, a.ctypes.data, 3, s.ctypes.data, 1)
    assert ints(out) == [3, 2, 3]
    O.lib.poly_add_scalar(out.ctypes.data, a.ctypes.data, 3, s.ctypes.data, 0)
    assert ints(out) == [3, 4, 5]
    O.lib.poly_sub_scalar(out.ctypes.data, a.ctypes.data, 3, s.ctypes.data, 1)
    assert ints(out) == [Q - 1, 2, 3]
    O.lib.poly_sub_scalar(out.ctypes.data, a.ctypes.data, 3, s.ctypes.data, 0)
    assert ints(out) == [3, 4, 5]                                    # reference quirk: Lagrange branch adds
    O.lib.poly_shift_left(out.ctypes.data, a.ctypes.data, 3, 1)
    assert ints(out) == [2, 3, 1]


def test_poly_mul_literals_and_fast():
    """polynomial.rs:437-451 (1+x)^2 = [1,2,1]; faithful == fast == schoolbook"""
    one_x = frs([1, 1])
    assert ints(O.poly_binop("poly_mul", one_x, one_x)) == [1, 2, 1]
    rnd = random.Random(3)
    for na, nb in ((1, 1), (3, 2), (5, 8), (9, 9)):
        a, b = [rnd.randrange(Q) for _ in range(na)], [rnd.randrange(Q) for _ in range(nb)]
        want = [0] * (na + nb - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                want[i + j] = (want[i + j] + x * y) % Q
        assert ints(O.poly_binop("poly_mul", frs(a), frs(b))) == want
        assert ints(O.poly_binop("poly_mul_fast", frs(a), frs(b))) == want


def test_poly_eval():
    rnd = random.Random(4)
    c, x = [rnd.randrange(Q) for _ in range(11)], rnd.randrange(Q)
    want = sum(v * pow(x, i, Q) for i, v in enumerate(c)) % Q
    assert O.fr_to_int(O.poly_eval(frs(c), fr(x))) == want
    assert O.fr_to_int(O.poly_eval(frs(c), fr(x), fast=True)) == want
    assert O.fr_to_int(O.poly_eval(frs([1, 3, 2]), fr(2))) == 15       # polynomial.rs:34-45 comment example


def test_poly_div_exact_and_quirk():
    """polynomial.rs:314-380"""
    # (3x^3 - x^2 - x - 1) / (x - 1) = 3x^2 + 2x + 1
    assert ints(O.poly_binop("poly_div", frs([-1, -1, -1, 3]), frs([-1, 1]))) == [1, 2, 3]
    # trailing zeros are trimmed on both operands first
    assert ints(O.poly_binop("poly_div", frs([-1, -1, -1, 3, 0, 0]), frs([-1, 1, 0]))) == [1, 2, 3]
    # remainder discarded: (x^2 + 1) / (x + 1) -> quotient x - 1
    assert ints(O.poly_binop("poly_div", frs([1, 0, 1]), frs([1, 1]))) == [Q - 1, 1]
    # quirk: (x^4 - 1)/(x^2 - 1) has true quotient x^2 + 1 = [1,0,1]; the reference returns [1,1]
    assert ints(O.poly_binop("poly_div", frs([-1, 0, 0, 0, 1]), frs([-1, 0, 1]))) == [1, 1]
    # dividend shorter than divisor -> empty quotient; zero dividend -> empty
    assert ints(O.poly_binop("poly_div", frs([5]), frs([1, 1]))) == []
    assert ints(O.poly_binop("poly_div", frs([0, 0]), frs([1, 1]))) == []
    with pytest.raises(AssertionError):
        O.poly_binop("poly_div", frs([1, 2]), frs([0, 0]))           # zero divisor panics
    # random exact division, no zero quotient coefficients => equals true quotient
    rnd = random.Random(6)
    qt, dv = [rnd.randrange(1, Q) for _ in range(9)], [rnd.randrange(1, Q) for _ in range(4)]
    prod = ints(O.poly_binop("poly_mul_fast", frs(qt), frs(dv)))
    assert ints(O.poly_binop("poly_div", frs(prod), frs(dv))) == qt

"""CPU check of the product's __host__ __device__ arithmetic headers (bigint/fields/g1 .hpp): the same
templates the kernels instantiate are compiled for the host (tests/hostcheck) with the device's 32-bit
column multiplier selected, and compared with the oracle.  Catches logic errors before any GPU time."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from tests.bigint_model import P, Q

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "hostcheck", "libhostcheck.so")


@pytest.fixture(scope="module")
def hc():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    subprocess.check_call(["make", "-C", os.path.join(HERE, "hostcheck"), "-s"])
    lib = C.CDLL(SO)
    for name in dir(lib):
        pass
    return lib


def call(lib, name, out_words, *ins):
    out = np.zeros(out_words, dtype=np.uint32)
    keep = [np.ascontiguousarray(x).view(np.uint32) if isinstance(x, np.ndarray) else x for x in ins]
    args = [out.ctypes.data] + [k.ctypes.data if isinstance(k, np.ndarray) else k for k in keep]
    fn = getattr(lib, name)
    fn.restype = None
    fn.argtypes = [C.c_void_p] + [C.c_void_p if isinstance(k, np.ndarray) else (C.c_uint32 if i == 1 else C.c_int)
                                  for i, k in enumerate(keep)]
    fn(*args)
    return out.view(np.uint64)


def test_fr_ops(hc):
    rnd = random.Random(11)
    vals = [0, 1, Q - 1, Q - 2, 2**255 % Q] + [rnd.randrange(Q) for _ in range(60)]
    for i in range(0, len(vals) - 1):
        a, b = O.fr_from_int(vals[i]), O.fr_from_int(vals[i + 1])
        assert (call(hc, "hc_fr_mul", 8, a, b) == O.fr_bin("fr_mul", a, b)).all()
        assert (call(hc, "hc_fr_add", 8, a, b) == O.fr_bin("fr_add", a, b)).all()
        assert (call(hc, "hc_fr_sub", 8, a, b) == O.fr_bin("fr_sub", a, b)).all()
        assert (call(hc, "hc_fr_neg", 8, a) == O.fr_un("fr_neg", a)).all()
        assert O.unlimbs(call(hc, "hc_fr_from_mont", 8, a)) == vals[i]
    # host inversion (binary Euclid on 64-bit limbs, fields.hpp) against the oracle's power, edge values included; 0 -> 0
    for v in vals + [2, 3, Q // 2, 2**64, 2**128 - 1, 2**254 % Q]:
        a = O.fr_from_int(v)
        assert (call(hc, "hc_fr_inv", 8, a) == O.fr_invert(a)[0]).all(), v
    # non-canonical limbs (a caller's record): multiples of q are the zero they represent -- 0 -> 0, and the loop terminates
    for k in (1, 2):
        assert not call(hc, "hc_fr_inv", 8, np.array(O.limbs(k * Q, 4), dtype=np.uint64)).any()
    assert (call(hc, "hc_fr_inv", 8, np.array(O.limbs(Q + 5, 4), dtype=np.uint64)) == call(hc, "hc_fr_inv", 8, np.array(O.limbs(5, 4), dtype=np.uint64))).all()


def test_fp_ops(hc):
    rnd = random.Random(12)
    vals = [0, 1, P - 1, P - 2, (P - 1) // 2] + [rnd.randrange(P) for _ in range(60)]
    for i in range(0, len(vals) - 1):
        a, b = O.fp_from_int(vals[i]), O.fp_from_int(vals[i + 1])
        assert (call(hc, "hc_fp_mul", 12, a, b) == O.fp_bin("fp_mul", a, b)).all()
        assert (call(hc, "hc_fp_add", 12, a, b) == O.fp_bin("fp_add", a, b)).all()
        assert (call(hc, "hc_fp_sub", 12, a, b) == O.fp_bin("fp_sub", a, b)).all()
        assert (call(hc, "hc_fp_neg", 12, a) == O.fp_un("fp_neg", a)).all()
    for v in vals + [2, 3, P // 2, 2**64, 2**192 - 1, 2**380]:
        a = O.fp_from_int(v)
        assert (call(hc, "hc_fp_inv", 12, a) == O.fp_invert(a)[0]).all(), v
    for k in (1, 2, 9):                                   # 9p < 2^384: the largest multiple twelve limbs can hold
        assert not call(hc, "hc_fp_inv", 12, np.array(O.limbs(k * P, 6), dtype=np.uint64)).any()
    assert (call(hc, "hc_fp_inv", 12, np.array(O.limbs(P + 5, 6), dtype=np.uint64)) == call(hc, "hc_fp_inv", 12, np.array(O.limbs(5, 6), dtype=np.uint64))).all()


def test_g1_ops(hc):
    rnd = random.Random(13)
    g = O.g1_generator()
    pts = [g, O.g1_identity(), O.g1_double(g)] + [O.g1_mul(g, O.fr_from_int(rnd.randrange(Q))) for _ in range(4)]
    for a in pts:
        assert O.g1_eq(call(hc, "hc_g1_double", 36, a), O.g1_double(a))
        for b in pts:
            got = call(hc, "hc_g1_add", 36, a, b)
            assert (got == O.g1_add(a, b)).all()                       # same formulas => same projective limbs
            baff = O.g1_to_affine(b)
            dev_aff = np.zeros(12, dtype=np.uint64) if baff[12] else baff[:12].copy()   # device identity = (0,0)
            got = call(hc, "hc_g1_add_mixed", 36, a, dev_aff)
            assert O.g1_eq(got, O.g1_add(a, b))
    # P + (-P), P + P through the mixed formula
    gaff = O.g1_to_affine(g)[:12].copy()
    assert O.g1_eq(call(hc, "hc_g1_add_mixed", 36, g, gaff), O.g1_double(g))
    assert O.lib.g1_is_identity(call(hc, "hc_g1_add_mixed", 36, O.g1_neg(g), gaff).ctypes.data)
    k = rnd.randrange(Q)
    kc = np.array(O.limbs(k, 4), dtype=np.uint64)
    assert O.g1_eq(call(hc, "hc_g1_mul_scalar", 36, g, kc), O.g1_mul(g, O.fr_from_int(k)))
    for small, bits in ((0, 0), (1, 1), (5, 3), (0x7ABC, 15)):
        fn = hc.hc_g1_mul_small
        out = np.zeros(36, dtype=np.uint32)
        gg = g.copy()
        fn.restype, fn.argtypes = None, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
        fn(out.ctypes.data, gg.ctypes.data, small, bits)
        assert O.g1_eq(out.view(np.uint64), O.g1_mul(g, O.fr_from_int(small)))
    aff = call(hc, "hc_g1_to_affine", 24, pts[4])
    assert (aff == O.g1_to_affine(pts[4])[:12]).all()
    assert (call(hc, "hc_g1_to_affine", 24, O.g1_identity()) == 0).all()


def test_fp28_unsaturated_field(hc):
    """fp28.hpp: 14 x 28-bit lazy limbs, R' = 2^392 -- multiply and domain conversions vs the oracle"""
    rnd = random.Random(14)
    vals = [0, 1, P - 1, P - 2, (P - 1) // 2, 2**380, 2**381 - 1] + [rnd.randrange(P) for _ in range(80)]
    for i in range(len(vals) - 1):
        a, b = O.fp_from_int(vals[i]), O.fp_from_int(vals[i + 1])
        assert (call(hc, "hc_fp28_roundtrip", 12, a) == a).all()
        assert (call(hc, "hc_fp28_mul", 12, a, b) == O.fp_bin("fp_mul", a, b)).all()


def test_g1_28_mixed_add(hc):
    """g1_28.hpp: complete mixed addition on lazy limbs, incl. identity / doubling / inverse cases and long chains"""
    rnd = random.Random(15)
    g = O.g1_generator()

    def dev_aff(p):
        a = O.g1_to_affine(p)
        return np.zeros(12, dtype=np.uint64) if a[12] else a[:12].copy()

    def add28(acc, pt, neg=0, reps=1):
        fn = hc.hc_g1_28_add_mixed
        fn.restype, fn.argtypes = None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        out = np.zeros(36, dtype=np.uint32)
        a, q = np.ascontiguousarray(acc), np.ascontiguousarray(pt)
        fn(out.ctypes.data, a.ctypes.data, q.ctypes.data, neg, reps)
        return out.view(np.uint64)

    ident = np.zeros(36, dtype=np.uint32)
    hc.hc_g1_28_identity.restype, hc.hc_g1_28_identity.argtypes = None, [C.c_void_p]
    hc.hc_g1_28_identity(ident.ctypes.data)
    assert (ident.view(np.uint64) == O.g1_identity()).all()
    pts = [g, O.g1_identity(), O.g1_double(g)] + [O.g1_mul(g, O.fr_from_int(rnd.randrange(Q))) for _ in range(5)]
    for a in pts:
        for b in pts[:1] + pts[2:]:                                   # the kernel never feeds an identity point
            assert O.g1_eq(add28(a, dev_aff(b)), O.g1_add(a, b))
            assert O.g1_eq(add28(a, dev_aff(b), neg=1), O.g1_add(a, O.g1_neg(b)))
    assert O.g1_eq(add28(g, dev_aff(g)), O.g1_double(g))              # P + P
    assert O.lib.g1_is_identity(add28(g, dev_aff(g), neg=1).ctypes.data)   # P - P
    # bounds stay stable over a long run: 200 repeated additions = 200 * P
    k = 200
    assert O.g1_eq(add28(O.g1_identity(), dev_aff(pts[4]), reps=k), O.g1_mul(pts[4], O.fr_from_int(k)))
    assert O.g1_eq(add28(pts[5], dev_aff(pts[4]), neg=1, reps=k), O.g1_add(pts[5], O.g1_neg(O.g1_mul(pts[4], O.fr_from_int(k)))))


def test_g1_28_add_double_mul_small(hc):
    """g1_28.hpp: complete projective add / double / small multiples on lazy limbs (used by fix-up and reduce)"""
    rnd = random.Random(16)
    g = O.g1_generator()
    pts = [g, O.g1_identity(), O.g1_double(g)] + [O.g1_mul(g, O.fr_from_int(rnd.randrange(Q))) for _ in range(4)]

    def run(name, argtypes, *args):
        fn = getattr(hc, name)
        fn.restype, fn.argtypes = None, [C.c_void_p] + argtypes
        out = np.zeros(36, dtype=np.uint32)
        keep = [np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a for a in args]
        fn(out.ctypes.data, *[k.ctypes.data if isinstance(k, np.ndarray) else k for k in keep])
        return out.view(np.uint64)

    for a in pts:
        assert O.g1_eq(run("hc_g1_28_double", [C.c_void_p, C.c_int], a, 1), O.g1_double(a))
        for b in pts:
            assert O.g1_eq(run("hc_g1_28_add", [C.c_void_p, C.c_void_p, C.c_int], a, b, 1), O.g1_add(a, b))
        assert O.g1_eq(run("hc_g1_28_add", [C.c_void_p, C.c_void_p, C.c_int], a, O.g1_neg(a), 1), O.g1_identity())
    # the cooperative (lane-split) form of the same addition: every pair incl. identity / doubling / inverse, and a long chain
    for a in pts:
        for b in pts + [O.g1_neg(a)]:
            assert O.g1_eq(run("hc_g1_28_add_coop", [C.c_void_p, C.c_void_p, C.c_int], a, b, 1), O.g1_add(a, b))
    assert O.g1_eq(run("hc_g1_28_add_coop", [C.c_void_p, C.c_void_p, C.c_int], pts[3], pts[4], 150),
                   O.g1_add(pts[3], O.g1_mul(pts[4], O.fr_from_int(150))))
    # long chains keep the bounds: a + 150 b, 2^40 a
    a, b = pts[3], pts[4]
    assert O.g1_eq(run("hc_g1_28_add", [C.c_void_p, C.c_void_p, C.c_int], a, b, 150), O.g1_add(a, O.g1_mul(b, O.fr_from_int(150))))
    assert O.g1_eq(run("hc_g1_28_double", [C.c_void_p, C.c_int], a, 40), O.g1_mul(a, O.fr_from_int(1 << 40)))
    for k, bits in ((0, 0), (1, 1), (6, 3), (0x7FFF, 15), (0x4001, 15)):
        assert O.g1_eq(run("hc_g1_28_mul_small", [C.c_void_p, C.c_uint32, C.c_int], a, k, bits), O.g1_mul(a, O.fr_from_int(k)))
    isid = hc.hc_g1_28_is_identity
    isid.restype, isid.argtypes = C.c_int, [C.c_void_p]
    ident, gg = O.g1_identity(), g.copy()
    assert isid(ident.ctypes.data) == 1 and isid(gg.ctypes.data) == 0


def test_fr29_lazy_butterflies(hc):
    """fr29.hpp: 9 x 29-bit limbs, data kept in the reference's 2^256 Montgomery domain, twiddles in 2^261"""
    rnd = random.Random(17)
    fn = hc.hc_fr29_butterfly
    fn.restype, fn.argtypes = None, [C.c_void_p] * 5 + [C.c_int]
    vals = [0, 1, Q - 1, Q - 2, 2**254] + [rnd.randrange(Q) for _ in range(40)]
    for i in range(len(vals) - 2):
        u, v, w = (O.fr_from_int(x) for x in vals[i:i + 3])
        assert (call(hc, "hc_fr29_roundtrip", 8, u) == u).all()
        for reps in (1, 2, 7):
            ru, rv = np.zeros(8, dtype=np.uint32), np.zeros(8, dtype=np.uint32)
            fn(ru.ctypes.data, rv.ctypes.data, u.ctypes.data, v.ctypes.data, w.ctypes.data, reps)
            eu, ev = u, v
            for _ in range(reps):
                eu, ev = O.fr_bin("fr_add", eu, ev), O.fr_bin("fr_mul", O.fr_bin("fr_sub", eu, ev), w)
            assert (ru.view(np.uint64) == eu).all() and (rv.view(np.uint64) == ev).all()


def test_fr29_radix4_group_lazy_sums(hc):
    """fr29_radix4 (two stages on four elements, first-stage sums left unreduced) against big integers and against four
    fr29_butterfly calls, on inputs anywhere in [0, 2q) -- the kernels' inter-stage invariant -- including the extremes; the
    host build aborts if a limb or value bound of fr29_mul is violated (BP_FR29_CHECK)"""
    rnd = random.Random(29)
    fn = hc.hc_fr29_radix4
    fn.restype, fn.argtypes = None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    top = 2 * Q - 1
    # limb-extreme values below 2q: all 29-bit limbs full below the top one
    full = [(k << 232) - 1 for k in (1, 2, (2 * Q) >> 232)] + [((2 * Q) >> 232 << 232) + 5]
    special = [0, 1, Q - 1, Q, Q + 1, top, top - 1, 2 ** 255, 2 ** 254 - 1] + [v for v in full if v < 2 * Q]
    cases = [[top] * 4, [0] * 4, [top, 0, top, 0], [0, top, 0, top], [top, top, 0, 0], [0, 0, top, top], [Q, Q, Q, Q], [full[0]] * 4]
    cases += [[rnd.choice(special) for _ in range(4)] for _ in range(150)]
    cases += [[rnd.randrange(2 * Q) for _ in range(4)] for _ in range(150)]
    wsp = [1, Q - 1, 2, (Q + 1) // 2, 2 ** 254]
    for a in cases:
        w = [rnd.choice(wsp) if rnd.random() < 0.3 else rnd.randrange(1, Q) for _ in range(3)]          # true twiddle values
        a_arr = np.array([[(v >> (32 * j)) & 0xFFFFFFFF for j in range(8)] for v in a], dtype=np.uint32)
        w_arr = np.array([[((x << 256) % Q >> (32 * j)) & 0xFFFFFFFF for j in range(8)] for x in w], dtype=np.uint32)   # Montgomery form
        got = {}
        for lazy in (1, 0):
            out = np.zeros((4, 8), dtype=np.uint32)
            fn(out.ctypes.data, a_arr.ctypes.data, w_arr.ctypes.data, lazy)
            got[lazy] = [sum(int(out[i, j]) << (32 * j) for j in range(8)) for i in range(4)]
        d02, d13 = (a[0] - a[2]) * w[0], (a[1] - a[3]) * w[1]
        want = [(a[0] + a[1] + a[2] + a[3]) % Q, ((a[0] + a[2]) - (a[1] + a[3])) * w[2] % Q, (d02 + d13) % Q, (d02 - d13) * w[2] % Q]
        assert got[1] == want, (a, w)
        assert got[0] == want, (a, w)

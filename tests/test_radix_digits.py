"""The digit radix of fixed-base MSM tables (baby_plonk_rust_amd/csrc/msm_digits.hpp, round 6): the host's choice of R per table width and
the very lines the kernels cut a scalar into radix-R digits with, run on the CPU (tests/hostcheck) against Python integers.
Replaces nothing in the reference by itself -- src/msm.rs:119-139 cuts 4-bit windows -- but every bucket method yields the same group
element; what must hold is sum_w d_w R^w == k with |d_w| <= R / 2 for EVERY k < q, which is what this file checks."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from tests.bigint_model import Q

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "hostcheck", "libhostcheck.so")


@pytest.fixture(scope="module")
def hc():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    subprocess.check_call(["make", "-C", os.path.join(HERE, "hostcheck"), "-s"])
    lib = C.CDLL(SO)
    lib.hc_radix_info.restype = None
    lib.hc_radix_info.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    lib.hc_radix_digits.restype = C.c_int
    lib.hc_radix_digits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    return lib


def windows_of(c):
    """W as make_plan (msm.hip) finds it for power-of-two windows: ceil(255 / c), one more while the unsigned top window cannot hold
    the largest scalar's top digit plus the carry of the signed ones below"""
    W = max(2, (255 + c - 1) // c)
    while True:
        bias = sum(1 << (c * w + c - 1) for w in range(W - 1))
        if c * (W - 1) < 288 and ((Q - 1 + bias) >> (c * (W - 1))) <= 1 << (c - 1):
            return W
        W += 1


def info(hc, c, W):
    out = np.zeros(17, dtype=np.uint32)
    hc.hc_radix_info(out.ctypes.data, c, W)
    R = int(out[0])
    m = sum(int(v) << (32 * i) for i, v in enumerate(out[1:9]))
    bias = sum(int(v) << (32 * i) for i, v in enumerate(out[9:17]))
    return R, m, bias


def test_radix_choice_per_width(hc):
    used = {}
    for c in range(4, 25):
        W = windows_of(c)
        R, m, bias = info(hc, c, W)
        if R == 0:
            continue
        used[c] = (W, R)
        assert R % 2 == 0 and R <= 0.9 * 2 ** c
        assert 2 ** 256 < R ** W < 1.9 * 2 ** 256
        assert (R - 2) ** W <= 2 ** 256 * 1.02 ** W                      # within 2 % of the smallest admissible radix
        assert m == -(-2 ** 512 // R ** W) and m < 2 ** 256                # ceil
        assert bias == sum((R // 2) * R ** w for w in range(W - 1))
        # the largest scalar keeps its (unsigned) top digit inside the live buckets
        assert (Q - 1 + bias) // R ** (W - 1) <= R // 2
    # (msm_radix_compute answers for every width; the library's make_plan takes the radix from 21 bits only -- at 20 bits the 2^19-bucket
    # tree runs every wide level in one wave round and gains nothing from fewer live buckets, profiles/r06_tail_ab.txt)
    # the widths that waste the most bucket range: 20 bits (thirteen windows), 22 (twelve)
    assert used[20] == (13, 0xD0000) and used[20][1] // 2 == 425984       # against 524 288 buckets at 2^19
    assert used[22][0] == 12 and used[22][1] // 2 < 0.64 * 2 ** 21
    assert 16 not in used and used[17][0] == 16                              # sixteen 16-bit windows waste nothing: power-of-two windows stay; 17 bits need 16 windows too and get R just above 2^16
    print({c: (W, hex(R), R // 2) for c, (W, R) in used.items()})


def digits_of(hc, k, c, W):
    d = np.zeros(W, dtype=np.int32)
    kk = np.array([(k >> (32 * i)) & 0xFFFFFFFF for i in range(8)], dtype=np.uint32)
    ok = hc.hc_radix_digits(d.ctypes.data, kk.ctypes.data, c, W)
    return ok, [int(v) for v in d]


@pytest.mark.parametrize("c", [13, 14, 18, 19, 20, 21, 22, 23, 24])
def test_radix_digits_reconstruct_every_kind_of_scalar(hc, c):
    W = windows_of(c)
    R, m, bias = info(hc, c, W)
    if R == 0:
        pytest.skip("power-of-two windows at this width")
    rnd = random.Random(c)
    ks = [0, 1, 2, R // 2 - 1, R // 2, R // 2 + 1, R - 1, R, R + 1, Q - 1, Q - 2, Q // 2, Q // 3, 2 ** 254, 2 ** 254 - 1, 2 ** 200 + 1]
    for j in range(1, W):                                   # exact multiples of R^j and their neighbours: the digit boundaries of the fixed-point division
        for t in (1, R // 2, R // 2 + 1, R - 1, rnd.randrange(1, R)):
            for e in (-2, -1, 0, 1, 2):
                ks.append((t * R ** j + e) % Q)
    for j in range(W):                                      # every digit at +-R/2, all others zero; long runs of extreme digits
        ks.append(((R // 2) * R ** j) % Q)
        ks.append((sum((R // 2) * R ** i for i in range(j + 1))) % Q)
        ks.append((sum((R // 2 - 1) * R ** i for i in range(j + 1))) % Q)
    ks += [rnd.randrange(Q) for _ in range(700)]
    ks += [rnd.randrange(2 ** rnd.randrange(1, 255)) for _ in range(150)]      # short scalars
    for k in ks:
        ok, d = digits_of(hc, k, c, W)
        assert ok == 1, hex(k)
        assert all(abs(v) <= R // 2 for v in d), (hex(k), d)
        assert d[W - 1] >= 0
        assert sum(v * R ** w for w, v in enumerate(d)) == k, (hex(k), d)
    # not scalars (>= q: a canonical-bytes input the status word rejects): either no entries at all, or digits that stay inside the
    # live buckets -- never an index beyond them
    for k in [Q, Q + 1, 2 ** 255, 2 ** 256 - 1, 2 ** 256 - 2 ** 200] + [rnd.randrange(Q, 2 ** 256) for _ in range(150)]:
        ok, d = digits_of(hc, k, c, W)
        assert ok in (0, 1)
        if ok:
            assert all(abs(v) <= R // 2 for v in d)

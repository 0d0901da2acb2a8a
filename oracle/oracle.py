"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.

Array conventions (numpy, C-contiguous, dtype uint64):
  Fr           [..., 4]   Montgomery limbs (scalar.rs:16-22)
  Fp           [..., 6]   Montgomery limbs (fp.rs:11-15)
  G1Projective [..., 18]  x | y | z       (g1.rs:442-446)
  G1Affine     [..., 13]  x | y | infinity (g1.rs:28-32)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h")) or f == "Makefile"]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _SO


def _load():
    build()
    lib = C.CDLL(_SO)
    vp, sz, u64, i32 = C.c_void_p, C.c_size_t, C.c_uint64, C.c_int
    sig = {
        "fr_zero": (None, [vp]), "fr_one": (None, [vp]), "fp_zero": (None, [vp]), "fp_one": (None, [vp]),
        "fr_add": (None, [vp, vp, vp]), "fr_sub": (None, [vp, vp, vp]), "fr_neg": (None, [vp, vp]),
        "fr_mul": (None, [vp, vp, vp]), "fr_square": (None, [vp, vp]),
        "fr_pow": (None, [vp, vp, vp]), "fr_pow_vartime": (None, [vp, vp, vp]), "fr_invert": (i32, [vp, vp]),
        "fr_from_u64": (None, [vp, u64]), "fr_from_raw": (None, [vp, vp]), "fr_from_bytes": (i32, [vp, vp]),
        "fr_to_bytes": (None, [vp, vp]), "fr_from_bytes_wide": (None, [vp, vp]), "fr_from_u512": (None, [vp, vp]),
        "fp_add": (None, [vp, vp, vp]), "fp_sub": (None, [vp, vp, vp]), "fp_neg": (None, [vp, vp]),
        "fp_mul": (None, [vp, vp, vp]), "fp_square": (None, [vp, vp]), "fp_invert": (i32, [vp, vp]),
        "fp_sqrt": (i32, [vp, vp]), "fp_from_bytes": (i32, [vp, vp]), "fp_to_bytes": (None, [vp, vp]),
        "fp_lexicographically_largest": (i32, [vp]),
        "g1_affine_identity": (None, [vp]), "g1_affine_generator": (None, [vp]),
        "g1_identity": (None, [vp]), "g1_generator": (None, [vp]), "g1_is_identity": (i32, [vp]),
        "g1_is_on_curve": (i32, [vp]), "g1_eq": (i32, [vp, vp]), "g1_neg": (None, [vp, vp]),
        "g1_double": (None, [vp, vp]), "g1_add": (None, [vp, vp, vp]), "g1_add_mixed": (None, [vp, vp, vp]),
        "g1_mul": (None, [vp, vp, vp]), "g1_to_affine": (None, [vp, vp]), "g1_from_affine": (None, [vp, vp]),
        "g1_batch_normalize": (None, [vp, vp, sz]), "g1_to_uncompressed": (None, [vp, vp]),
        "g1_from_uncompressed_unchecked": (i32, [vp, vp]), "g1_to_compressed": (None, [vp, vp]),
        "g1_from_compressed_unchecked": (i32, [vp, vp]),
        "msm_get_c_bit_chunk": (u64, [vp, sz, sz]), "msm_c_bit_msm": (None, [vp, vp, vp, sz, sz]),
        "msm_bucket_msm": (None, [vp, vp, sz, vp, sz, sz, sz]),
        "msm_bucket_msm_mt": (None, [vp, vp, sz, vp, sz, sz, sz, i32]),
        "ntt_root_of_unity": (None, [vp, u64]), "ntt_roots_of_unity": (None, [vp, u64]),
        "ntt_find_next_power_of_two": (sz, [sz, sz]),
        "ntt_381": (i32, [vp, vp, sz]), "i_ntt_381": (i32, [vp, vp, sz]),
        "ntt_fast": (i32, [vp, sz, i32]), "ntt_fast_mt": (i32, [vp, sz, i32, i32]),
        "poly_coeffs_evaluate": (None, [vp, vp, sz, vp]), "poly_coeffs_evaluate_fast": (None, [vp, vp, sz, vp]),
        "poly_shift_left": (None, [vp, vp, sz, sz]),
        "poly_add_scalar": (None, [vp, vp, sz, vp, i32]), "poly_sub_scalar": (None, [vp, vp, sz, vp, i32]),
        "poly_mul_scalar": (None, [vp, vp, sz, vp]),
        "poly_add": (sz, [vp, vp, sz, vp, sz, i32]), "poly_sub": (sz, [vp, vp, sz, vp, sz, i32]),
        "poly_mul": (sz, [vp, vp, sz, vp, sz]), "poly_mul_fast": (sz, [vp, vp, sz, vp, sz]),
        "poly_div": (sz, [vp, vp, sz, vp, sz]),
        "prover_round2_z": (i32, [vp, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp]),
        "oracle_splitmix_scalars": (None, [vp, sz, u64]),
        "oracle_points_progression": (None, [vp, sz, vp, vp]),
        "oracle_points_to_bytes96": (None, [vp, vp, sz]), "oracle_proj_from_bytes96": (None, [vp, vp, sz]),
        "oracle_dot_progression": (None, [vp, vp, sz, vp, vp]),
        "oracle_now": (C.c_double, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()
ERR = (1 << 64) - 1  # (size_t)-1


def _p(a):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def u64(shape):
    return np.zeros(shape, dtype=np.uint64)


def arr(x, width=None):
    a = np.ascontiguousarray(np.array(x, dtype=np.uint64))
    if width is not None:
        assert a.shape[-1] == width, a.shape
    return a


# ---------------------------------------------------------------- Fr
Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB


def limbs(v, n):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def unlimbs(a):
    return sum(int(x) << (64 * i) for i, x in enumerate(a))


def fr_from_int(v):
    """canonical integer -> Montgomery limbs (via from_raw, scalar.rs:343)"""
    return _call("fr_from_raw", u64(4), limbs(v % Q, 4))


def fr_to_int(a):
    b, a = np.zeros(32, dtype=np.uint8), np.ascontiguousarray(a, dtype=np.uint64)
    lib.fr_to_bytes(_p(b), _p(a))
    return int.from_bytes(bytes(b), "little")


def fr_array_from_ints(vals):
    out = u64((len(vals), 4))
    for i, v in enumerate(vals):
        out[i] = fr_from_int(v)
    return out


def fr_array_to_ints(a):
    return [fr_to_int(a[i]) for i in range(a.shape[0])]


def _call(name, out, *ins):
    """call lib.<name>(out, *ins) keeping the converted temporaries alive across the call"""
    keep = [arr(x) for x in ins]
    res = getattr(lib, name)(_p(out), *[_p(k) for k in keep])
    return out if res is None else (out, res)


def fr_bin(name, a, b):
    return _call(name, u64(4), a, b)


def fr_un(name, a):
    return _call(name, u64(4), a)


def fp_from_int(v):
    out = u64(6)
    b = np.frombuffer((v % P).to_bytes(48, "big"), dtype=np.uint8).copy()
    assert lib.fp_from_bytes(_p(out), _p(b)) == 1
    return out


def fp_to_int(a):
    b, a = np.zeros(48, dtype=np.uint8), np.ascontiguousarray(a, dtype=np.uint64)
    lib.fp_to_bytes(_p(b), _p(a))
    return int.from_bytes(bytes(b), "big")


def fp_bin(name, a, b):
    return _call(name, u64(6), a, b)


def fp_un(name, a):
    return _call(name, u64(6), a)


def fr_const(name):
    """FR_MODULUS / FR_R / FR_R2 / FR_R3 / FR_ROOT_OF_UNITY / FR_ROOT_OF_UNITY_INV as limb arrays"""
    return np.ctypeslib.as_array((C.c_uint64 * 4).in_dll(lib, name)).copy()


def fp_one():
    out = u64(6)
    lib.fp_one(_p(out))
    return out


def _bytes_arr(b):
    return np.frombuffer(bytes(b), dtype=np.uint8).copy()


def fp_from_bytes(b):
    out, buf = u64(6), _bytes_arr(b)
    return out, bool(lib.fp_from_bytes(_p(out), _p(buf)))


def fp_to_bytes(a):
    b, a = np.zeros(48, dtype=np.uint8), arr(a)
    lib.fp_to_bytes(_p(b), _p(a))
    return bytes(b)


def fp_invert(a):
    out, a = u64(6), arr(a)
    return out, bool(lib.fp_invert(_p(out), _p(a)))


def fp_sqrt(a):
    out, a = u64(6), arr(a)
    return out, bool(lib.fp_sqrt(_p(out), _p(a)))


def fp_lex_largest(a):
    a = arr(a)
    return bool(lib.fp_lexicographically_largest(_p(a)))


def fr_from_bytes(b):
    out, buf = u64(4), _bytes_arr(b)
    return out, bool(lib.fr_from_bytes(_p(out), _p(buf)))


def fr_to_bytes(a):
    b, a = np.zeros(32, dtype=np.uint8), arr(a)
    lib.fr_to_bytes(_p(b), _p(a))
    return bytes(b)


def fr_from_bytes_wide(b):
    out, buf = u64(4), _bytes_arr(b)
    lib.fr_from_bytes_wide(_p(out), _p(buf))
    return out


def fr_from_u512(l):
    return _call("fr_from_u512", u64(4), l)


def fr_invert(a):
    out, a = u64(4), arr(a)
    return out, bool(lib.fr_invert(_p(out), _p(a)))


# ---------------------------------------------------------------- G1
def g1_generator():
    out = u64(18)
    lib.g1_generator(_p(out))
    return out


def g1_identity():
    out = u64(18)
    lib.g1_identity(_p(out))
    return out


def g1_add(a, b):
    return _call("g1_add", u64(18), a, b)


def g1_double(a):
    return _call("g1_double", u64(18), a)


def g1_neg(a):
    return _call("g1_neg", u64(18), a)


def g1_mul(p, s):
    return _call("g1_mul", u64(18), p, s)


def g1_to_affine(p):
    return _call("g1_to_affine", u64(13), p)


def g1_from_affine(a):
    return _call("g1_from_affine", u64(18), a)


def g1_add_mixed(p, a):
    return _call("g1_add_mixed", u64(18), p, a)


def g1_eq(a, b):
    a, b = arr(a), arr(b)
    return bool(lib.g1_eq(_p(a), _p(b)))


def g1_to_uncompressed(aff):
    out, aff = np.zeros(96, dtype=np.uint8), arr(aff)
    lib.g1_to_uncompressed(_p(out), _p(aff))
    return bytes(out)


def g1_to_compressed(aff):
    out, aff = np.zeros(48, dtype=np.uint8), arr(aff)
    lib.g1_to_compressed(_p(out), _p(aff))
    return bytes(out)


def g1_from_uncompressed(b):
    out, buf = u64(13), np.frombuffer(b, dtype=np.uint8).copy()
    ok = lib.g1_from_uncompressed_unchecked(_p(out), _p(buf))
    return out, bool(ok)


def g1_from_compressed(b):
    out, buf = u64(13), np.frombuffer(b, dtype=np.uint8).copy()
    ok = lib.g1_from_compressed_unchecked(_p(out), _p(buf))
    return out, bool(ok)


def g1_bytes96(p):
    """projective -> 96-byte uncompressed affine encoding (g1.rs:246-260)"""
    return g1_to_uncompressed(g1_to_affine(p))


def proj_from_bytes96(buf):
    n = len(buf) // 96
    out, src = u64((n, 18)), np.frombuffer(bytes(buf), dtype=np.uint8).copy()
    lib.oracle_proj_from_bytes96(_p(out), _p(src), n)
    return out


# ---------------------------------------------------------------- MSM / DFT / poly
def bucket_msm(points, scalars, b=256, c=4, threads=1):
    """src/msm.rs:76-118; points [n,18], scalars [m,4]"""
    points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 18)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    out = u64(18)
    if threads > 1:
        lib.msm_bucket_msm_mt(_p(out), _p(points), len(points), _p(scalars), len(scalars), b, c, threads)
    else:
        lib.msm_bucket_msm(_p(out), _p(points), len(points), _p(scalars), len(scalars), b, c)
    return out


def ntt_381(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    out = u64(a.shape)
    if lib.ntt_381(_p(out), _p(a), len(a)) != 0:
        raise AssertionError("not a power of two (utils.rs:65)")
    return out


def i_ntt_381(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    out = u64(a.shape)
    if lib.i_ntt_381(_p(out), _p(a), len(a)) != 0:
        raise AssertionError("not a power of two (utils.rs:108)")
    return out


def ntt_fast(a, inverse=False, threads=1):
    out = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4).copy()
    if lib.ntt_fast_mt(_p(out), len(out), int(inverse), threads) != 0:
        raise AssertionError("not a power of two")
    return out


def splitmix_scalars(n, seed):
    out = u64((n, 4))
    lib.oracle_splitmix_scalars(_p(out), n, seed)
    return out


def points_progression(n, a_int, d_int):
    """P_i = (a + i d) G as affine [n,13]"""
    out, a, d = u64((n, 13)), fr_from_int(a_int), fr_from_int(d_int)
    lib.oracle_points_progression(_p(out), n, _p(a), _p(d))
    return out


def points_to_bytes96(aff):
    aff = np.ascontiguousarray(aff, dtype=np.uint64).reshape(-1, 13)
    out = np.zeros(96 * len(aff), dtype=np.uint8)
    lib.oracle_points_to_bytes96(_p(out), _p(aff), len(aff))
    return out


def affine_to_proj(aff):
    aff = np.ascontiguousarray(aff, dtype=np.uint64).reshape(-1, 13)
    out = u64((len(aff), 18))
    for i in range(len(aff)):
        out[i] = g1_from_affine(aff[i])
    return out


def poly_binop(name, a, b, basis=1):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    out = u64((max(len(a) + len(b), 1), 4))
    fn = getattr(lib, name)
    n = fn(_p(out), _p(a), len(a), _p(b), len(b), basis) if name in ("poly_add", "poly_sub") else \
        fn(_p(out), _p(a), len(a), _p(b), len(b))
    if n == ERR:
        raise AssertionError(name + ": reference panics here")
    return out[:n].copy()


def poly_eval(coeffs, x, fast=False):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    out, x = u64(4), arr(x)
    (lib.poly_coeffs_evaluate_fast if fast else lib.poly_coeffs_evaluate)(_p(out), _p(coeffs), len(coeffs), _p(x))
    return out


def dot_progression(scalars, a_int, d_int):
    """integer k = sum_i s_i (a + i d) mod q"""
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    out, a, d = u64(4), fr_from_int(a_int), fr_from_int(d_int)
    lib.oracle_dot_progression(_p(out), _p(scalars), len(scalars), _p(a), _p(d))
    return fr_to_int(out)


def round2_z(a, b, c, s1, s2, s3, beta, gamma, k1=None, k2=None):
    """src/prover.rs:279-319; all arrays [n,4] Montgomery; returns z values [n,4] or raises where the reference panics"""
    cols = [np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4) for x in (a, b, c, s1, s2, s3)]
    n = len(cols[0])
    out = u64((n, 4))
    k1 = fr_from_int(2) if k1 is None else arr(k1)
    k2 = fr_from_int(3) if k2 is None else arr(k2)
    beta, gamma = arr(beta), arr(gamma)
    rc = lib.prover_round2_z(_p(out), *[_p(x) for x in cols], n, _p(beta), _p(gamma), _p(k1), _p(k2))
    if rc != 0:
        raise AssertionError("round_2: " + ("zero denominator" if rc == -1 else "z_n != 1"))
    return out

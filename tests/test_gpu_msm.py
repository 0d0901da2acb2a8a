"""-m gpu: G1 MSM through the C ABI vs the oracle (src/msm.rs restatement), the reference's fixtures and
closed-form answers.  Bit-exact on the 96-byte affine encoding."""
import os
import random

import numpy as np
import pytest

import baby_plonk_rust_amd as bp
from oracle import oracle as O
from tests import bigint_model as M
from tests.gpu_common import NTHREADS, Q, closed_form, experiment, oracle_dot, progression_bytes

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)
UNCOMP = open(os.path.join(HERE, "golden", "g1_uncompressed_valid_test_vectors.dat"), "rb").read()


@pytest.fixture(scope="module")
def ctx():
    return bp.default_context()


def frs(vals):
    return bp.scalars_from_ints([v % Q for v in vals])


def test_golden_wire_vectors_as_srs(ctx):
    """the reference's 1000-point fixture (i*G, i = 0..999, point 0 = identity) as an SRS:
    round-trips through HBM, and sum s_i (i G) = (sum i s_i) G; also equals the oracle's bucket_msm"""
    h = ctx.srs_load(UNCOMP)
    assert ctx.srs_len(h) == 1000
    assert ctx.srs_export(h) == UNCOMP
    rnd = random.Random(21)
    sc = [rnd.randrange(Q) for _ in range(1000)]
    got = ctx.msm(h, frs(sc))
    assert got == M.enc96(M.ec_mul(sum(i * s for i, s in enumerate(sc))))
    assert got == O.g1_bytes96(O.bucket_msm(O.proj_from_bytes96(UNCOMP), O.fr_array_from_ints(sc), threads=NTHREADS))
    # canonical little-endian scalar format (Scalar::to_bytes)
    le = np.frombuffer(b"".join(s.to_bytes(32, "little") for s in sc), dtype=np.uint8).reshape(-1, 32)
    assert ctx.msm(h, le, fmt=bp.FR_BYTES_LE) == got
    ctx.srs_free(h)


def test_generate_srs_and_monomial_commit(ctx):
    """src/setup.rs:46-116 restated"""
    s = bp.Setup.generate_srs(8, 2, ctx)
    pts = s.powers_of_x()
    for i in range(8):
        assert pts[96 * i: 96 * i + 96] == M.enc96(M.ec_mul(pow(2, i, Q)))
    s10 = bp.Setup.generate_srs(2, 10, ctx)
    P = lambda v: bp.Polynomial(frs(v), bp.BASIS_MONOMIAL, ctx)
    assert s10.commit(P([2, 3])) == M.enc96(M.ec_mul(2 + 30))
    assert s.commit(P([0, 1])) == M.enc96(M.ec_mul(2))
    assert s.commit(P([0, 0, 1])) == M.enc96(M.ec_mul(4))
    lhs = M.ec_mul(2 - 1, M.ec_mul(1 + 2 * 2 + 3 * 4))                     # commit(3x^2+2x+1) * (tau - 1)
    assert s.commit(P([-1, -1, -1, 3])) == M.enc96(lhs)
    with pytest.raises(bp.BpError) as e:
        s.commit(bp.Polynomial(frs([1, 2]), bp.BASIS_LAGRANGE, ctx))        # setup.rs:34 assert_eq!
    assert e.value.code == -5
    # toy-proof SRS (tests/verify_proof_test.rs:16): 14 powers of tau = 101, vs the oracle's sequential *= tau
    s101 = bp.Setup.generate_srs(14, 101, ctx)
    cur, exp = O.g1_generator(), b""
    for _ in range(14):
        exp += O.g1_bytes96(cur)
        cur = O.g1_mul(cur, O.fr_from_int(101))
    assert s101.powers_of_x() == exp


def test_zip_truncation_and_edges(ctx):
    rnd = random.Random(22)
    a, d, n = rnd.randrange(Q), rnd.randrange(Q), 300
    pts = progression_bytes(n, a, d)
    h = ctx.srs_load(pts)
    sc = [rnd.randrange(Q) for _ in range(n)]
    assert ctx.msm(h, frs(sc)) == closed_form(sc, a, d)
    assert ctx.msm(h, frs(sc[:17])) == closed_form(sc[:17], a, d)                  # fewer scalars than points
    assert ctx.msm(h, frs(sc + sc)) == closed_form(sc, a, d)                       # more scalars than points (msm.rs:29)
    assert ctx.msm(h, frs([])) == M.enc96(None)
    assert ctx.msm(h, frs([0] * n)) == M.enc96(None)
    assert ctx.msm(h, frs([1] * n)) == closed_form([1] * n, a, d)
    assert ctx.msm(h, frs([Q - 1] * n)) == closed_form([Q - 1] * n, a, d)
    one_hot = [0] * n
    one_hot[123] = 2**254 + 12345
    assert ctx.msm(h, frs(one_hot)) == closed_form(one_hot, a, d)
    assert ctx.msm(h, frs([7])) == closed_form([7], a, d)
    small = [rnd.randrange(16) for _ in range(n)]                                  # witness-like small values
    assert ctx.msm(h, frs(small)) == closed_form(small, a, d)
    ctx.srs_free(h)
    # tau = 1 SRS (prover.rs:684): all points equal, every bucket add is a doubling
    s1 = bp.Setup.generate_srs(64, 1, ctx)
    assert ctx.msm(s1.handle, frs(sc[:64])) == M.enc96(M.ec_mul(sum(sc[:64])))
    # tau = 0: P_0 = G, the rest are the identity
    s0 = bp.Setup.generate_srs(5, 0, ctx)
    assert s0.powers_of_x() == M.enc96(M.ec_mul(1)) + M.enc96(None) * 4
    assert ctx.msm(s0.handle, frs(sc[:5])) == M.enc96(M.ec_mul(sc[0]))
    # P and -P in the same SRS with equal scalars cancel
    neg = M.ec_mul(Q - 5)
    hh = ctx.srs_load(M.enc96(M.ec_mul(5)) + M.enc96(neg))
    assert ctx.msm(hh, frs([99, 99])) == M.enc96(None)


def test_bad_inputs(ctx):
    with pytest.raises(bp.BpError) as e:
        ctx.srs_load(bytes([0x1F]) + bytes([0xFF] * 95))                 # x >= p
    assert e.value.code == -3
    with pytest.raises(bp.BpError) as e:
        ctx.srs_load((5).to_bytes(48, "big") + (7).to_bytes(48, "big"))  # not on the curve
    assert e.value.code == -3
    with pytest.raises(bp.BpError):
        ctx.msm(12345678, frs([1]))                                      # unknown handle
    h = ctx.srs_generate_progression(4, 3, 5)
    bad = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in (1, Q, 2, Q + 5)), dtype=np.uint8).reshape(-1, 32)
    with pytest.raises(bp.BpError) as e:
        ctx.msm(h, bad, fmt=bp.FR_BYTES_LE)                              # Scalar::from_bytes rejects >= q
    assert e.value.code == -4
    ok = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in (1, Q - 1, 2, 5)), dtype=np.uint8).reshape(-1, 32)
    assert ctx.msm(h, ok, fmt=bp.FR_BYTES_LE) == closed_form([1, Q - 1, 2, 5], 3, 5)
    ctx.srs_free(h)


@pytest.mark.parametrize("logn", [10, 14, 16, 20])
def test_vs_oracle_bucket_msm(ctx, logn):
    """BASELINE configs[1] and [2]: 2^16- and 2^20-point MSM bit-exact vs the restated src/msm.rs CPU path (the 2^20
    case costs the oracle about a core-minute, spread over the box's threads)"""
    n = 1 << logn
    a, d = 0x1234567 + logn, 0xABCDEF01
    aff = O.points_progression(n, a, d)
    h = ctx.srs_load(bytes(O.points_to_bytes96(aff)))
    sc = O.splitmix_scalars(n, 0x5EED0000 + logn)
    got = ctx.msm(h, sc)
    want = O.g1_bytes96(O.bucket_msm(O.affine_to_proj(aff) if n <= 4096 else _bulk_proj(aff), sc, threads=NTHREADS))
    assert got == want
    assert got == M.enc96(M.ec_mul(oracle_dot(sc, a, d)))
    if logn >= 16:                                                   # and through the fixed-base tables (the bench path)
        ctx.srs_precompute(h)
        assert ctx.msm(h, sc) == want and ctx.msm_stats()["tables"]
    ctx.srs_free(h)


def _bulk_proj(aff):
    out = np.zeros((len(aff), 18), dtype=np.uint64)
    out[:, :12] = aff[:, :12]
    one = O.fp_one()
    out[:, 12:] = one
    return out


@experiment
@pytest.mark.parametrize("c,chunk", [(4, 4), (7, 8), (11, 16), (13, 32), (16, 64), (16, 4), (5, 64)])
def test_window_and_chunk_independence(ctx, c, chunk, monkeypatch):
    """the group element does not depend on the window width or on how the sorted list is cut"""
    monkeypatch.setenv("BP_MSM_C", str(c))
    monkeypatch.setenv("BP_MSM_CHUNK", str(chunk))
    n, a, d = 5000, 777, 999331
    h = ctx.srs_generate_progression(n, a, d)
    sc = O.splitmix_scalars(n, 0xC0FFEE + c)
    sc[10] = 0
    sc[11] = bp.scalar_from_int(1)
    sc[12] = bp.scalar_from_int(Q - 1)
    assert ctx.msm(h, sc) == M.enc96(M.ec_mul(oracle_dot(sc, a, d)))
    assert ctx.msm_stats()["window_bits"] == c
    # skewed: one bucket per window holds everything (a long chain of partials through the fix-up kernel)
    same = np.repeat(bp.scalar_from_int(0x0123456789ABCDEF0123456789ABCDEF)[None, :], n, axis=0)
    assert ctx.msm(h, same) == M.enc96(M.ec_mul(oracle_dot(same, a, d)))
    ctx.srs_free(h)


def test_generated_progression_matches_oracle_points(ctx):
    a, d = 31337, 424242
    h = ctx.srs_generate_progression(50, a, d)
    assert ctx.srs_export(h) == progression_bytes(50, a, d)
    assert ctx.srs_export(h, 7, 3) == progression_bytes(50, a, d)[96 * 7: 96 * 10]
    ctx.srs_free(h)


def test_sharded_partials_combine(ctx):
    """multi-GPU path on one GPU: point-range shards -> 144-byte partials -> host combine"""
    n, a, d, shards = 4096, 5, 11, 4
    h = ctx.srs_generate_progression(n, a, d)
    sc = O.splitmix_scalars(n, 0xD157)
    per = n // shards
    parts = b"".join(ctx.msm_partial(h, sc[r * per:(r + 1) * per], first=r * per) for r in range(shards))
    assert bp.sum_partials(parts) == ctx.msm(h, sc) == M.enc96(M.ec_mul(oracle_dot(sc, a, d)))
    ctx.srs_free(h)


def test_full_size_2p20_closed_form(ctx):
    """BASELINE configs[2]: 2^20 points; size-independent check through the closed form"""
    import torch
    n, a, d = 1 << 20, 0x1F2E3D4C5B6A7988, 0x1020304050607
    h = ctx.srs_generate_progression(n, a, d)
    sc = O.splitmix_scalars(n, 0x5EED0014)
    want = M.enc96(M.ec_mul(oracle_dot(sc, a, d)))
    assert ctx.msm(h, sc) == want
    # HBM-resident scalars (bench path) give the same partial
    t = torch.from_numpy(sc.view(np.int64)).cuda()
    torch.cuda.synchronize()
    assert bp.sum_partials(ctx.msm_partial(h, None, device_ptr=t.data_ptr(), n=n)) == want
    # linearity: MSM(s + s') = MSM(s) + MSM(s')
    sc2 = O.splitmix_scalars(n, 0xABCD)
    both = np.zeros_like(sc)
    O.lib.poly_add(both.ctypes.data, sc.ctypes.data, n, sc2.ctypes.data, n, 1)
    p1, p2 = ctx.msm_partial(h, sc), ctx.msm_partial(h, sc2)
    assert bp.sum_partials(p1 + p2) == ctx.msm(h, both)
    ctx.srs_free(h)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 31, 32, 33, 100, 257, 1000, 4097, 30000])
def test_odd_sizes_vs_oracle(ctx, n):
    """the toy proof commits to 9..14 coefficients; nothing here assumes powers of two"""
    a, d = 99991 + n, 31337
    h = ctx.srs_generate_progression(n + 3, a, d)                 # SRS longer than the scalar vector (zip truncation)
    sc = O.splitmix_scalars(n, 0xABC0 + n)
    got = ctx.msm(h, sc)
    assert got == M.enc96(M.ec_mul(oracle_dot(sc, a, d)))
    if n <= 1000:
        aff = O.points_progression(n, a, d)
        assert got == O.g1_bytes96(O.bucket_msm(O.affine_to_proj(aff), sc, threads=NTHREADS))
    ctx.srs_free(h)


@pytest.mark.parametrize("b,c", [(256, 4), (256, 8), (256, 3), (256, 5), (255, 5), (128, 4), (200, 8), (17, 16)])
def test_bucket_msm_any_window_parameters_vs_oracle(ctx, b, c):
    """BucketMSM::bucket_msm(points, scalars, b, c) for window parameters other than the reference's one call site (256, 4): the
    reference then drops the low 256 - c * floor(b / c) bits of every scalar (msm.rs:83, 119-139).  The mirror reproduces the
    oracle's literal restatement byte for byte (VERDICT r02 missing #6); parameters that make the reference panic raise."""
    n = 300
    aff = O.points_progression(n, 0xABCDEF, 0x13579)
    sc = O.splitmix_scalars(n, 0xB0C0 + b + c)
    sc[3] = bp.scalar_from_int(Q - 1)
    want = O.g1_bytes96(O.bucket_msm(O.affine_to_proj(aff), sc, b, c, threads=NTHREADS))
    assert bp.BucketMSM.bucket_msm(bytes(O.points_to_bytes96(aff)), sc, b, c, ctx) == want
    for bad_b, bad_c in ((256, 0), (3, 4), (300, 4), (256, 64)):
        with pytest.raises(bp.BpError):
            bp.BucketMSM.bucket_msm(bytes(O.points_to_bytes96(aff)), sc, bad_b, bad_c, ctx)


def _loaded_hip_runtime():
    """ctypes handle of the HIP runtime this process has ALREADY loaded (the one libbp_msm_ntt.so and torch call into)"""
    import ctypes as C
    with open("/proc/self/maps") as f:
        paths = sorted({line.split()[-1] for line in f if "libamdhip64" in line})
    assert paths, "no HIP runtime mapped"
    return [C.CDLL(p) for p in paths]


def test_stale_hip_error_of_the_calling_thread_is_not_ours(ctx):
    """hipGetLastError() reports the calling thread's LAST error whoever caused it (torch, RCCL, a previous context's
    teardown: gpurun_out/r3_t37.log).  Plant one -- hipSetDevice(9999) -- in front of bp_init and of an MSM: both must succeed."""
    def plant():
        for hip in _loaded_hip_runtime():
            hip.hipSetDevice.restype = int
            assert hip.hipSetDevice(9999) != 0                 # invalid device ordinal: now pending in this thread
    plant()
    fresh = bp.Context(0)                                       # bp_init: stream / events / LDS-limit launches behind a stale error
    try:
        h = fresh.srs_generate_progression(64, 3, 5)
        sc = [7 * i + 1 for i in range(64)]
        plant()
        assert fresh.msm(h, frs(sc)) == closed_form(sc, 3, 5)
        plant()
        v = bp.scalars_from_ints(list(range(1, 257)))
        assert (fresh.ntt(fresh.ntt(v), inverse=True) == v).all()
    finally:
        fresh.close()
    plant()
    h = ctx.srs_generate_progression(32, 2, 1)
    plant()
    assert ctx.msm(h, frs(list(range(32)))) == closed_form(list(range(32)), 2, 1)
    ctx.srs_free(h)

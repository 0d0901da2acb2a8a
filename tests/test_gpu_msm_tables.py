"""-m gpu: G1 MSM through the SRS's fixed-base window tables (bp_srs_precompute) -- the same group element as the
table-free path, the oracle (src/msm.rs restatement) and the closed forms, bit-exact on the 96-byte encoding."""
import os
import random

import numpy as np
import pytest

import baby_plonk_rust_amd as bp
from baby_plonk_rust_amd._lib import MSM_BLOB_BYTES
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


@pytest.mark.parametrize("c", [0, 4, 5, 9, 12])
def test_golden_fixture_with_tables(ctx, c):
    """the reference's 1000-point fixture (i*G, point 0 = identity): every table width gives the oracle's bytes"""
    h = ctx.srs_load(UNCOMP)
    rnd = random.Random(210 + c)
    sc = [rnd.randrange(Q) for _ in range(1000)]
    plain = ctx.msm(h, frs(sc))
    assert not ctx.msm_stats()["tables"]
    info = ctx.srs_precompute(h, c)
    assert info["windows"] * 1000 * 128 == info["bytes"] and (c == 0 or info["window_bits"] == c)
    got = ctx.msm(h, frs(sc))
    st = ctx.msm_stats()
    assert st["tables"] and st["window_bits"] == info["window_bits"]
    assert got == plain == M.enc96(M.ec_mul(sum(i * s for i, s in enumerate(sc))))
    assert got == O.g1_bytes96(O.bucket_msm(O.proj_from_bytes96(UNCOMP), O.fr_array_from_ints(sc), threads=NTHREADS))
    assert ctx.srs_export(h) == UNCOMP                       # the SRS itself is untouched
    # every table row on its own: a one-hot scalar 2^(c w) at point j reads T[w][j] only
    cb = info["window_bits"]
    for w in range(info["windows"]):
        if cb * w >= 255:
            break
        one_hot = [0] * 1000
        one_hot[7 + w] = 1 << (cb * w)
        assert ctx.msm(h, frs(one_hot)) == M.enc96(M.ec_mul((7 + w) << (cb * w)))
    ctx.srs_precompute(h, bp.SRS_TABLES_OFF)
    assert ctx.srs_table_info(h) == {"window_bits": 0, "windows": 0, "bytes": 0}
    assert ctx.msm(h, frs(sc)) == plain and not ctx.msm_stats()["tables"]
    ctx.srs_free(h)


@pytest.mark.parametrize("c", [6, 11])
def test_edges_with_tables(ctx, c):
    rnd = random.Random(22 + c)
    a, d, n = rnd.randrange(Q), rnd.randrange(Q), 300
    h = ctx.srs_load(progression_bytes(n, a, d))
    ctx.srs_precompute(h, c)
    sc = [rnd.randrange(Q) for _ in range(n)]
    for case in (sc, sc[:17], sc + sc, [1] * n, [Q - 1] * n, [7], [rnd.randrange(16) for _ in range(n)],
                 [0x0123456789ABCDEF0123456789ABCDEF] * n):
        assert ctx.msm(h, frs(case)) == closed_form(case[:n], a, d)
        assert ctx.msm_stats()["tables"] == (8 * min(len(case), n) >= (1 << c))
    assert ctx.msm(h, frs([])) == M.enc96(None)
    assert ctx.msm(h, frs([0] * n)) == M.enc96(None)
    one_hot = [0] * n
    one_hot[123] = 2**254 + 12345
    assert ctx.msm(h, frs(one_hot)) == closed_form(one_hot, a, d)
    # canonical little-endian scalars, and the rejection of values >= q, go through the same digit kernel
    le = np.frombuffer(b"".join(s.to_bytes(32, "little") for s in sc), dtype=np.uint8).reshape(-1, 32)
    assert ctx.msm(h, le, fmt=bp.FR_BYTES_LE) == closed_form(sc, a, d)
    bad = le.copy()
    bad[5] = np.frombuffer(Q.to_bytes(32, "little"), dtype=np.uint8)
    with pytest.raises(bp.BpError) as e:
        ctx.msm(h, bad, fmt=bp.FR_BYTES_LE)
    assert e.value.code == -4
    # point-range shards read the table rows at an offset
    parts = b"".join(ctx.msm_partial(h, frs(sc[r * 75:(r + 1) * 75]), first=r * 75) for r in range(4))
    assert bp.sum_partials(parts) == closed_form(sc, a, d)
    ctx.srs_free(h)


def test_degenerate_srs_with_tables(ctx):
    rnd = random.Random(5)
    sc = [rnd.randrange(Q) for _ in range(64)]
    s1 = bp.Setup.generate_srs(64, 1, ctx)                   # tau = 1 (prover.rs:684): all points equal, adds are doublings
    assert ctx.srs_table_info(s1.handle)["window_bits"] > 0  # a Setup builds its tables
    assert ctx.msm(s1.handle, frs(sc)) == M.enc96(M.ec_mul(sum(sc))) and ctx.msm_stats()["tables"]
    s0 = bp.Setup.generate_srs(5, 0, ctx)                    # tau = 0: G then identities; identity rows stay (0, 0)
    assert ctx.msm(s0.handle, frs(sc[:5])) == M.enc96(M.ec_mul(sc[0]))
    hh = ctx.srs_load(M.enc96(M.ec_mul(5)) + M.enc96(M.ec_mul(Q - 5)))
    ctx.srs_precompute(hh, 4)
    assert ctx.msm(hh, frs([99, 99])) == M.enc96(None)       # P and -P cancel in every table row
    assert ctx.msm(hh, frs([Q - 1, 3])) == M.enc96(M.ec_mul((5 * (Q - 1) - 15) % Q))
    with pytest.raises(bp.BpError):
        ctx.srs_precompute(hh, 25)                            # widths are 4..24
    with pytest.raises(bp.BpError):
        ctx.srs_precompute(hh, 3)
    with pytest.raises(bp.BpError):
        ctx.srs_precompute(987654321, 0)
    ctx.srs_free(hh)


def test_setup_commit_uses_tables(ctx):
    """Setup::commit (setup.rs:32-37) on the toy-proof SRS (tests/verify_proof_test.rs:16), tables on and off"""
    on, off = bp.Setup.generate_srs(14, 101, ctx), bp.Setup.generate_srs(14, 101, ctx, tables=False)
    rnd = random.Random(14)
    for k in (1, 9, 14):
        co = [rnd.randrange(Q) for _ in range(k)]
        p = bp.Polynomial(frs(co), bp.BASIS_MONOMIAL, ctx)
        want = M.enc96(M.ec_mul(sum(v * pow(101, i, Q) for i, v in enumerate(co)) % Q))
        assert on.commit(p) == want and ctx.msm_stats()["tables"] == (8 * k >= 1 << ctx.srs_table_info(on.handle)["window_bits"])
        assert off.commit(p) == want and not ctx.msm_stats()["tables"]


@pytest.mark.parametrize("logn", [12, 16])
def test_vs_oracle_bucket_msm_with_tables(ctx, logn):
    """BASELINE configs[1] with tables: 2^16-point MSM bit-exact vs the restated src/msm.rs CPU path"""
    n = 1 << logn
    a, d = 0x7654321 + logn, 0x10FEDCBA
    aff = O.points_progression(n, a, d)
    h = ctx.srs_load(bytes(O.points_to_bytes96(aff)))
    info = ctx.srs_precompute(h)
    sc = O.splitmix_scalars(n, 0x7AB1E000 + logn)
    got = ctx.msm(h, sc)
    assert ctx.msm_stats()["tables"] and info["window_bits"] == (min(16, logn + 2) if logn >= 14 else max(4, logn - 4))
    proj = np.zeros((n, 18), dtype=np.uint64)
    proj[:, :12] = aff[:, :12]
    proj[:, 12:] = O.fp_one()
    assert got == O.g1_bytes96(O.bucket_msm(proj, sc, threads=NTHREADS))
    assert got == M.enc96(M.ec_mul(oracle_dot(sc, a, d)))
    ctx.srs_free(h)


def test_full_size_2p20_with_tables(ctx):
    """BASELINE configs[2] through the tables: closed form, equality with the table-free path, linearity"""
    import torch
    n, a, d = 1 << 20, 0x1F2E3D4C5B6A7988, 0x1020304050607
    h = ctx.srs_generate_progression(n, a, d)
    sc = O.splitmix_scalars(n, 0x5EED0014)
    plain = ctx.msm(h, sc)
    want = M.enc96(M.ec_mul(oracle_dot(sc, a, d)))
    info = ctx.srs_precompute(h, 16)                                 # the widest window whose buckets fit one LDS histogram (round 2's choice here)
    assert info == {"window_bits": 16, "windows": 16, "bytes": 16 * n * 128}
    assert ctx.msm(h, sc) == want == plain and ctx.msm_stats()["tables"] and ctx.msm_stats()["window_bits"] == 16
    info = ctx.srs_precompute(h)                                     # auto: 13 windows of 20 bits from 2^20 points (profiles/r03_window_width_ab.txt)
    assert info == {"window_bits": 20, "windows": 13, "bytes": 13 * n * 128}
    assert ctx.msm(h, sc) == want == plain and ctx.msm_stats()["tables"] and ctx.msm_stats()["window_bits"] == 20
    t = torch.from_numpy(sc.view(np.int64)).cuda()
    torch.cuda.synchronize()
    assert bp.sum_partials(ctx.msm_partial(h, None, device_ptr=t.data_ptr(), n=n)) == want
    sc2 = O.splitmix_scalars(n, 0xABCD)
    both = np.zeros_like(sc)
    O.lib.poly_add(both.ctypes.data, sc.ctypes.data, n, sc2.ctypes.data, n, 1)
    assert bp.sum_partials(ctx.msm_partial(h, sc) + ctx.msm_partial(h, sc2)) == ctx.msm(h, both)
    # a shorter polynomial against the same tables (rows stay srs_len apart), and a shard in the middle
    assert ctx.msm(h, sc[:300000]) == M.enc96(M.ec_mul(oracle_dot(sc[:300000], a, d)))
    lo = 123457
    assert bp.sum_partials(ctx.msm_partial(h, sc[:50000], first=lo)) == M.enc96(M.ec_mul(oracle_dot(sc[:50000], a + lo * d, d)))
    ctx.srs_free(h)


def test_north_star_shard_2p21_auto_width(ctx):
    """what ONE of 8 GPUs holds in the north-star configuration (2^24 points over 8 GPUs): a 2^21-point SRS shard with its own tables at the
    automatic width (20 bits, 13 windows; entries need 26 bits, so the partition sort runs on two-word records there) -- closed form on the
    shard's own point range, with the scalars generated in HBM"""
    import torch
    n, a, d, seed, first = 1 << 21, 0x0F1E2D3C4B5A6978, 0x1122334455, 0x5EED0015, 5 << 21         # shard 5 of 8
    h = ctx.srs_generate_progression(n, a + first * d, d)
    info = ctx.srs_precompute(h)
    assert info == {"window_bits": 20, "windows": 13, "bytes": 13 * n * 128}
    t = torch.empty(n * 4, dtype=torch.int64, device="cuda")
    ctx.synthetic_scalars_device(t.data_ptr(), n, seed)
    sc = O.splitmix_scalars(n, seed)
    want = M.enc96(M.ec_mul(O.dot_progression(sc, a + first * d, d)))
    assert bp.sum_partials(ctx.msm_partial(h, None, device_ptr=t.data_ptr(), n=n)) == want
    st = ctx.msm_stats()
    assert st["tables"] and st["window_bits"] == 20 and 0 < st["mixed_adds"] <= 13 * n
    assert ctx.msm(h, sc) == want                                   # the same from host scalars
    ctx.srs_free(h)


def test_full_size_2p24_closed_form_both_paths(ctx):
    """BASELINE configs[3] size on one GPU: 2^24 points generated in HBM, scalars generated in HBM by the same SplitMix64 stream
    the oracle reproduces on the CPU; the result must equal the closed form (sum s_i (a + i d)) G with and without tables"""
    import torch
    n, a, d, seed = 1 << 24, 0x0F1E2D3C4B5A6978, 0x1122334455, 0x5EED0018
    h = ctx.srs_generate_progression(n, a, d)
    t = torch.empty(n * 4, dtype=torch.int64, device="cuda")
    ctx.synthetic_scalars_device(t.data_ptr(), n, seed)
    sc = O.splitmix_scalars(n, seed)                                # the same stream on the host
    assert (t[:4 * 1000].cpu().numpy().view(np.uint64).reshape(-1, 4) == sc[:1000]).all()
    want = M.enc96(M.ec_mul(O.dot_progression(sc, a, d)))
    plain = bp.sum_partials(ctx.msm_partial(h, None, device_ptr=t.data_ptr(), n=n))
    assert plain == want and not ctx.msm_stats()["tables"]
    info = ctx.srs_precompute(h)                                     # auto width at 2^24 points: 22 bits = 12 windows, 2^21 buckets (two-level radix sort)
    assert info["window_bits"] == 22 and info["windows"] == 12 and info["bytes"] == 12 * n * 128
    assert bp.sum_partials(ctx.msm_partial(h, None, device_ptr=t.data_ptr(), n=n)) == want and ctx.msm_stats()["tables"]
    assert ctx.msm_stats()["window_bits"] == 22
    info = ctx.srs_precompute(h, 20)                                 # the width of 2^20 .. 2^23 points
    assert info["window_bits"] == 20 and info["windows"] == 13 and info["bytes"] == 13 * n * 128
    assert bp.sum_partials(ctx.msm_partial(h, None, device_ptr=t.data_ptr(), n=n)) == want and ctx.msm_stats()["window_bits"] == 20
    info = ctx.srs_precompute(h, 16)                                 # and the one-histogram-per-window width at the same size
    assert info["window_bits"] == 16 and info["bytes"] == 16 * n * 128
    assert bp.sum_partials(ctx.msm_partial(h, None, device_ptr=t.data_ptr(), n=n)) == want and ctx.msm_stats()["tables"]
    # point-range shards as 8 GPUs would hold them (2^21 points each), combined on the host
    per = n // 8
    parts = b"".join(ctx.msm_partial(h, None, first=r * per, device_ptr=t.data_ptr() + 32 * r * per, n=per) for r in range(8))
    assert bp.sum_partials(parts) == want
    ctx.srs_free(h)


def test_full_size_2p24_vs_literal_bucket_msm(ctx):
    """BASELINE configs[3] size against the LITERAL restatement of src/msm.rs:76-118: one 2^24-point MSM through the
    22-bit tables, byte for byte equal to the oracle's bucket_msm(256, 4) over the same 2^24 points and scalars (64 windows spread over the
    box's threads: about a minute), with the closed form beside it.  The oracle gets the points the way the Rust caller would hand them
    over -- 96-byte encodings read back from the library -- and the closed form pins that they are the progression asked for."""
    import torch
    n, a, d, seed = 1 << 24, 0x0123456789ABCDEF01, 0x0F0E0D0C0B0A, 0x5EED2424
    h = ctx.srs_generate_progression(n, a, d)
    info = ctx.srs_precompute(h)
    assert info["window_bits"] == 22
    t = torch.empty(n * 4, dtype=torch.int64, device="cuda")
    ctx.synthetic_scalars_device(t.data_ptr(), n, seed)
    got = bp.sum_partials(ctx.msm_partial(h, None, device_ptr=t.data_ptr(), n=n))
    assert ctx.msm_stats()["tables"] and ctx.msm_stats()["window_bits"] == 22
    del t
    sc = O.splitmix_scalars(n, seed)
    assert got == M.enc96(M.ec_mul(O.dot_progression(sc, a, d)))
    proj = np.empty((n, 18), dtype=np.uint64)
    step = 1 << 21
    for lo in range(0, n, step):                                     # 96-byte encodings -> the oracle's projective limbs, 2^21 at a time
        proj[lo:lo + step] = O.proj_from_bytes96(ctx.srs_export(h, lo, step))
    ctx.srs_free(h)
    # spot-check the exported points against the oracle's own progression (first points and a far one)
    head = O.proj_from_bytes96(bytes(O.points_to_bytes96(O.points_progression(64, a, d))))
    assert (proj[:64] == head).all()
    far = n - 12345
    assert bytes(O.g1_bytes96(proj[far])) == M.enc96(M.ec_mul((a + far * d) % Q))
    want = O.g1_bytes96(O.bucket_msm(proj, sc, threads=NTHREADS))
    assert got == want


@pytest.mark.parametrize("c,log_n", [(17, 14), (18, 15), (19, 16), (20, 17), (21, 18), (22, 19), (24, 21)])
def test_windows_wider_than_16_bits(ctx, c, log_n):
    """fixed-base tables with c > 16: one window's 2^(c-1) buckets exceed the LDS histogram, so the entries go through the
    partitioned (radix) bucket sort (msm_digit_records, msm_radix_*: runs of ~12 Ki entries sorted by one workgroup each, long
    runs -- the narrow top window of c = 18, 19, 21 -- slice-parallel) and the bit-plane tree takes extra merge steps.
    Same bytes as the closed form and as the c <= 16 paths; every table row probed on its own; skewed and edge scalars."""
    n, a, d = (1 << log_n) + 13, Q - 98765, 0x0F1E2D3C4B5A6978
    h = ctx.srs_generate_progression(n, a, d)
    sc = O.splitmix_scalars(n, 0x5EED0000 + c)
    want = M.enc96(M.ec_mul(oracle_dot(sc, a, d)))
    assert ctx.msm(h, sc) == want and not ctx.msm_stats()["tables"]
    info = ctx.srs_precompute(h, c)
    assert info["window_bits"] == c and info["windows"] == {17: 16, 18: 15, 19: 14, 20: 13, 21: 13, 22: 12, 24: 11}[c]
    assert ctx.msm(h, sc) == want
    st = ctx.msm_stats()
    assert st["tables"] and st["window_bits"] == c and 0 < st["mixed_adds"] <= info["windows"] * n
    # a prefix and an offset range (zip truncation / shards) through the same tables
    assert bp.sum_partials(ctx.msm_partial(h, sc[: n // 2 + 3])) == M.enc96(M.ec_mul(oracle_dot(sc[: n // 2 + 3], a, d)))
    assert bp.sum_partials(ctx.msm_partial(h, sc[: n - 101], first=101)) == M.enc96(M.ec_mul(oracle_dot(sc[: n - 101], a + 101 * d, d)))
    # every row of the tables alone
    for w in range(info["windows"]):
        if c * w >= 255:
            break
        one_hot = np.zeros((n, 4), dtype=np.uint64)
        j = (977 * (w + 1)) % n
        one_hot[j] = bp.scalar_from_int(1 << (c * w))
        assert ctx.msm(h, one_hot) == M.enc96(M.ec_mul(((a + j * d) << (c * w)) % Q)), w
    # edge scalars: q - 1 (top digit, sign carries through every window), 1, 0, all equal (one bucket per window, the long fix-up)
    for val in (Q - 1, 1, 0, 0x123456789ABCDEF0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF % Q):
        same = np.tile(bp.scalar_from_int(val), (n, 1))
        k = val * ((n * a + d * (n * (n - 1) // 2)) % Q) % Q
        assert ctx.msm(h, same) == M.enc96(M.ec_mul(k)), hex(val)
    small = bp.scalars_from_ints([(i * 7919) % 65536 for i in range(n)]) if n <= (1 << 16) + 13 else None
    if small is not None:                                       # 16-bit witness-like values: only the low windows are populated
        assert ctx.msm(h, small) == M.enc96(M.ec_mul(oracle_dot(small, a, d)))
    ctx.srs_free(h)


@pytest.mark.parametrize("c", list(range(4, 25)))
def test_table_width_sweep_on_the_shipped_library(ctx, c):
    """every table width the ABI accepts (bp_srs_precompute(h, c), c = 4 .. 24) x eight sizes from 2^9 to 2^18 (odd exponents with an odd tail) x scalar shapes
    {uniform, all equal, 0 / 1, one-hot, q - 1} against the closed form, on whichever library is loaded -- the SHIPPED one in the default run
    (the window / sort / fix-up / tree variants used to be reachable through experiment-build knobs only).  The widths
    select the code paths by themselves: c <= 16 packed one-word records and up to 15 tree levels, c >= 17 the partitioned sort with final
    runs, long runs for the narrow top windows (c = 18, 19, 21, 23), cooperative and wide tree levels, per-edge and per-bucket fix-up; sizes
    with 8 n < 2^c keep to the table-free path (per-window bucket sets, running-sum reduction) and must say so."""
    rnd = random.Random(0xC0DE00 + c)
    for log_n in (9, 10, 12, 13, 15, 16, 17, 18):
        n = (1 << log_n) + (rnd.randrange(1, 64) if log_n % 2 else 0)
        a, d = rnd.randrange(1, Q), rnd.randrange(1, Q)
        h = ctx.srs_generate_progression(n, a, d)
        info = ctx.srs_precompute(h, c)
        assert info["window_bits"] == c and info["bytes"] == info["windows"] * n * 128
        expect_tables = 8 * n >= (1 << c)
        sum_pts = (n * a + d * (n * (n - 1) // 2)) % Q                 # sum_i (a + i d)
        uni = O.splitmix_scalars(n, 0x5EEDC000 + 64 * c + log_n)
        assert ctx.msm(h, uni) == M.enc96(M.ec_mul(oracle_dot(uni, a, d))), (c, log_n, "uniform")
        st = ctx.msm_stats()
        assert bool(st["tables"]) == expect_tables and (not expect_tables or st["window_bits"] == c), (c, log_n, st)
        val = rnd.randrange(1, Q)
        assert ctx.msm(h, np.tile(bp.scalar_from_int(val), (n, 1))) == M.enc96(M.ec_mul(val * sum_pts % Q)), (c, log_n, "all equal")
        bits = np.frombuffer(rnd.randbytes(n), dtype=np.uint8) & 1
        zo = np.zeros((n, 4), dtype=np.uint64)
        zo[bits == 1] = bp.scalar_from_int(1)
        idx = np.nonzero(bits)[0]
        k01 = (int(len(idx)) * a + d * int(idx.sum(dtype=np.int64))) % Q
        assert ctx.msm(h, zo) == M.enc96(M.ec_mul(k01)), (c, log_n, "0/1")
        j, v = rnd.randrange(n), rnd.randrange(1, Q)
        hot = np.zeros((n, 4), dtype=np.uint64)
        hot[j] = bp.scalar_from_int(v)
        assert ctx.msm(h, hot) == M.enc96(M.ec_mul(v * (a + j * d) % Q)), (c, log_n, "one-hot")
        assert ctx.msm(h, np.tile(bp.scalar_from_int(Q - 1), (n, 1))) == M.enc96(M.ec_mul((Q - 1) * sum_pts % Q)), (c, log_n, "q - 1")
        ctx.srs_free(h)


# (width, log2 points, windows, radix): the widths that take radix-R digits (csrc/msm_digits.hpp: from 21 bits), at sizes where an MSM really uses the
# tables (8 n >= 2^c).  The values are the ones tests/test_radix_digits.py checks on the CPU.
RADIX_CASES = [(21, 18, 13, 0xD0000), (22, 19, 12, 0x288000), (23, 20, 12, 0x288000), (24, 21, 11, 0x9C0000)]


@pytest.mark.parametrize("c,log_n,W,R", RADIX_CASES)
def test_radix_r_digits_through_the_tables(ctx, c, log_n, W, R):
    """Round 6: table widths of 21 bits and more cut the scalars in a radix R that is not a power of two (T[w][i] = R^w P_i).  Scalars that sit ON the
    digit boundaries of that radix -- multiples of R^j and their neighbours, digits of exactly +-R/2, the largest scalar -- spread over the vector
    beside uniform ones, against the closed form sum_i s_i (a + i d) G (src/setup.rs:32-37 -> src/msm.rs:76-118: any bucket method yields the same element)."""
    rnd = random.Random(0xAD1C5 + c)
    n = (1 << log_n) + 37
    a, d = rnd.randrange(1, Q), rnd.randrange(1, Q)
    h = ctx.srs_generate_progression(n, a, d)
    info = ctx.srs_precompute(h, c)
    assert info["window_bits"] == c and info["windows"] == W
    uni = O.splitmix_scalars(n, 0xAD1C5000 + c)
    want_uni = M.enc96(M.ec_mul(oracle_dot(uni, a, d)))
    assert ctx.msm(h, uni) == want_uni and ctx.msm_stats()["tables"] and ctx.msm_stats()["window_bits"] == c
    edge = [0, 1, R // 2 - 1, R // 2, R // 2 + 1, R - 1, R, R + 1, Q - 1, Q - 2, Q // 2, 2 ** 254]
    for j in range(1, W):
        for t in (1, R // 2, R // 2 + 1, R - 1):
            for e in (-1, 0, 1):
                edge.append((t * R ** j + e) % Q)
        edge.append(sum((R // 2) * R ** i for i in range(j + 1)) % Q)
        edge.append(sum((R // 2 - 1) * R ** i for i in range(j + 1)) % Q)
    mixed = uni.copy()
    pos = rnd.sample(range(n), len(edge))
    mixed[pos] = frs(edge)
    assert ctx.msm(h, mixed) == M.enc96(M.ec_mul(oracle_dot(mixed, a, d))), (c, "edge scalars among uniform ones")
    only = np.zeros((n, 4), dtype=np.uint64)                          # the edge scalars alone: everything else zero (no entries at all)
    only[pos] = frs(edge)
    assert ctx.msm(h, only) == M.enc96(M.ec_mul(oracle_dot(only, a, d))), (c, "edge scalars alone")
    top = np.tile(bp.scalar_from_int(Q - 1), (n, 1))                  # every top digit at its maximum
    sum_pts = (n * a + d * (n * (n - 1) // 2)) % Q
    assert ctx.msm(h, top) == M.enc96(M.ec_mul((Q - 1) * sum_pts % Q)), (c, "q - 1 everywhere")
    le = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in (5, Q - 1, 7)), dtype=np.uint8).reshape(-1, 32)
    assert ctx.msm(h, le, fmt=bp.FR_BYTES_LE) == M.enc96(M.ec_mul((5 * a + (Q - 1) * (a + d) + 7 * (a + 2 * d)) % Q))       # (three scalars: the table-free path)
    bad = np.tile(np.frombuffer((2 ** 256 - 1).to_bytes(32, "little"), dtype=np.uint8), (n, 1))           # not scalars: rejected, and no bucket index leaves the live range
    with pytest.raises(bp.BpError) as ei:
        ctx.msm(h, bad, fmt=bp.FR_BYTES_LE)
    assert ei.value.code == -4
    assert ctx.msm(h, uni) == want_uni                                # the context is intact afterwards
    ctx.srs_free(h)


@experiment            # every-position tables exist in the experiment build only (measured slower twice)
@pytest.mark.parametrize("w,log_n", [(6, 9), (8, 12), (11, 13), (14, 15), (16, 16), (18, 17), (19, 17), (22, 17)])
def test_every_position_tables_naf_digits(ctx, w, log_n):
    """bp_srs_precompute(handle, 256 + w): tables of EVERY bit position (256 rows, T[p][i] = 2^p P_i) carry the scalars' width-w
    non-adjacent form -- odd signed digits at arbitrary positions, 2^(w-2) buckets holding digit 2b + 1, one bucket set.
    Same bytes as the closed form and the other paths; every row probed alone; edge and skewed scalars; prefixes and offsets;
    canonical-bytes scalars and their rejection; records of two shards combined."""
    n, a, d = (1 << log_n) + 7, Q - 424242, 0x1A2B3C4D5E6F7081
    h = ctx.srs_generate_progression(n, a, d)
    sc = O.splitmix_scalars(n, 0x5EED1000 + w)
    want = M.enc96(M.ec_mul(oracle_dot(sc, a, d)))
    assert ctx.msm(h, sc) == want and not ctx.msm_stats()["tables"]
    info = ctx.srs_precompute(h, 256 + w)
    assert info == {"window_bits": 256 + w, "windows": 256, "bytes": 256 * n * 128}
    assert ctx.msm(h, sc) == want
    st = ctx.msm_stats()
    assert st["tables"] and st["window_bits"] == 256 + w
    assert 0 < st["mixed_adds"] <= (255 // w + 1) * n and st["mixed_adds"] < n * (256 / (w + 1) + 1.5)     # the NAF's density
    assert bp.sum_partials(ctx.msm_partial(h, sc[: n // 2 + 3])) == M.enc96(M.ec_mul(oracle_dot(sc[: n // 2 + 3], a, d)))
    assert bp.sum_partials(ctx.msm_partial(h, sc[: n - 101], first=101)) == M.enc96(M.ec_mul(oracle_dot(sc[: n - 101], a + 101 * d, d)))
    for p in list(range(0, 255, 17)) + [253, 254]:               # rows alone: scalar 2^p (digit 1 at position p) and 3 * 2^p
        one_hot = np.zeros((n, 4), dtype=np.uint64)
        j = (977 * (p + 1)) % n
        one_hot[j] = bp.scalar_from_int((1 << p) % Q)
        assert ctx.msm(h, one_hot) == M.enc96(M.ec_mul(((a + j * d) << p) % Q)), p
    for val in (Q - 1, Q - 2, 1, 2, 3, 0, (1 << 254) + 1, (1 << 254) - 1,
                0x123456789ABCDEF0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF % Q, int("55" * 31, 16), int("AA" * 31, 16) % Q):
        same = np.tile(bp.scalar_from_int(val % Q), (n, 1))
        k = (val % Q) * ((n * a + d * (n * (n - 1) // 2)) % Q) % Q
        assert ctx.msm(h, same) == M.enc96(M.ec_mul(k)), hex(val)
    small = bp.scalars_from_ints([(i * 7919) % 65536 for i in range(n)])
    assert ctx.msm(h, small) == M.enc96(M.ec_mul(oracle_dot(small, a, d)))
    ints = bp.scalars_to_ints(sc[:200])
    le = np.frombuffer(b"".join(s.to_bytes(32, "little") for s in ints), dtype=np.uint8).reshape(-1, 32)
    assert ctx.msm(h, le, fmt=bp.FR_BYTES_LE) == M.enc96(M.ec_mul(oracle_dot(sc[:200], a, d)))
    bad = le.copy()
    bad[5] = np.frombuffer(Q.to_bytes(32, "little"), dtype=np.uint8)
    with pytest.raises(bp.BpError) as e:
        ctx.msm(h, bad, fmt=bp.FR_BYTES_LE)
    assert e.value.code == -4
    # two point-range shards as device records, combined on the host (bp_msm_blobs_combine: the records say "odd digits")
    import torch
    t = torch.from_numpy(sc.view(np.int64)).cuda()
    rec = torch.zeros(2 * MSM_BLOB_BYTES, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    half = n // 2
    ctx.msm_blob_device(h, rec.data_ptr(), None, first=0, device_ptr=t.data_ptr(), n=half)
    ctx.msm_blob_device(h, rec.data_ptr() + MSM_BLOB_BYTES, None, first=half, device_ptr=t.data_ptr() + 32 * half, n=n - half)
    assert bp.combine_blobs(rec.cpu().numpy().tobytes()) == want
    with pytest.raises(bp.BpError):
        ctx.srs_precompute(h, 256 + 5)
    with pytest.raises(bp.BpError):
        ctx.srs_precompute(h, 256 + 23)
    ctx.srs_free(h)
    many = bp.Context([0, 0, 0])                                  # the same through a three-shard context
    hm = many.srs_generate_progression(n, a, d)
    many.srs_precompute(hm, 256 + w)
    assert many.msm(hm, sc) == want
    assert many.msm_stats()["tables"] == (8 * (n // 3) >= (1 << (w - 2)))      # short shards keep to the plain points
    many.close()


def test_table_memory_budget(ctx):
    """bp_srs_precompute decides on the device's free memory before it builds: when the tables do not fit, the automatic
    width falls back to no tables (same bytes out), an explicit width is refused with BP_ERR_TOO_LARGE and the sizes -- never an
    out-of-memory failure halfway through -- and a refusal leaves the tables the SRS had.  The free-memory READING is injected through the
    library's internal test hook (bpx_set_free_bytes_probe): rounds 3-5 filled the device with a ~280-GB tensor until
    400 MiB were left, which made the outcome depend on when other processes' frees reached the driver."""
    import ctypes as C
    hook = ctx._lib.bpx_set_free_bytes_probe
    hook.restype, hook.argtypes = C.c_int, [C.c_void_p, C.c_uint64]
    n = 1 << 18
    h = ctx.srs_generate_progression(n, 9, 4)
    rnd = random.Random(77)
    sc = [rnd.randrange(Q) for _ in range(n)]
    want = closed_form(sc, 9, 4)
    assert ctx.msm(h, frs(sc)) == want
    try:
        assert hook(ctx._h, 400 << 20) == 0                       # "400 MiB free": 2^18 points x 16 rows x 128 B = 512 MiB do not fit
        info = ctx.srs_precompute(h, 0)
        assert info["bytes"] == 0 and info["windows"] == 0, info
        assert ctx.msm(h, frs(sc)) == want
        with pytest.raises(bp.BpError) as ei:
            ctx.srs_precompute(h, 16)
        assert ei.value.code == -10 and "GiB" in str(ei.value), ei.value
        assert hook(ctx._h, 0) == 0                               # the real reading again: the tables fit
        info = ctx.srs_precompute(h, 0)
        assert info["bytes"] > 0 and info["window_bits"] == 16
        assert ctx.msm(h, frs(sc)) == want
        # a refused explicit width leaves the SRS as it was -- its 16-bit tables included (they used to be released before the check)
        assert hook(ctx._h, 400 << 20) == 0
        with pytest.raises(bp.BpError) as ei:
            ctx.srs_precompute(h, 8)                              # 32 rows = 1 GiB: does not fit in "400 MiB" + the 512 MiB the old tables hold
        assert ei.value.code == -10
        assert ctx.srs_table_info(h)["window_bits"] == 16 and ctx.srs_table_info(h)["bytes"] == 16 * n * 128
        assert ctx.msm(h, frs(sc)) == want and ctx.msm_stats()["tables"]
        ctx.srs_precompute(h, 16)                                 # the same width again fits: the bytes of the tables it replaces count as free
        assert ctx.srs_table_info(h)["window_bits"] == 16
    finally:
        hook(ctx._h, 0)
    ctx.srs_free(h)

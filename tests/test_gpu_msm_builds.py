"""-m gpu, experiment build only (BABY_PLONK_LIBRARY=exp): every build of the MSM pipeline stages gives the same bytes.

The bucket sort has three builds (partition sort with packed or two-word records, two-level radix sort, one-histogram counting
sort), the fix-up two (a lane pair per bucket, a lane per chunk edge), the bit-plane tree two kinds of level (one addition per
lane through HBM, cooperative additions inside a workgroup).  The library picks among them by size; the experiment knobs force
each one here, at a size where every branch (long final runs, long buckets, queued merges) is reached by skewed scalars.
Expected values: the closed form sum_i s_i (a + i d) G by independent big-int arithmetic (tests/bigint_model.py); the group
element is what src/msm.rs:76-118 returns for the same inputs whatever the bucket method."""
import numpy as np
import pytest

import baby_plonk_rust_amd as bp
from oracle import oracle as O
from tests import bigint_model as M
from tests.gpu_common import Q, experiment, oracle_dot

pytestmark = [pytest.mark.gpu, experiment]        # every case forces a build through a BP_* knob

BUILDS = [
    {"BP_MSM_SORT": "2"},                                   # partition sort, packed records (the default)
    {"BP_MSM_SORT": "2", "BP_MSM_PACKED": "0"},             # partition sort, two-word records
    {"BP_MSM_SORT": "2", "BP_MSM_PART_SLICE": "64"},        # many small slices: one record per (slice, partition) or none
    {"BP_MSM_SORT": "2", "BP_MSM_RADIX_BITS": "0"},         # a single final run (no partition level), packed
    {"BP_MSM_SORT": "1"},                                   # two-level radix sort
    {"BP_MSM_SORT": "0"},                                   # histogram sort (c <= 16; wider windows fall back to the default)
    {"BP_MSM_FIXUP": "1"},                                  # fix-up per bucket
    {"BP_MSM_FIXUP": "2"},                                  # fix-up per chunk edge
    {"BP_MSM_PLANES_WIDE_MIN": "256"},                      # tree: every level with >= 256 additions through HBM
    {"BP_MSM_PLANES_WIDE_MIN": "1000000000"},               # tree: cooperative steps only
]


@pytest.fixture(scope="module")
def ctx():
    return bp.default_context()


@pytest.fixture(scope="module")
def problem(ctx):
    n, a, d = (1 << 17) + 5, Q - 13579, 0x1122334455667788
    h = ctx.srs_generate_progression(n, a, d)
    sets = {}
    rnd = O.splitmix_scalars(n, 0xB1D5)
    rnd[7] = 0
    rnd[8] = bp.scalar_from_int(Q - 1)
    sets["random"] = rnd
    sets["all_equal"] = np.tile(bp.scalar_from_int(0x0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF % Q), (n, 1))       # one bucket per window holds n entries
    sets["zero_one"] = bp.scalars_from_ints([i & 1 for i in range(n)])                                                       # one bucket of n / 2 entries in all
    sets["bytes16"] = bp.scalars_from_ints([(i * 7919) % 65536 for i in range(n)])                                           # witness-like small values
    want = {k: M.enc96(M.ec_mul(oracle_dot(v, a, d))) for k, v in sets.items()}
    yield h, n, sets, want
    ctx.srs_free(h)


@pytest.mark.parametrize("tables", [1, 13, 16, 18, 20, 256 + 12], ids=lambda t: "tables_%d" % t)
@pytest.mark.parametrize("build", BUILDS, ids=lambda b: ",".join("%s=%s" % (k[7:], v) for k, v in b.items()))
def test_builds_agree(ctx, problem, tables, build, monkeypatch):
    h, n, sets, want = problem
    for k, v in build.items():
        monkeypatch.setenv(k, v)
    info = ctx.srs_precompute(h, tables)
    assert info["window_bits"] == (0 if tables == 1 else tables)
    for name, sc in sets.items():
        assert ctx.msm(h, sc) == want[name], (name, tables, build)
        st = ctx.msm_stats()
        assert st["tables"] == (tables != 1) and 0 < st["mixed_adds"] <= 64 * n
    # a prefix that ends inside a slice and a shard at an offset, through the same build
    m = n // 3 + 11
    rnd = sets["random"]
    assert bp.sum_partials(ctx.msm_partial(h, rnd[:m], first=29)) == M.enc96(M.ec_mul(oracle_dot(rnd[:m], Q - 13579 + 29 * 0x1122334455667788, 0x1122334455667788)))
    ctx.srs_precompute(h, bp.SRS_TABLES_OFF)


def test_environment_knobs_out_of_range_are_ignored(ctx, problem, monkeypatch):
    """a stray variable in the embedding process must not change kernel shapes or divide by zero"""
    h, n, sets, want = problem
    for k, v in {"BP_MSM_CHUNK": "0", "BP_MSM_SLICES": "0", "BP_MSM_SEG": "0", "BP_MSM_C": "99", "BP_MSM_SORT": "7", "BP_MSM_PART_SLICE": "3",
                 "BP_MSM_RADIX_BITS": "x", "BP_MSM_PLANES_WIDE_MIN": "0", "BP_MSM_FIXUP": "-1"}.items():
        monkeypatch.setenv(k, v)
    assert ctx.msm(h, sets["random"]) == want["random"]
    ctx.srs_precompute(h, 0)
    assert ctx.msm(h, sets["random"]) == want["random"]
    ctx.srs_precompute(h, bp.SRS_TABLES_OFF)

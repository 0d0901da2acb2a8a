"""-m gpu: Fr NTT through the C ABI vs the oracle (faithful O(n^2) restatement of src/utils.rs:63-129 at
small n, its O(n log n) twin elsewhere).  Bit-exact on the Montgomery limbs."""
import random

import numpy as np
import pytest

import baby_plonk_rust_amd as bp
from oracle import oracle as O
from tests import bigint_model as M
from tests.gpu_common import NTHREADS, Q

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return bp.default_context()


def frs(vals):
    return bp.scalars_from_ints([v % Q for v in vals])


@pytest.mark.parametrize("logn", list(range(0, 13)))
def test_small_sizes_vs_oracle(ctx, logn):
    n = 1 << logn
    a = O.splitmix_scalars(n, 0xF40000 + logn)
    fwd = bp.ntt_381(a, ctx)
    assert (fwd == O.ntt_fast(a)).all()
    inv = bp.i_ntt_381(a, ctx)
    assert (inv == O.ntt_fast(a, inverse=True)).all()
    assert (bp.i_ntt_381(fwd, ctx) == a).all()
    if logn <= 7:                                     # the reference's literal algorithm
        assert (fwd == O.ntt_381(a)).all() and (inv == O.i_ntt_381(a)).all()
    if logn <= 5:                                     # independent big-int DFT
        assert bp.scalars_to_ints(fwd) == M.dft(bp.scalars_to_ints(a))


def test_simple_vectors_and_errors(ctx):
    assert bp.scalars_to_ints(bp.ntt_381(frs([1, 0, 0, 0, 0, 0, 0, 0]), ctx)) == [1] * 8
    assert bp.scalars_to_ints(bp.ntt_381(frs([3, 3]), ctx)) == [6, 0]           # setup.rs:128-135 input
    assert bp.scalars_to_ints(bp.i_ntt_381(frs([6, 0]), ctx)) == [3, 3]
    for bad in (3, 6, 12, 1000):
        with pytest.raises(bp.BpError) as e:
            bp.ntt_381(frs(list(range(bad))), ctx)                              # utils.rs:65 assert!
        assert e.value.code == -2
    # canonical little-endian format
    vals = [5, 7, Q - 1, 0]
    le = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint64).reshape(-1, 4)
    out = ctx.ntt(le, fmt=bp.FR_BYTES_LE)
    got = [int.from_bytes(out[i].tobytes(), "little") for i in range(4)]
    assert got == M.dft(vals)
    # roots_of_unity (utils.rs:45-52) and test_root_of_unity (utils.rs:239-242)
    w4 = bp.scalar_to_int(bp.root_of_unity(4))
    assert pow(w4, 4, Q) == 1 and pow(w4, 2, Q) != 1
    assert bp.scalars_to_ints(bp.roots_of_unity(8, ctx)) == [pow(M.omega(8), i, Q) for i in range(8)]
    assert (bp.roots_of_unity(1 << 12, ctx) == _roots_oracle(1 << 12)).all()


def _roots_oracle(n):
    out = O.u64((n, 4))
    O.lib.ntt_roots_of_unity(out.ctypes.data, n)
    return out


@pytest.mark.parametrize("logn", [13, 14, 15, 16, 17, 18, 20])
def test_multi_pass_sizes_vs_oracle(ctx, logn):
    """BASELINE configs[1] (2^16) and the 2^20 size, both directions"""
    n = 1 << logn
    a = O.splitmix_scalars(n, 0xF40000 + logn)
    fwd = bp.ntt_381(a, ctx)
    assert (fwd == O.ntt_fast(a, threads=NTHREADS)).all()
    inv = bp.i_ntt_381(a, ctx)
    assert (inv == O.ntt_fast(a, inverse=True, threads=NTHREADS)).all()


def test_batch_columns_and_stride(ctx):
    """independent columns (the multi-GPU sharding unit): batch with a stride larger than the length"""
    for logn, batch, row in ((6, 5, 64), (9, 3, 700), (12, 4, 4096 + 64), (14, 3, 1 << 14)):
        n = 1 << logn
        data = O.splitmix_scalars(batch * row, 0xBA7C4 + logn).reshape(batch, row, 4)
        out = ctx.ntt_batch(data, stride=n)
        for b in range(batch):
            assert (out[b, :n] == O.ntt_fast(data[b, :n])).all()
            assert (out[b, n:] == data[b, n:]).all()            # padding untouched
        back = ctx.ntt_batch(out, inverse=True, stride=n)
        assert (back == data).all()


def test_2p24_round_trip_and_spot_checks(ctx):
    """the 2^24 size: inverse(forward(x)) == x, linearity, and spot outputs against the direct sum"""
    import torch
    logn, n = 24, 1 << 24
    a = O.splitmix_scalars(n, 0xF40018)
    t = torch.from_numpy(a.view(np.int64).copy()).cuda()
    torch.cuda.synchronize()
    ctx.ntt_device(t.data_ptr(), logn)
    fwd = t.cpu().numpy().view(np.uint64)
    w = M.omega(n)
    for x in (0, 1, 12345, n // 2 + 77, n - 1):
        # out[x] = sum_y in[y] w^(xy) = polynomial with coefficients `in` evaluated at w^x
        assert (fwd[x] == O.poly_eval(a, O.fr_from_int(pow(w, x, Q)), fast=True)).all()
    ctx.ntt_device(t.data_ptr(), logn, inverse=True)
    assert (t.cpu().numpy().view(np.uint64) == a).all()
    assert ctx.ntt_stats()["passes"] == 3


def test_2p24_full_array_vs_oracle(ctx):
    """the 2^24 size element for element: forward and inverse transforms of 2^24 elements in HBM equal the oracle's
    O(n log n) twin of utils.rs:63-129 on every one of the 2^24 outputs (the twin itself equals the literal O(n^2) restatement up to
    2^10, tests/test_oracle_msm_ntt_poly.py)"""
    import torch
    logn, n = 24, 1 << 24
    a = O.splitmix_scalars(n, 0xF40024)
    t = torch.from_numpy(a.view(np.int64).copy()).cuda()
    torch.cuda.synchronize()
    ctx.ntt_device(t.data_ptr(), logn)
    want = O.ntt_fast(a, threads=NTHREADS)
    assert (t.cpu().numpy().view(np.uint64).reshape(-1, 4) == want).all()
    del want
    t.copy_(torch.from_numpy(a.view(np.int64)))
    torch.cuda.synchronize()
    ctx.ntt_device(t.data_ptr(), logn, inverse=True)
    assert (t.cpu().numpy().view(np.uint64).reshape(-1, 4) == O.ntt_fast(a, inverse=True, threads=NTHREADS)).all()


def test_async_transforms_back_to_back(ctx):
    """bp_ntt_fr_device_async: transforms enqueued on the context's stream without a host wait in between"""
    import torch
    x = O.splitmix_scalars(1 << 14, 0xA5)
    t = torch.from_numpy(x.view(np.int64).copy()).cuda()
    torch.cuda.synchronize()
    ctx.ntt_device_async(t.data_ptr(), 14)                       # forward, inverse, forward: ends as one forward transform
    ctx.ntt_device_async(t.data_ptr(), 14, inverse=True)
    ctx.ntt_device_async(t.data_ptr(), 14)
    ctx.synchronize()
    assert (t.cpu().numpy().view(np.uint64).reshape(-1, 4) == O.ntt_fast(x)).all()
    st = ctx.ntt_stats()
    assert st["device_ms"] > 0 and st["passes"] == 2
    ctx.ntt_device(t.data_ptr(), 14, inverse=True)               # a blocking call after enqueued ones
    assert (t.cpu().numpy().view(np.uint64).reshape(-1, 4) == x).all()

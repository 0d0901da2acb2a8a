"""The N > 1 path on CPU: world_size 2 (and 3) over gloo.  The per-rank MSM partial is produced by the CPU oracle here
(no GPU in this container); what is under test is the product's sharding arithmetic, the all-gather of the 144-byte
partials and the host-side combine (bp_g1_sum_partials from libbp_msm_ntt.so)."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from baby_plonk_rust_amd import dist as bpd


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        a, d = 424242, 171717
        lo, hi = bpd.shard_range(n, rank, world)
        # this rank's point range and scalar slice of the global problem
        aff = O.points_progression(hi - lo, a + lo * d, d)
        sc = O.splitmix_scalars(n, 0xD157)[lo:hi]
        partial = O.bucket_msm(O.affine_to_proj(aff), sc).tobytes()       # 144-byte G1Projective image
        got = bpd.combine_partials(partial)
        cols = {j: np.full((4, 4), j, dtype=np.uint64) for j in bpd.my_columns(5, rank, world)}
        allc = bpd.all_gather_columns(cols, 5)
        q.put((rank, got, [int(c[0, 0]) for c in allc]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 64), (3, 50)])
def test_point_range_sharding_allgather_combine(world, n):
    from oracle import oracle as O
    from tests import bigint_model as M
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = M.enc96(M.ec_mul(O.dot_progression(O.splitmix_scalars(n, 0xD157), 424242, 171717)))
    for rank, got, cols in res:
        assert got == want, rank                    # every rank ends with the same, correct commitment
        assert cols == [0, 1, 2, 3, 4]


def test_shard_range_partition():
    for n in (0, 1, 7, 64, 1000, 1 << 20):
        for world in (1, 2, 3, 4, 8):
            edges = [bpd.shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            for (a, b), (c, d) in zip(edges, edges[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    assert [bpd.column_owner(j, 4) for j in range(6)] == [0, 1, 2, 3, 0, 1]
    assert sorted(sum((bpd.my_columns(23, r, 8) for r in range(8)), [])) == list(range(23))


def test_bench_refuses_to_fake_ranks_without_gpus():
    """`python bench.py --gpus N` must start N ranks or fail loudly -- never run one rank and call it N.
    In this container no GPU is visible, so any N > 1 is refused before anything is launched."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         timeout=300, cwd=root, env=env)
    assert out.returncode == 2 and "--gpus 2" in out.stderr and not out.stdout.strip()

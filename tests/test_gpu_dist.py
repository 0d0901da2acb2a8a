"""-m gpu: the N > 1 path with real GPU partials: two ranks (gloo, both on GPU 0 -- this box has one card), each holding a
point-range shard of the SRS in HBM, all-gather of the 144-byte partials, host combine.  Also rehearses bench.py's N = 2
control flow end to end (barriers, max-over-ranks timing, rank-0 JSON line)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import baby_plonk_rust_amd as bp
        from baby_plonk_rust_amd import dist as bpd
        from oracle import oracle as O
        ctx = bp.Context(0)
        a, d = 424242, 171717
        lo, hi = bpd.shard_range(n, rank, world)
        srs = ctx.srs_generate_progression(hi - lo, a + lo * d, d)            # this rank's point range, resident in HBM
        sc = O.splitmix_scalars(n, 0xD157)[lo:hi]
        q.put((rank, bpd.msm_sharded(ctx, srs, sc)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_point_range_shards_on_one_gpu():
    from oracle import oracle as O
    from tests import bigint_model as M
    world, n = 2, 5000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = M.enc96(M.ec_mul(O.dot_progression(O.splitmix_scalars(n, 0xD157), 424242, 171717)))
    assert all(got == want for _, got in res)


def test_bench_two_rank_rehearsal():
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-n", "14",
           "--ntt-log-n", "14", "--prove-log-n", "10", "--prove-reps", "2", "--backend", "gloo", "--skip-cpu"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0 and line["ntt"]["value"] > 0
    assert line["config"]["msm_points_per_gpu"] == 1 << 14
    assert line["prove"]["gates"] == 1 << 10 and line["prove"]["value"] > 0 and line["prove"]["parallelism"] == "independent proofs x2"

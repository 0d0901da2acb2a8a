#!/usr/bin/env python3
"""host-side split of one step of the one-process-per-GPU MSM path (dist.ShardedMsm) under RCCL with the ranks given by the launcher:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/exchange_split.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import baby_plonk_rust_amd as bp
from baby_plonk_rust_amd import _lib, api
from baby_plonk_rust_amd import dist as bpd

rank = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
torch.cuda.set_device(rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
ctx = bp.Context(rank)
n = 1 << 20
srs = ctx.srs_generate_progression(n, 12345, 6789)
ctx.srs_precompute(srs, 0)
t = torch.empty(n * 4, dtype=torch.int64, device="cuda")
ctx.synthetic_scalars_device(t.data_ptr(), n, 7)
torch.cuda.synchronize()
ex = bpd.ShardedMsm(ctx)
for _ in range(5):
    want = ex(srs, None, device_ptr=t.data_ptr(), n=n)
# the same steps by hand, timed on the host
acc = [0.0] * 7
reps = 20
for _ in range(reps):
    t0 = time.perf_counter()
    ex.stream.wait_stream(torch.cuda.current_stream(ex.gpu))
    ex.ctx.msm_blob_device(srs, ex.mine.data_ptr(), None, first=0, device_ptr=t.data_ptr(), n=n, wait=False)
    t1 = time.perf_counter()
    with torch.cuda.stream(ex.stream):
        ex.record_done.record()
        one = ex.host[:_lib.MSM_BLOB_BYTES]
        t2 = time.perf_counter()
        dist.all_gather_into_tensor(ex.gathered, ex.mine)
        t3 = time.perf_counter()
        ex.ctx.msm_blobs_sum_device(ex.gathered.data_ptr(), ex.world, ex.summed.data_ptr(), wait=False)
        one.copy_(ex.summed, non_blocking=True)
        ex.all_done.record()
        t4 = time.perf_counter()
        ex.ctx.synchronize()
        t5 = time.perf_counter()
    out = api.combine_blobs(one.numpy().tobytes())
    t6 = time.perf_counter()
    assert out == want
    for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t6 - t0)):
        acc[i] += d
names = ["enqueue MSM", "record", "all_gather call", "sum + copy + record", "wait", "combine", "TOTAL"]
if dist.get_rank() == 0:
    print("world %d; per step, us: " % dist.get_world_size() + "  ".join("%s %.0f" % (k, 1e6 * v / reps) for k, v in zip(names, acc)), flush=True)
    print("device ms of the MSM: %.3f; GPU-side exchange %.3f ms" % (ctx.msm_stats()["device_ms"], 1e3 * ex.exchange_s), flush=True)
    # the plain path for comparison
    ex.close()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = ctx.msm_partial(srs, None, device_ptr=t.data_ptr(), n=n)
    print("plain bp_msm_g1_partial: %.0f us per step" % (1e6 * (time.perf_counter() - t0) / reps), flush=True)
dist.destroy_process_group()

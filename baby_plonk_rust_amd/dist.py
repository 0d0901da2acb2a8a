"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

MSM  shards by contiguous POINT RANGE: rank r keeps SRS[lo_r, hi_r) resident in its HBM and receives
     the matching scalar slice; each rank runs the whole Pippenger pipeline on its slice and produces ONE 144-byte
     projective partial.  Elliptic-curve addition is not an RCCL reduction operator, so the "reduce" is a single
     all-gather followed by world-1 complete additions on every rank.  What travels is each rank's record of per-window
     partial sums, left in HBM by bp_msm_g1_blob_device (ShardedMsm); the 144-byte projective form (combine_partials)
     remains for partials that are already on the host.
NTT  shards by INDEPENDENT COLUMNS: polynomial j of a batch belongs to rank j mod world; no collective inside a
     transform.  all_gather_columns() is provided for callers that need every column everywhere afterwards.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _lib, api


def shard_range(n, rank, world):
    """contiguous point range [lo, hi) of rank `rank`; the first n % world ranks get one extra point"""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def column_owner(j, world):
    return j % world


def my_columns(n_columns, rank, world):
    return list(range(rank, n_columns, world))


def _backend_device(ctx_device, group=None):
    """tensors of a collective live on the GPU under RCCL ("nccl") and on the host under gloo (CPU rehearsals)"""
    return torch.device("cuda", ctx_device) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def allgather_partials(partial144, device=None, group=None):
    """144-byte variant (partials already on the host, e.g. produced by another library): every rank contributes 144 bytes"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bytes(partial144)
    world = dist.get_world_size(group)
    dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
    mine = torch.frombuffer(bytearray(partial144), dtype=torch.uint8).to(dev)
    out = torch.empty(world * 144, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.cpu().numpy().tobytes()


def combine_partials(partial144, device=None, group=None):
    """all-gather + world-1 complete additions + one affine normalisation -> 96-byte encoding, same on every rank"""
    return api.sum_partials(allgather_partials(partial144, device, group))


def join_library_communicator(ctx, group=None, device=None):
    """Give `ctx` the library's own RCCL communicator over the ranks of `group` (bp_comm_unique_id on rank 0, the 128 bytes carried by
    torch.distributed, bp_comm_init_rank everywhere).  Returns True when this call created it, False when every rank's context already
    had a matching one.  EVERY step is agreed on before the next one, and every rank issues the same collectives in the same order
    on every path -- a rank that fails at some step says so inside the next collective instead of staying away from it (ADVICE r05:
    a per-rank try/except around a collective leaves the other ranks waiting).  Any disagreement or failure raises on ALL ranks."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = device if device is not None else _backend_device(ctx.device, group)

    def agree_min(values):
        t = torch.tensor(values, dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return [int(v) for v in t.cpu()]

    # step 1: which contexts already hold a communicator of this shape?  1 = matching, 0 = none, -1 = another shape
    have = ctx.comm_info()
    mine = 1 if have == (rank, world) else (0 if have[1] == 0 else -1)
    lo, neg_hi = agree_min([mine, -mine])
    if lo < 0:
        raise RuntimeError("join_library_communicator: some rank's context holds a communicator of another shape (here: rank %d of %d)" % have)
    if lo == 1:
        return False
    if -neg_hi == 1:
        raise RuntimeError("join_library_communicator: some ranks' contexts already have a communicator and others have none")
    # step 2: rank 0 makes the id; its failure travels as None through the SAME broadcast every rank takes part in
    box, id_error = [None], None
    if rank == 0:
        try:
            box[0] = api.Context.comm_unique_id()
        except Exception as e:
            id_error = e
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group, device=dev)
    if box[0] is None:
        raise RuntimeError("join_library_communicator: rank 0 could not create the communicator id (%r)" % (id_error,))
    # step 3: ncclCommInitRank on every rank (bounded inside the library: bp_comm_set_timeout_ms), then agree on the outcome
    ok, init_error = 1, None
    try:
        ctx.comm_init_rank(box[0], rank, world)
    except Exception as e:
        ok, init_error = 0, e
    if agree_min([ok])[0] != 1:
        if ok:
            ctx.comm_destroy()
        raise RuntimeError("join_library_communicator: bp_comm_init_rank failed on some rank (here: %r)" % (init_error,))
    return True


class ShardedMsm:
    """The MSM path of one rank: the rank's partial sums stay in HBM, ONE all-gather of the records over RCCL/xGMI, the
    device-side pre-sum, ONE device-to-host copy, host combine.
    Under the "nccl" backend all of that is ONE call into the library (bp_msm_g1_allgather, capi_comm.hip: the communicator lives in
    the bp_ctx) and there is no other path: the Python mirror and a Rust host that binds the C ABI run the same lines, and a
    communicator that cannot be created is an error on every rank, not a silent detour (round 6; rounds 4-5 kept a torch all-gather
    fallback -- with no world > 1 ever run it was not knowable which of the two a first real run would have timed).
    Under gloo (CPU rehearsals of the control flow: RCCL refuses two ranks on one GPU) the same steps are spelled out here with a
    host-side gather; without a process group the record goes straight to the host.
    Stream-ordered end to end: the library context is put on a torch stream owned by this object (bp_set_stream) and the record,
    the collective, the pre-sum and the copy are enqueued on it in that order -- the copy's is the only host wait of a call.
    `exchange_s`: GPU time from "record complete" to "gathered, summed and copied" (two events on the stream).
    close() (or garbage collection) gives the context its own stream back."""

    def __init__(self, ctx, group=None):
        self.ctx, self.group = ctx, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.gpu = torch.device("cuda", ctx.device)
        self.stream = torch.cuda.Stream(self.gpu)
        ctx.set_stream(self.stream.cuda_stream)          # every later call on ctx is ordered on this stream
        self.collective = dist.is_initialized()          # under a launcher even a single rank goes through the collective
        self.on_gpu = self.collective and dist.get_backend(group) == "nccl"
        # torch.distributed only carries the 128-byte communicator id from rank 0 to the others, once
        self.c_path = self.on_gpu
        self.own_comm = join_library_communicator(ctx, group, self.gpu) if self.on_gpu else False
        if not self.c_path:
            with torch.cuda.stream(self.stream):
                self.mine = torch.zeros(_lib.MSM_BLOB_BYTES, dtype=torch.uint8, device=self.gpu)
            self.gathered = torch.empty(self.world * _lib.MSM_BLOB_BYTES, dtype=torch.uint8, device="cpu")
            # pinned landing area of the record's D2H (a fresh pageable tensor per call cost ~30 us of the step)
            self.host = torch.empty(_lib.MSM_BLOB_BYTES, dtype=torch.uint8).pin_memory() if torch.cuda.is_available() else None
            self.record_done = torch.cuda.Event(enable_timing=True)
            self.all_done = torch.cuda.Event(enable_timing=True)
        self.exchange_s = 0.0
        # The wait is bp_synchronize (the library polls its stream, which IS this torch stream): torch's Event.synchronize() came back
        # ~90 us after the GPU had finished (measured in round 4: 2 786 us per 2^20-point step against 2 650 without the exchange).

    def __call__(self, srs_handle_local, scalars_local=None, device_ptr=None, n=None, first=0):
        # another ShardedMsm on the same context (or its close()) may have moved the context to a different stream since __init__:
        # the record below must be written in THIS object's stream order, or the all-gather behind it reads a stale record
        self.ctx.set_stream(self.stream.cuda_stream)
        self.stream.wait_stream(torch.cuda.current_stream(self.gpu))        # behind whatever produced the scalars; enqueue only
        if self.c_path:                                                     # the whole exchange under the C ABI, on this stream
            out = self.ctx.msm_allgather(srs_handle_local, scalars_local, first=first, device_ptr=device_ptr, n=n)
            self.exchange_s = 1e-3 * self.ctx.comm_last_exchange_ms()
            return out
        self.ctx.msm_blob_device(srs_handle_local, self.mine.data_ptr(), scalars_local, first=first, device_ptr=device_ptr, n=n, wait=False)
        with torch.cuda.stream(self.stream):
            self.record_done.record()
            self.host.copy_(self.mine, non_blocking=True)
            self.all_done.record()
            self.ctx.synchronize()                                                            # the only host wait (see __init__)
            if self.collective:                                                               # gloo: host-side gather of the records
                dist.all_gather_into_tensor(self.gathered, self.host.clone(), group=self.group)
                host = self.gathered
            else:
                host = self.host
        out = api.combine_blobs(host.numpy().tobytes())
        self.exchange_s = 1e-3 * self.record_done.elapsed_time(self.all_done)      # GPU-side: record complete -> copied
        return out

    def close(self):
        if getattr(self, "ctx", None) is not None and getattr(self.ctx, "_h", None):
            self.ctx.set_stream(None)                    # waits for the stream, then back to the context's own
            if getattr(self, "own_comm", False) and self.ctx.comm_info()[1]:
                self.ctx.comm_destroy()
        self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def msm_sharded(ctx, srs_handle_local, scalars_local, device_ptr=None, n=None, group=None):
    """sum over ALL ranks' (point, scalar) pairs; `srs_handle_local` is this rank's point-range shard"""
    ex = ShardedMsm(ctx, group)
    try:
        return ex(srs_handle_local, scalars_local, device_ptr=device_ptr, n=n)
    finally:
        ex.close()


def all_gather_columns(local_columns, n_columns, group=None, device=None):
    """local_columns: {column index: [N, 4] uint64 array or int64 tensor} of this rank's finished columns (column j belongs
    to rank j mod world) -> list of all columns.  One all-gather of a [columns per rank, N, 4] tensor: on the GPU over
    RCCL/xGMI under "nccl" (32 MiB per 2^20-element column), on the host under gloo."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [local_columns[j] for j in range(n_columns)]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per_rank = (n_columns + world - 1) // world
    sample = next(iter(local_columns.values())) if local_columns else None
    as_tensor = isinstance(sample, torch.Tensor)
    if device is None:
        device = sample.device if as_tensor and dist.get_backend(group) == "nccl" else _backend_device(torch.cuda.current_device() if torch.cuda.is_available() else 0, group)
    shape = torch.tensor(list(sample.shape[:1]) if sample is not None else [0], dtype=torch.int64, device=device)
    dist.all_reduce(shape, op=dist.ReduceOp.MAX, group=group)            # ranks without a column still need the length
    N = int(shape[0])
    mine = torch.zeros((per_rank, N, 4), dtype=torch.int64, device=device)
    for j, v in local_columns.items():
        assert column_owner(j, world) == rank
        t = v if as_tensor else torch.from_numpy(np.ascontiguousarray(v, dtype=np.uint64).view(np.int64))
        mine[j // world].copy_(t.reshape(N, 4))
    out = torch.empty((world * per_rank, N, 4), dtype=torch.int64, device=device)        # rank r's block at [r * per_rank, ...)
    dist.all_gather_into_tensor(out, mine, group=group)
    cols = [out[(j % world) * per_rank + j // world] for j in range(n_columns)]
    return cols if as_tensor else [c.cpu().numpy().view(np.uint64) for c in cols]

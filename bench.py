#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path: G1 MSM scalar-muls/s (+ Fr NTT elements/s) on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM:
one 2^log_n-point G1 MSM per rank against that rank's resident SRS shard (point-range sharding),
followed -- only when N > 1 -- by the single RCCL all-gather of the 144-byte projective partials and the
7-addition combine.  Weak scaling: every rank owns 2^log_n points, the job computes one N * 2^log_n-point MSM.
The Fr NTT (independent columns, one 2^ntt_log_n vector per rank, no collective) is timed the same way in a
second region and reported in the "ntt" object of the same JSON line.

Workload at N = 1: BASELINE.json configs[2], the 2^20-point MSM the metric is quoted on (+ a 2^20 NTT).
The SRS is what Setup holds for the life of a prover (src/setup.rs:7-10): resident in HBM together with its
fixed-base window tables, built once by bp_srs_precompute before the timed region (build time and size are
reported).  The same MSM without tables (raw points only, the uncached bucket_msm seam) is timed beside it
in "msm_without_tables"; --no-tables makes that the headline instead.
Inputs: points P_i = (a + i d) G generated on the GPU, scalars = SplitMix64 ->
from_bytes_wide generated on the GPU; nothing is read from disk.  The CPU oracle is used for the
`cpu_baseline` leg only (rank 0, N = 1, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

import baby_plonk_rust_amd as bp
from baby_plonk_rust_amd import dist as bpd

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
# The dominant kernel is integer-issue bound: second ceiling from this repo's own measurements --
# tools/ubench_int.hip (profiles/r01_ubench_int_issue_rates.txt): v_mad_u64_u32 issues at 57 lanes/clk/CU, like a carry add
VALU_PEAK_LANE_INSTR_S = 57.0 * 256 * 2.4e9
ACC_INSTR_PER_ADD = 5000         # VALU instructions per iteration of msm_accumulate's loop (hipcc -S: 5006, of which 3729 v_mad_u64_u32)
MSM_BYTES_PER_UNIT = 128         # 32 B scalar + 96 B affine point per scalar-mul
NTT_BYTES_PER_UNIT = 64          # 32 B read + 32 B write per element
GOLDEN = 0x9E3779B97F4A7C15
A0, D0 = 0x1F2E3D4C5B6A79881122334455667788, 0x0102030405060708090A0B0C0D0E0F10


def measured_traffic(key):
    """HBM bytes per launch from the committed PMC passes (profiles/r01_hbm_traffic.json), or None when the
    running configuration is not the profiled one"""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as f:
            return json.load(f)[key]["hbm_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(sample_log_n, threads_all):
    """reference-faithful CPU path (oracle restatement of src/msm.rs: c = 4, 64 windows, projective adds),
    single thread like the reference, on a bounded sample of the same synthetic workload"""
    from oracle import oracle as O
    n = 1 << sample_log_n
    aff = O.points_progression(n, A0, D0)
    proj = np.zeros((n, 18), dtype=np.uint64)
    proj[:, :12] = aff[:, :12]
    proj[:, 12:] = O.fp_one()
    sc = O.splitmix_scalars(n, 0x5EED0000 + 20)
    t0 = time.perf_counter()
    r1 = O.bucket_msm(proj, sc, 256, 4, threads=1)
    t1 = time.perf_counter()
    r2 = O.bucket_msm(proj, sc, 256, 4, threads=threads_all)
    t2 = time.perf_counter()
    assert O.g1_eq(r1, r2)
    ntt_log = 20
    x = O.splitmix_scalars(1 << ntt_log, 0xF40014)
    t3 = time.perf_counter()
    O.ntt_fast(x)
    t4 = time.perf_counter()
    xq = O.splitmix_scalars(1 << 8, 0xF40008)
    t5 = time.perf_counter()
    O.ntt_381(xq)                                 # the reference's literal O(n^2) DFT with a 256-bit pow per term (utils.rs:63-81)
    t6 = time.perf_counter()
    return {
        "value": n / (t1 - t0), "unit": "scalar-muls/s", "cores": 1, "kind": "port",
        "sample": "2^%d-point bucket_msm(b=256,c=4) restated from src/msm.rs, same synthetic inputs, %.1f s" % (sample_log_n, t1 - t0),
        "all_cores": {"value": n / (t2 - t1), "cores": threads_all, "note": "same algorithm, 64 windows over OpenMP threads"},
        "ntt": {"value": (1 << ntt_log) / (t4 - t3), "unit": "elements/s", "cores": 1,
                "sample": "2^%d radix-2 NTT (output-identical O(n log n) twin of utils.rs:63-81), %.2f s" % (ntt_log, t4 - t3),
                "reference_quadratic_form": {"value": (1 << 8) / (t6 - t5), "unit": "elements/s",
                                             "sample": "utils.rs:63-81 as written (n^2 terms, one pow each) at n = 2^8, %.2f s; cost grows as n^2" % (t6 - t5)}},
        "host": "%d logical CPUs" % (os.cpu_count() or 0),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=20, help="MSM points per GPU = 2^log_n")
    ap.add_argument("--ntt-log-n", type=int, default=20, help="NTT length per GPU = 2^ntt_log_n")
    ap.add_argument("--cpu-sample-log-n", type=int, default=18)
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--prove-log-n", type=int, default=20, help="gates of the synthetic circuit of the proofs/s leg = 2^prove_log_n (0 = skip)")
    ap.add_argument("--prove-reps", type=int, default=3)
    ap.add_argument("--prove-streams", type=int, default=2, help="concurrent provers per GPU in the proofs/s throughput figure")
    ap.add_argument("--other-sizes", type=int, nargs="*", default=[16, 24], help="log2 sizes also measured at N = 1 (MSM with tables + NTT, 3 runs each)")
    ap.add_argument("--no-tables", action="store_true", help="headline = MSM on the raw SRS, no fixed-base window tables")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                                                      "the N > 1 control flow on a box with fewer GPUs than ranks)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and world > n_dev:
        raise SystemExit("bench.py: %d ranks but %d GPUs (one process per GPU)" % (world, n_dev))
    dev_index = local_rank % max(n_dev, 1)            # == local_rank except in a gloo rehearsal
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ctx = bp.Context(dev_index)
    n = 1 << args.log_n
    # this rank's point range [rank*n, (rank+1)*n) of the global progression, resident in HBM
    srs = ctx.srs_generate_progression(n, A0 + rank * n * D0, D0)
    scal = torch.empty(n * 4, dtype=torch.int64, device=dev)
    ctx.synthetic_scalars_device(scal.data_ptr(), n, (0x5EED0000 + args.log_n + GOLDEN * 8 * rank * n) & (2**64 - 1))

    def msm_step():
        # per-rank Pippenger on the resident shard, then (N > 1) the single RCCL all-gather of 144 B per rank
        # and the N-1 complete additions on every rank (baby_plonk_rust_amd/dist.py)
        return bpd.msm_sharded(ctx, srs, None, device_ptr=scal.data_ptr(), n=n)

    def timed_msm():
        for _ in range(args.warmup):
            res = msm_step()
        barrier()
        acc, devt = [], []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = msm_step()
            st = ctx.msm_stats()
            acc.append(st["accumulate_ms"])
            devt.append(st["device_ms"])
        barrier()
        return res, time.perf_counter() - t0, acc, devt, ctx.msm_stats()

    # secondary leg first: the other table setting, same inputs
    table_info, table_build_s = {"window_bits": 0, "windows": 0, "bytes": 0}, 0.0
    if args.no_tables:
        t0 = time.perf_counter()
        table_info = ctx.srs_precompute(srs, 0)
        table_build_s = time.perf_counter() - t0
        other = timed_msm()
        ctx.srs_precompute(srs, bp.SRS_TABLES_OFF)
    else:
        other = timed_msm()
        t0 = time.perf_counter()
        table_info = ctx.srs_precompute(srs, 0)
        table_build_s = time.perf_counter() - t0
    result, elapsed, acc_ms, dev_ms, stats = timed_msm()
    assert other[0] == result, "MSM with and without fixed-base tables disagree"
    assert stats["tables"] == (not args.no_tables)

    # ---- NTT leg (independent columns, no collective) ----
    nn = 1 << args.ntt_log_n
    vec = torch.empty(nn * 4, dtype=torch.int64, device=dev)
    ctx.synthetic_scalars_device(vec.data_ptr(), nn, (0xF40000 + args.ntt_log_n + GOLDEN * 8 * rank * nn) & (2**64 - 1))
    for _ in range(args.warmup):
        ctx.ntt_device(vec.data_ptr(), args.ntt_log_n)
    barrier()
    ntt_ms = []
    t1 = time.perf_counter()
    for _ in range(args.steps):
        ctx.ntt_device(vec.data_ptr(), args.ntt_log_n)
        ntt_ms.append(ctx.ntt_stats()["device_ms"])
    barrier()
    ntt_elapsed = time.perf_counter() - t1
    ntt_passes = ctx.ntt_stats()["passes"]

    # ---- the metric's other sizes (BASELINE: "at 2^20 / 2^24", 2^16 = configs[1]); single GPU only, best of 3 runs ----
    sizes = {}
    if world == 1:
        for lg in args.other_sizes:
            if lg == args.log_n or lg < 10 or lg > 26:
                continue
            m = 1 << lg
            ctx.srs_free(srs)
            del scal, vec
            torch.cuda.empty_cache()
            srs = ctx.srs_generate_progression(m, A0, D0)
            if not args.no_tables:
                ctx.srs_precompute(srs, 0)
            scal = torch.empty(m * 4, dtype=torch.int64, device=dev)
            ctx.synthetic_scalars_device(scal.data_ptr(), m, (0x5EED0000 + lg) & (2**64 - 1))
            vec = torch.empty(m * 4, dtype=torch.int64, device=dev)
            ctx.synthetic_scalars_device(vec.data_ptr(), m, (0xF40000 + lg) & (2**64 - 1))
            best_msm, best_ntt = None, None
            for _ in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.msm_partial(srs, None, device_ptr=scal.data_ptr(), n=m)
                dt = time.perf_counter() - t0
                best_msm = dt if best_msm is None or dt < best_msm else best_msm
                ctx.ntt_device(vec.data_ptr(), lg)
                dn = ctx.ntt_stats()["device_ms"] * 1e-3
                best_ntt = dn if best_ntt is None or dn < best_ntt else best_ntt
            st = ctx.msm_stats()
            sizes["2^%d" % lg] = {"msm_scalar_muls_per_s": m / best_msm, "msm_ms": 1e3 * best_msm, "msm_accumulate_ms": st["accumulate_ms"],
                                  "window_bits": st["window_bits"], "tables": st["tables"],
                                  "ntt_elements_per_s": m / best_ntt, "ntt_device_ms": 1e3 * best_ntt, "ntt_passes": ctx.ntt_stats()["passes"]}

    # ---- prover leg (BASELINE configs[4]): independent proofs per GPU (replicas, no collective) ----
    prove = None
    if args.prove_log_n:
        import hashlib
        import random
        from baby_plonk_rust_amd.synthetic import Q as FR_Q, chained_multiplications
        pn = 1 << args.prove_log_n
        ctx.srs_free(srs)
        del scal, vec
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        cols, pk = chained_multiplications(pn, 1000 + rank)
        t_circuit_host = time.perf_counter() - t0
        import threading
        t0 = time.perf_counter()
        provers = []
        for j in range(max(1, args.prove_streams)):              # each concurrent prover owns a context (= HIP stream), SRS tables, circuit
            pctx = ctx if j == 0 else bp.Context(dev_index)
            psetup = bp.Setup.generate_srs(pn + 6, 0x1234567 + args.prove_log_n, pctx, tables=not args.no_tables)
            provers.append(bp.Prover(psetup, bp.Circuit(pk, pctx)))
        t_setup = time.perf_counter() - t0
        wit = [torch.from_numpy(c.view(np.int64)).to(dev) for c in cols]
        blinders = [random.Random(5).randrange(1, FR_Q) for _ in range(11)]
        ptrs = [w.data_ptr() for w in wit]
        blobs = [p.prove_device(ptrs[0], ptrs[1], ptrs[2], None, blinders) for p in provers]      # warm-up (workspaces, NTT tables)
        assert all(b == blobs[0] for b in blobs)
        # latency: one prover, proofs back to back
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.prove_reps):
            blob = provers[0].prove_device(ptrs[0], ptrs[1], ptrs[2], None, blinders)
        barrier()
        single_elapsed = time.perf_counter() - t0
        round_ms = provers[0].last_stats()["round_ms"]
        # throughput: all provers of this GPU at once (one host thread each; the library calls release the GIL), so one
        # proof's latency-bound tails (bucket reduction, scans, host transcript) overlap another proof's bulk kernels
        def run(p):
            for _ in range(args.prove_reps):
                p.prove_device(ptrs[0], ptrs[1], ptrs[2], None, blinders)
        barrier()
        t0 = time.perf_counter()
        threads = [threading.Thread(target=run, args=(p,)) for p in provers]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        barrier()
        prove_elapsed = time.perf_counter() - t0
        prove = {"elapsed": prove_elapsed, "single_elapsed": single_elapsed, "round_ms": round_ms, "sha": hashlib.sha256(blob).hexdigest()[:16],
                 "setup_s": t_setup, "circuit_host_s": t_circuit_host, "streams": len(provers)}

    other_elapsed = other[1]
    if world > 1:
        t = torch.tensor([elapsed, ntt_elapsed, other_elapsed, prove["elapsed"] if prove else 0.0, prove["single_elapsed"] if prove else 0.0],
                         dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, ntt_elapsed, other_elapsed = float(t[0]), float(t[1]), float(t[2])
        if prove:
            prove["elapsed"], prove["single_elapsed"] = float(t[3]), float(t[4])

    if rank == 0:
        units = world * n * args.steps
        value = units / elapsed
        acc = float(np.mean(acc_ms)) * 1e-3
        achieved = MSM_BYTES_PER_UNIT * n / acc / 1e9
        ntt_t = float(np.mean(ntt_ms)) * 1e-3
        ntt_achieved = NTT_BYTES_PER_UNIT * nn / ntt_t / 1e9
        line = {
            "metric": "g1_msm_scalar_muls_per_s",
            "baseline_metric": "G1 MSM scalar-muls/s + Fr NTT elements/s at 2^20/2^24; proof bit-exact",
            "value": value, "unit": "scalar-muls/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 limbs (381-bit Fp Montgomery, 255-bit Fr)", "data": "synthetic",
            "config": {"workload": "2^%d-point BLS12-381 G1 MSM per GPU (global 2^%d x %d points, point-range shards, "
                                   "RCCL all-gather of 144-B partials), SRS %s; + 2^%d Fr NTT per GPU; BASELINE configs[2] at N=1"
                                   % (args.log_n, args.log_n, world,
                                      "raw points only" if args.no_tables else "resident with its fixed-base window tables (Setup)",
                                      args.ntt_log_n),
                       "msm_points_per_gpu": n, "window_bits": stats["window_bits"], "ntt_len_per_gpu": nn,
                       "srs_tables": {"used": stats["tables"], "window_bits": table_info["window_bits"], "windows": table_info["windows"],
                                      "bytes_per_gpu": table_info["bytes"], "build_s": table_build_s},
                       "parallelism": "point-range x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "msm_accumulate", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic("msm_accumulate_2p20_c16" + ("" if args.no_tables else "_tables"))
                         if (args.log_n == 20 and stats["window_bits"] == 16) else None,
                         "kernel_ms": acc * 1e3, "algorithmic_bytes_per_launch": MSM_BYTES_PER_UNIT * n,
                         "note": "integer-ALU bound by design (11 Fp mul per bucket add); see DESIGN.md"},
            "roofline_valu_issue": {"bound": "valu_issue (integer multiply-add)", "kernel": "msm_accumulate",
                                    "achieved": stats["mixed_adds"] * ACC_INSTR_PER_ADD / acc, "peak": VALU_PEAK_LANE_INSTR_S,
                                    "unit": "lane-instructions/s", "frac": stats["mixed_adds"] * ACC_INSTR_PER_ADD / acc / VALU_PEAK_LANE_INSTR_S,
                                    "note": "mixed additions per launch x static instruction count of the loop body / measured kernel time, against "
                                            "the measured v_mad_u64_u32 issue rate (57 lanes/clk/CU x 256 CU x 2.4 GHz)"},
            "msm_device_ms": float(np.mean(dev_ms)),
            ("msm_with_tables" if args.no_tables else "msm_without_tables"): {
                "value": units / other_elapsed, "unit": "scalar-muls/s", "ms_per_step": 1e3 * other_elapsed / args.steps,
                "device_ms": float(np.mean(other[3])), "accumulate_ms": float(np.mean(other[2])), "window_bits": other[4]["window_bits"]},
            "ntt": {"metric": "fr_ntt_elements_per_s", "value": world * nn * args.steps / ntt_elapsed, "unit": "elements/s",
                    "ms_per_step": 1e3 * ntt_elapsed / args.steps, "passes": ntt_passes,
                    "roofline": {"bound": "hbm", "kernel": "ntt_pass_* (all passes of one transform)", "achieved": ntt_achieved,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ntt_achieved / HBM_PEAK_GBS,
                                 "traffic": measured_traffic("ntt_2p20_two_passes") if (args.ntt_log_n == 20 and ntt_passes == 2) else None,
                                 "kernel_ms": ntt_t * 1e3, "algorithmic_bytes_per_launch": NTT_BYTES_PER_UNIT * nn}},
            "result_sha": __import__("hashlib").sha256(result).hexdigest()[:16],
        }
        if sizes:
            line["other_sizes"] = sizes
        if prove:
            line["prove"] = {"metric": "plonk_proofs_per_s", "value": world * prove["streams"] * args.prove_reps / prove["elapsed"], "unit": "proofs/s",
                             "gates": 1 << args.prove_log_n, "concurrent_provers_per_gpu": prove["streams"],
                             "latency_ms_per_proof_single_prover": 1e3 * prove["single_elapsed"] / args.prove_reps,
                             "proofs_per_s_single_prover_per_gpu": args.prove_reps / prove["single_elapsed"],
                             "round_ms": prove["round_ms"], "proofs_timed_per_gpu": prove["streams"] * args.prove_reps,
                             "parallelism": "independent proofs x%d" % world,
                             "workload": "bp_prove: prover.rs rounds 1-5 + host transcript on a synthetic 2^%d-gate circuit (chained "
                                         "multiplications), witness and circuit resident in HBM, 624-byte proof out; BASELINE configs[4]"
                                         % args.prove_log_n,
                             "proof_sha_rank0": prove["sha"], "srs_and_circuit_setup_s": prove["setup_s"],
                             "synthetic_circuit_host_s": prove["circuit_host_s"]}
        if world == 1 and not args.skip_cpu:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample_log_n, min(64, os.cpu_count() or 1))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

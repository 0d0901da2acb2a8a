#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path: G1 MSM scalar-muls/s (+ Fr NTT elements/s, proofs/s) on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1 needs N ranks, one per GPU.  Launched by `python -m torch.distributed.run ... bench.py --gpus N ...` every process is a
rank (RANK / LOCAL_RANK / WORLD_SIZE in the environment).  Launched bare (`python bench.py --gpus N`, WORLD_SIZE unset) the
parent starts that same torch.distributed.run command as a child process BEFORE it makes any GPU call, relays rank 0's JSON
line and exits with the child's code; it refuses loudly when fewer than N GPUs are visible.

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM.  Legs, each timed
with W warm-up steps and exactly K steps between barrier + synchronize, max over ranks:
  value / weak scaling   one 2^log_n-point G1 MSM per rank against that rank's resident SRS shard (point-range sharding,
                         SURVEY.md 8e) + for N > 1 the single RCCL all-gather of the ranks' partial-sum records and the
                         combine: the job computes one N * 2^log_n-point MSM.  BASELINE configs[2] at N = 1.
  strong_scaling         BASELINE configs[3]: ONE 2^24-point MSM, 2^24 / N points per rank, same exchange; its result hash
                         is the same for every N.
  ntt / sizes            2^20 Fr NTT per rank (independent columns, no collective); at N = 1 also the metric's other sizes
                         (2^16 = configs[1], 2^24) with their own roofline objects.
  ntt_columns (ranks)    23 independent 2^20-element columns dealt over the ranks (column j -> rank j mod N), every rank transforms
                         its share, ONE all-gather of the finished columns over RCCL; elements/s and the all-gather's ms.
  seams (N = 1)          msm_host_scalars: pageable host scalars in (what Setup::commit(&Polynomial) hands over, PCIe
                         inclusive); msm_uncached_seam: upload 2^log_n points + multiply + free (bucket_msm(&[G1Projective])).
  prove                  BASELINE configs[4]: bp_prove on a synthetic 2^20-gate circuit; independent proofs per GPU
                         (throughput) and, for N > 1, ONE proof on a context spanning all N GPUs (latency, row e3).
The SRS is what Setup holds for the life of a prover (src/setup.rs:7-10): resident in HBM together with its fixed-base
window tables, built once before the timed region (build time and size reported); the same MSM on raw points is timed
beside it (`msm_without_tables`), --no-tables makes that the headline.
Inputs: points P_i = (a + i d) G and scalars SplitMix64 -> from_bytes_wide, both generated on the
GPU.  The CPU oracle is used for the `cpu_baseline` leg only (rank 0, N = 1, bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
# msm_accumulate is integer-issue bound.  Issue rates measured by tools/ubench_int.hip on this part
# (profiles/r01_ubench_int_issue_rates.txt), lanes per clock per CU: v_mad_u64_u32 and the other quarter-rate integer ops
# 57, plain 32-bit VALU ops 90; 256 CUs, 2.4 GHz peak engine clock.
RATE_QUARTER, RATE_FULL, N_CU, CLOCK_HZ = 57.0, 90.0, 256, 2.4e9
MSM_BYTES_PER_UNIT = 128         # 32 B scalar + 96 B affine point per scalar-mul
NTT_BYTES_PER_UNIT = 64          # 32 B read + 32 B write per element
NTT_COLUMNS = 23                 # independent transforms of one proof (prover.rs:386-450): the batch of the N > 1 columns leg
GOLDEN = 0x9E3779B97F4A7C15
A0, D0 = 0x1F2E3D4C5B6A79881122334455667788, 0x0102030405060708090A0B0C0D0E0F10
MASK64 = 2**64 - 1


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=20, help="weak-scaling MSM points per GPU = 2^log_n")
    ap.add_argument("--ntt-log-n", type=int, default=20, help="NTT length per GPU = 2^ntt_log_n")
    ap.add_argument("--group-legs-child", default="", help=argparse.SUPPRESS)       # internal: JSON job of the child that runs the bp_init_multi legs
    ap.add_argument("--strong-log-n", type=int, default=24, help="strong-scaling leg: ONE 2^strong_log_n-point MSM over all GPUs (0 = skip)")
    ap.add_argument("--strong-steps", type=int, default=0, help="steps of the strong-scaling leg (0 = min(steps, 10))")
    ap.add_argument("--cpu-sample-log-n", type=int, default=18)
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-seams", action="store_true")
    ap.add_argument("--skip-pipelined", action="store_true", help="no three-commitments-in-flight leg (its overlapping kernels would inflate the "
                                                                   "per-kernel averages of a rocprofv3 --kernel-trace run of the serial legs)")
    ap.add_argument("--skip-group-legs", action="store_true", help="N > 1: do not run the two bp_init_multi legs (group_commit, one proof over all GPUs)")
    ap.add_argument("--group-legs-timeout", type=int, default=240, help="seconds after which a bp_init_multi leg's child process is killed (its entry then says so)")
    ap.add_argument("--prove-log-n", type=int, default=20, help="gates of the synthetic circuit of the proofs/s leg = 2^prove_log_n (0 = skip)")
    ap.add_argument("--prove-reps", type=int, default=3)
    ap.add_argument("--prove-streams", type=int, default=3, help="concurrent provers per GPU in the proofs/s throughput figure "
                    "(measured on one MI355X at 2^20 gates, tools/defaults_sanity.sh: 1 -> 33.5, 2 -> 35.4, 3 -> 36.2, 4 -> 36.3 proofs/s)")
    ap.add_argument("--other-sizes", type=int, nargs="*", default=[16, 24], help="log2 sizes also measured at N = 1 (MSM with tables + NTT)")
    ap.add_argument("--no-tables", action="store_true", help="headline = MSM on the raw SRS, no fixed-base window tables")
    ap.add_argument("--leg-timeout", type=int, default=900, help="under a launcher: seconds one leg may take before the rank reports which leg it was "
                                                                  "stuck in and exits non-zero (the lines of the finished legs are out by then); 0 = no watchdog")
    ap.add_argument("--comm-timeout-ms", type=int, default=120000, help="bound of every wait behind a library collective and of ncclCommInitRank "
                                                                         "(bp_comm_set_timeout_ms): a missing or stuck rank is BP_ERR_COMM, not a stall")
    ap.add_argument("--rehearse-hang-after", default="", help=argparse.SUPPRESS)     # tests: the LAST rank stops taking part after the named leg
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                                                      "the N > 1 control flow on a box with fewer GPUs than ranks)")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: become the launcher.  Nothing here touches the GPU
    (torch.cuda.device_count() does not initialise it), the ranks are fresh child processes."""
    import torch
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and n_dev < args.gpus:
        sys.stderr.write("bench.py: --gpus %d needs %d ranks with one GPU each, but %d GPU%s visible\n"
                         % (args.gpus, args.gpus, n_dev, " is" if n_dev == 1 else "s are"))
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env, cwd=ROOT)
    # Rank 0 prints a COMPLETE line after the weak leg and again after every later leg (`legs_finished` grows, `provisional` turns false
    # on the last one).  Each is relayed the moment it arrives: a hang or a kill in a later leg -- no world > 1 has ever run -- leaves the
    # newest finished line in the caller's tail instead of nothing.
    lines, final = 0, False
    for out in child.stdout:
        if out.startswith('{"metric"'):
            lines += 1
            final = '"provisional": false' in out
            sys.stdout.write(out if out.endswith("\n") else out + "\n")
            sys.stdout.flush()
        else:
            sys.stderr.write(out)
    rc = child.wait()
    if rc != 0 or not final:
        sys.stderr.write("bench.py: the %d-rank run failed (exit code %d; %d result line%s relayed, the last one %s)\n"
                         % (args.gpus, rc, lines, "" if lines == 1 else "s", "final" if final else "provisional or missing"))
        return rc or 1
    return 0


# Numbers that need a separate pass (PMC counters, the static instruction mix, the register-only microbenchmarks) are READ from the
# committed files of the newest round that has them -- never typed in here -- and every figure derived from them names its file.
PROFILE_ROUNDS = ("r06", "r05", "r04", "r03", "r02")


def profile_lookup(suffix, key):
    """(value, file name) of `key` in the newest profiles/rNN_<suffix> that holds it; (None, None) when the running configuration was not
    profiled (e.g. another size or table width)"""
    for rnd in PROFILE_ROUNDS:
        name = "%s_%s" % (rnd, suffix)
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                v = json.load(f).get(key)
        except Exception:
            continue
        if v is not None:
            return v, name
    return None, None


def ubench_rates():
    """register-only rates of the two inner operations on the whole chip, from the newest profiles/rNN_ubench_valu_floor.txt
    (tools/ubench_g1add, tools/ubench_fr29 run on the GPU box): best additions/s of g1_add_mixed28, fewest clocks per wave-butterfly"""
    import re
    for rnd in PROFILE_ROUNDS:
        name = "%s_ubench_valu_floor.txt" % rnd
        try:
            text = open(os.path.join(ROOT, "profiles", name)).read()
        except Exception:
            continue
        adds = [float(m) for m in re.findall(r"^g1_add_mixed28 .*?([0-9.]+e[+-]?[0-9]+) additions/s", text, flags=re.M)]
        clks = [float(m) for m in re.findall(r"butterfl.*?([0-9.]+) clk per butterfly per SIMD", text)]
        if adds and clks:
            return {"adds_per_s": max(adds), "butterfly_clk": min(clks), "source": "profiles/" + name}
    return {"adds_per_s": 7.343e9, "butterfly_clk": 1311.0, "source": "fallback constants (no profiles/rNN_ubench_valu_floor.txt found)"}


UBENCH = ubench_rates()
BARE_ADDS_PER_S = UBENCH["adds_per_s"]                                           # register-only g1_add_mixed28, whole chip
BARE_BUTTERFLIES_PER_S = 1024 * 64 / UBENCH["butterfly_clk"] * 2.4e9             # register-only fr29_butterfly, 1 024 SIMDs at 2.4 GHz


def msm_roofline(n, acc_s, adds, window_bits, tables, traffic_key=None):
    """HBM roofline (the contract's) and the integer-issue roofline (the one that binds) of msm_accumulate for one launch"""
    achieved = MSM_BYTES_PER_UNIT * n / acc_s / 1e9
    traffic, traffic_file = profile_lookup("hbm_traffic.json", traffic_key) if traffic_key else (None, None)
    designed = (128 * adds + 4 * adds) if adds else None       # one 128-byte point / table slot + one sorted index per bucket addition
    hbm = {"bound": "hbm", "kernel": "msm_accumulate", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
           "traffic_source": ("profiles/%s: %s" % (traffic_file, traffic.get("source"))) if traffic else None,
           "kernel_ms": acc_s * 1e3, "algorithmic_bytes_per_launch": MSM_BYTES_PER_UNIT * n,
           "designed_bytes_per_launch": designed,
           "designed_frac": (designed / acc_s / 1e9 / HBM_PEAK_GBS) if designed else None,
           "designed_bytes_note": "what the kernel moves by design: a gathered 128-B slot and a 4-B list entry per bucket addition "
                                  "(13 per scalar at 20-bit windows with tables, 16 at 16 bits without), not the 128 B per scalar of the "
                                  "algorithmic figure; designed_frac = designed bytes / kernel time / peak, frac = algorithmic bytes / kernel time / peak",
           "note": "integer-issue bound by design (11 Fp products + 8 reductions per bucket addition); see roofline_valu_issue"}
    mix, mix_file = profile_lookup("msm_accumulate_instr_mix.json", "msm_accumulate<2>")
    issue = None
    if mix and adds:
        # lane-cycles of one loop iteration at the measured issue rates, times the additions actually performed
        per_add_s = (mix["quarter_rate"] / RATE_QUARTER + mix["full_rate"] / RATE_FULL) / (N_CU * CLOCK_HZ)
        ideal_s = adds * per_add_s
        issue = {"bound": "valu_issue", "kernel": "msm_accumulate", "achieved": adds / acc_s, "own_formula_register_only_rate": 1.0 / per_add_s,
                 "unit": "bucket additions/s", "frac": ideal_s / acc_s, "additions_per_launch": adds,
                 "frac_is": "a fraction of what THIS addition formula (complete mixed addition, 14 x 28-bit limbs: 11 products + 8 reductions) "
                            "reaches on registers only -- it says the loop is well scheduled, not that ~5 150 instructions per addition are "
                            "necessary; not a hardware peak",
                 "valu_per_addition": {"quarter_rate": mix["quarter_rate"], "full_rate": mix["full_rate"], "v_mad_u64_u32": mix.get("v_mad_u64_u32")},
                 "note": "additions counted by the kernel pipeline (non-zero digits) x static instruction classes of the loop body "
                         "(tools/instr_mix.py on the shipped code object: profiles/%s) at the issue rates tools/ubench_int.hip measured "
                         "(quarter-rate 57, full-rate 90 lanes/clk/CU, 256 CU, 2.4 GHz), over the measured kernel time" % mix_file,
                 "bare_kernel_peak": {"value": BARE_ADDS_PER_S, "frac": adds / acc_s / BARE_ADDS_PER_S,
                                      "source": "tools/ubench_g1add.hip: the same g1_add_mixed28 on registers only, two waves per SIMD (%s)"
                                                % UBENCH["source"]}}
    return hbm, issue


def ntt_roofline(nn, ntt_s, passes, traffic_key=None):
    achieved = NTT_BYTES_PER_UNIT * nn / ntt_s / 1e9
    traffic, traffic_file = profile_lookup("hbm_traffic.json", traffic_key) if traffic_key else (None, None)
    butterflies = (nn // 2) * (nn.bit_length() - 1)
    return {"valu": {"bound": "valu", "achieved": butterflies / ntt_s, "peak": BARE_BUTTERFLIES_PER_S, "unit": "butterflies/s",
                     "frac": butterflies / ntt_s / BARE_BUTTERFLIES_PER_S,
                     "note": "N/2 log2 N radix-2 butterflies over the kernel time, against fr29_butterfly on registers only (tools/ubench_fr29.hip: "
                             "%.0f clk per 64 butterflies per SIMD at four waves, %s); the passes also convert limbs, apply inter-pass twiddles and "
                             "pay ~10 us of ramp each (profiles/r02_ntt_radix4_ab.txt)" % (UBENCH["butterfly_clk"], UBENCH["source"])},
            "bound": "hbm", "kernel": "ntt_pass_* (all %d passes of one transform)" % passes, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
            "traffic_source": ("profiles/%s: %s" % (traffic_file, traffic.get("source"))) if traffic else None, "kernel_ms": ntt_s * 1e3,
            "algorithmic_bytes_per_launch": NTT_BYTES_PER_UNIT * nn}


def cpu_baseline(sample_log_n, threads_all):
    """reference-faithful CPU path (oracle restatement of src/msm.rs: c = 4, 64 windows, projective adds),
    single thread like the reference, on a bounded sample of the same synthetic workload"""
    import numpy as np
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
        "sample": "2^%d-point bucket_msm(b=256,c=4) restated from src/msm.rs, same synthetic inputs, %.1f s; the algorithm is "
                  "exactly linear in the point count (64 windows x (N bucket additions + 45 bucket-reduction operations)), so "
                  "the rate holds for 2^20 and 2^24 (2^20 would take %.0f s on one core)" % (sample_log_n, t1 - t0, (t1 - t0) * (1 << (20 - sample_log_n))),
        "all_cores": {"value": n / (t2 - t1), "cores": threads_all, "note": "same algorithm, 64 windows over OpenMP threads"},
        "ntt": {"value": (1 << ntt_log) / (t4 - t3), "unit": "elements/s", "cores": 1,
                "sample": "2^%d radix-2 NTT (output-identical O(n log n) twin of utils.rs:63-81), %.2f s" % (ntt_log, t4 - t3),
                "reference_quadratic_form": {"value": (1 << 8) / (t6 - t5), "unit": "elements/s",
                                             "sample": "utils.rs:63-81 as written (n^2 terms, one pow each) at n = 2^8, %.2f s; cost grows as n^2" % (t6 - t5)}},
        "host": "%d logical CPUs" % (os.cpu_count() or 0),
    }


def group_legs_child(job):
    """The two legs that drive ONE bp_init_multi context over all N GPUs (group_commit, one proof + one NTT over all GPUs), in a process
    of their own: started by rank 0 with a timeout, while the ranks wait on the host.  They are the only part of the bench whose
    GPU-to-GPU branches have not run on distinct GPUs yet, so a fault or a hang there must cost these two entries, not the line."""
    import hashlib
    import random
    import numpy as np
    import torch
    import baby_plonk_rust_amd as bp
    devs, world = job["devices"], len(job["devices"])
    torch.cuda.set_device(devs[0])
    dev = torch.device("cuda", devs[0])
    ctx = bp.Context(devs[0])
    out = {}
    # which pairs of the listed devices can address each other directly (hipDeviceCanAccessPeer): a missing link or a wrong cross-device
    # dependency must show up as a number in the line, not as a hang
    uniq = sorted(set(devs))
    peer = {"devices": devs, "can_access_peer": {"%d->%d" % (a, b): bool(torch.cuda.can_device_access_peer(a, b)) for a in uniq for b in uniq if a != b}}

    def synthetic(count, seed, first=0):
        t = torch.empty(count * 4, dtype=torch.int64, device=dev)
        ctx.synthetic_scalars_device(t.data_ptr(), count, (seed + GOLDEN * 8 * first) & MASK64)
        return t

    if job.get("commit"):
        c = job["commit"]
        try:
            total, k = c["total"], c["k"]
            gctx = bp.Context(devs)
            t0 = time.perf_counter()
            gsrs = gctx.srs_generate_progression(total, A0, D0)
            ginfo = gctx.srs_precompute(gsrs, bp.SRS_TABLES_OFF if job["no_tables"] else 0)
            g_setup = time.perf_counter() - t0
            hs = synthetic(total, 0x5EED0000 + c["log_n"], 0).cpu().numpy().view(np.uint64).reshape(total, 4)    # pageable, as a Vec<Scalar> is
            for _ in range(2):
                gres = gctx.msm(gsrs, hs)
            t0 = time.perf_counter()
            for _ in range(k):
                gres = gctx.msm(gsrs, hs)
            g_dt = time.perf_counter() - t0
            members = gctx.msm_member_stats()
            out["group_commit"] = {
                "metric": "g1_msm_scalar_muls_per_s", "value": total * k / g_dt, "unit": "scalar-muls/s", "n_gpus": world, "steps": k,
                "ms_per_step": 1e3 * g_dt / k, "scalars": "pageable host memory, %d MiB per step, one slice per member over its own PCIe link" % (32 * total >> 20),
                "window_bits": ginfo["window_bits"], "table_bytes_all_gpus": ginfo["bytes"], "srs_and_tables_s": g_setup,
                "per_member": members, "upload_ms_max": max(m["upload_ms"] for m in members), "device_ms_max": max(m["device_ms"] for m in members),
                "peer_access": peer,
                "same_result": gres.hex() == c["expect"], "result_sha": hashlib.sha256(gres).hexdigest()[:16],
                "how": "one bp_init_multi context in a process of its own beside the ranks (the drop-in for Setup::commit, setup.rs:32-37): SRS "
                       "sharded by point range, one persistent host thread per member, partial sums added on the host; no collective"}
            gctx.srs_free(gsrs)
            gctx.close()
            del hs
        except Exception as e:
            out["group_commit"] = {"n_gpus": world, "error": repr(e)[:300]}
    if job.get("prove"):
        pj = job["prove"]
        try:
            from baby_plonk_rust_amd.synthetic import Q as FR_Q, chained_multiplications
            pn = 1 << pj["log_n"]
            cols, pk = chained_multiplications(pn, 1000)                      # rank 0's circuit
            blinders = [random.Random(5).randrange(1, FR_Q) for _ in range(11)]
            gctx = bp.Context(devs)
            gsetup = bp.Setup.generate_srs(pn + 6, 0x1234567 + pj["log_n"], gctx, tables=not job["no_tables"])
            gprover = bp.Prover(gsetup, bp.Circuit(pk, gctx))
            gwit = [torch.from_numpy(c.view(np.int64)).to(dev) for c in cols]
            gp = [w.data_ptr() for w in gwit]
            gblob = gprover.prove_device(gp[0], gp[1], gp[2], None, blinders)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(pj["reps"]):
                gblob = gprover.prove_device(gp[0], gp[1], gp[2], None, blinders)
            g_elapsed = time.perf_counter() - t0
            out["group_proof"] = {"n_gpus": world, "latency_ms_per_proof": 1e3 * g_elapsed / pj["reps"], "round_ms": gprover.last_stats()["round_ms"],
                                  "peer_access": peer, "last_commit_per_member": gctx.msm_member_stats(),
                                  "same_proof_bytes_as_one_gpu": hashlib.sha256(gblob).hexdigest() == pj["expect_sha256"],
                                  "how": "one bp_init_multi context in a process of its own beside the ranks: SRS and the nine MSMs of a proof sharded by "
                                         "point range over the %d GPUs (peer copies of the scalar slices, one batched pipeline per member and round, "
                                         "partial sums added on the host), round 3's quotient split by coset over the first %d GPUs (a, b, c sent "
                                         "right after round 1), the other polynomial work on GPU 0" % (world, 4 if world >= 4 else 2)}
            # ONE host-to-host NTT through the same context (SURVEY 8e option ii): column slices over every GPU's PCIe link,
            # block exchange between the GPUs, outputs back; beside it the same call on a single-GPU context
            if pj.get("ntt_log_n", 0) >= 22 and world in (2, 4, 8):
                try:
                    hx = synthetic(1 << pj["ntt_log_n"], 0xF40000 + pj["ntt_log_n"]).cpu().numpy().view(np.uint64).reshape(-1, 4)
                    ms = {}
                    for name, c in (("one_gpu", ctx), ("all_gpus", gctx)):
                        best = None
                        for _ in range(3):
                            t0 = time.perf_counter()
                            hy = c.ntt(hx)
                            dt = time.perf_counter() - t0
                            best = dt if best is None or dt < best else best
                        ms[name] = {"ms_host_to_host": 1e3 * best, "kernel_ms": c.ntt_stats()["device_ms"], "members": c.ntt_stats()["members"],
                                    "sha": hashlib.sha256(hy.tobytes()).hexdigest()[:16]}
                    ms["same_output"] = ms["one_gpu"]["sha"] == ms["all_gpus"]["sha"]
                    ms["log_n"] = pj["ntt_log_n"]
                    out["group_proof"]["one_ntt_over_all_gpus"] = ms
                except Exception as e:
                    out["group_proof"]["one_ntt_over_all_gpus"] = {"error": repr(e)[:300]}
            gctx.close()
        except Exception as e:
            out["group_proof"] = {"n_gpus": world, "error": repr(e)[:300]}
    print("GROUP_LEGS " + json.dumps(out), flush=True)


def run_group_legs(job, timeout_s):
    """rank 0: the child of group_legs_child; whatever happens to it, an entry per requested leg comes back"""
    want = [k for k, leg in (("group_commit", "commit"), ("group_proof", "prove")) if job.get(leg)]
    try:
        cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--group-legs-child", json.dumps(job)], capture_output=True, text=True, timeout=timeout_s)
        for line in cp.stdout.splitlines():
            if line.startswith("GROUP_LEGS "):
                got = json.loads(line[len("GROUP_LEGS "):])
                return {k: got.get(k, {"n_gpus": len(job["devices"]), "error": "leg missing from the child's answer"}) for k in want}
        err = "child exited with %d: %s" % (cp.returncode, (cp.stderr or cp.stdout)[-300:])
    except subprocess.TimeoutExpired:
        err = "child killed after %d s" % timeout_s
    except Exception as e:
        err = repr(e)[:300]
    return {k: {"n_gpus": len(job["devices"]), "error": err} for k in want}


def main():
    args = parse_args()
    if args.group_legs_child:
        group_legs_child(json.loads(args.group_legs_child))
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    import hashlib

    import numpy as np
    import torch
    import torch.distributed as dist

    import baby_plonk_rust_amd as bp
    from baby_plonk_rust_amd import dist as bpd
    from baby_plonk_rust_amd.synthetic import Q as FR_Q

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and world > n_dev:
        raise SystemExit("bench.py: %d ranks but %d GPUs (one process per GPU)" % (world, n_dev))
    dev_index = local_rank % max(n_dev, 1)            # == local_rank except in a gloo rehearsal
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    ctl = None                                        # host-side group for object gathers and long waits
    use_dist = "WORLD_SIZE" in os.environ             # under a launcher even a single rank goes through the collectives
    if use_dist:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
            ctl = dist.new_group(backend="gloo")
        else:
            dist.init_process_group(args.backend)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(vals):
        if not use_dist:
            return [float(v) for v in vals]
        t = torch.tensor(vals, dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t]

    def gather_objects(obj):
        if not use_dist:
            return [obj]
        out = [None] * world
        dist.all_gather_object(out, obj, group=ctl)
        return out

    # ---------------------------------------------------------------- un-losable line + watchdog (N > 1 has never run on hardware)
    # Under a launcher rank 0 prints a COMPLETE line after the weak leg and after every later leg; a leg that hangs costs that leg, the
    # watchdog names it and ends the rank with a non-zero code (a fresh exit, never an exec), the launcher ends the others.
    progress = {"leg": "start-up", "t": time.monotonic()}
    if use_dist and args.leg_timeout > 0:
        import threading

        def watchdog():
            while True:
                time.sleep(5)
                if time.monotonic() - progress["t"] > args.leg_timeout:
                    sys.stderr.write("bench.py: rank %d spent more than %d s in the leg after '%s' -- giving up (the lines of the finished legs "
                                     "have been printed)\n" % (rank, args.leg_timeout, progress["leg"]))
                    sys.stderr.flush()
                    os._exit(3)

        threading.Thread(target=watchdog, daemon=True).start()

    ctx = bp.Context(dev_index)
    ctx.comm_set_timeout_ms(args.comm_timeout_ms)
    exchange = bpd.ShardedMsm(ctx)                     # under nccl: joins the library's communicator (every step agreed on by all ranks), or raises on all
    # every leg's result; None / empty until the leg has run (the line is built from whatever is there)
    pipelined = ntt = ntt_batch = ntt_cols = strong = group_commit = prove = None
    seams, sizes, red, legs_done, cols_all_box = {}, {}, {}, [], [None]

    def timed_msm(srs, d_scal, n, steps, warmup):
        """W + K steps of: per-rank Pippenger on the resident shard, then (N > 1) the single all-gather of the ranks'
        records left in HBM + one D2H + combine (baby_plonk_rust_amd/dist.py ShardedMsm)"""
        def step():
            if not use_dist:
                return bp.sum_partials(ctx.msm_partial(srs, None, device_ptr=d_scal.data_ptr(), n=n))
            return exchange(srs, device_ptr=d_scal.data_ptr(), n=n)
        for _ in range(warmup):
            res = step()
        barrier()
        acc, devt, exch = [], [], []
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step()
            st = ctx.msm_stats()
            acc.append(st["accumulate_ms"])
            devt.append(st["device_ms"])
            exch.append(exchange.exchange_s * 1e3)
        barrier()
        return {"result": res, "elapsed": time.perf_counter() - t0, "acc_ms": float(np.mean(acc)), "dev_ms": float(np.mean(devt)),
                "exchange_ms": float(np.mean(exch)) if use_dist else 0.0, "stats": ctx.msm_stats()}

    def timed_ntt(d_vec, log_n, steps, warmup):
        """K transforms of the HBM-resident vector enqueued back to back (bp_ntt_fr_device_async) and one wait: the step is the
        transform, not a host round trip; the blocking call (one hipStreamSynchronize per transform) is timed beside it.
        kernel_ms: HIP events around the passes of one transform of the TIMED region -- after the warm-up has built the pass tables
        (round 2 read them during the warm-up and reported a kernel time above the step time)."""
        for _ in range(max(warmup, 3)):
            ctx.ntt_device(d_vec.data_ptr(), log_n)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.ntt_device(d_vec.data_ptr(), log_n)
        blocking = time.perf_counter() - t0
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.ntt_device_async(d_vec.data_ptr(), log_n)
        ctx.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        # events around the passes of the LAST enqueued transform: its first event completes when the transform before it has
        # drained, so the interval is the transform's own GPU time inside the timed, back-to-back stream (a blocking call's events
        # also contain the host's launch gaps between the passes)
        dev_ms = float(ctx.ntt_stats()["device_ms"])
        # a kernel time cannot exceed the time of the step it is part of (events and wall clock differ by microseconds at most)
        consistent = dev_ms <= 1.03 * 1e3 * elapsed / steps + 0.003
        return {"elapsed": elapsed, "blocking_ms": 1e3 * blocking / steps, "dev_ms": dev_ms, "passes": ctx.ntt_stats()["passes"],
                "kernel_le_step": bool(consistent)}

    def fr_sums(t, count, first):
        """(sum_i s_i, sum_i (first + i) s_i) over the `count` raw 256-bit words of tensor t (int64 x 4 per element), as Python
        integers -- exact integer arithmetic on the GPU in 16-bit limbs, no modular reduction, no CPU oracle"""
        w = t.view(count, 4)
        idx = torch.arange(first, first + count, dtype=torch.int64, device=t.device)
        lo, hi = idx & 0xFFF, idx >> 12                                   # i = hi * 4096 + lo, both factors small enough for exact int64 sums
        s0 = s1 = 0
        for j in range(4):
            for part in range(4):
                limb = (w[:, j] >> (16 * part)) & 0xFFFF
                shift = 64 * j + 16 * part
                s0 += int(limb.sum().item()) << shift
                s1 += (int((limb * lo).sum().item()) + (int((limb * hi).sum().item()) << 12)) << shift
        return s0, s1

    R_INV = pow(1 << 256, -1, FR_Q)

    def expected_msm(sum_pairs, a0, d0):
        """96 bytes of (sum_i s_i (a0 + i d0)) G from the ranks' (sum s, sum i s) pairs (s in Montgomery form: one R^-1), through
        the library itself: a one-point MSM of G by that scalar"""
        s0 = sum(p[0] for p in sum_pairs)
        s1 = sum(p[1] for p in sum_pairs)
        k = (a0 * s0 + d0 * s1) % FR_Q * R_INV % FR_Q
        g = ctx.srs_generate_progression(1, 1, 0)
        out = ctx.msm(g, bp.scalars_from_ints([k]))
        ctx.srs_free(g)
        return out

    def synthetic(count, seed, first=0):
        """`count` scalars of the global stream `seed`, starting at element `first` (8 SplitMix64 words per element)"""
        t = torch.empty(count * 4, dtype=torch.int64, device=dev)
        ctx.synthetic_scalars_device(t.data_ptr(), count, (seed + GOLDEN * 8 * first) & MASK64)
        return t

    def checkpoint(leg, **times):
        """a leg has finished on this rank: max over ranks of its wall times (a collective -- every rank passes the same points in the same
        order), then, under a launcher, rank 0 prints a COMPLETE line from the legs finished so far (`provisional`: true)"""
        keys = sorted(times)
        if keys:
            red.update(zip(keys, max_over_ranks([times[k] for k in keys])))
        legs_done.append(leg)
        progress["leg"], progress["t"] = leg, time.monotonic()
        if use_dist:
            emit(final=False)
        if use_dist and args.rehearse_hang_after == leg and rank == world - 1 and world > 1:
            # rehearsal of a rank that stops taking part (tests/test_gpu_dist.py): the other ranks run into their next collective and stay there;
            # the watchdog of every rank names the leg and ends the rank, and the lines printed so far are what the caller keeps
            while True:
                time.sleep(3600)

    def emit(final):
        per_rank = gather_objects({"rank": rank, "device": dev_index, "weak_accumulate_ms": head["acc_ms"], "weak_device_ms": head["dev_ms"],
                                   "weak_exchange_ms": head["exchange_ms"],
                                   "strong_points": strong["m"] if strong else 0, "strong_accumulate_ms": strong["r"]["acc_ms"] if strong else 0.0,
                                   "strong_device_ms": strong["r"]["dev_ms"] if strong else 0.0,
                                   "strong_exchange_ms": strong["r"]["exchange_ms"] if strong else 0.0})
        progress["t"] = time.monotonic()
        if rank == 0:
            print(json.dumps(build_line(per_rank, final)), flush=True)

    def build_line(per_rank, final):
        elapsed, other_elapsed = red["elapsed"], red["other_elapsed"]
        ntt_elapsed, prove_elapsed, prove_single = red.get("ntt_elapsed", 0.0), red.get("prove_elapsed", 0.0), red.get("prove_single", 0.0)
        strong_elapsed, cols_elapsed, cols_gather = red.get("strong_elapsed", 0.0), red.get("cols_elapsed", 0.0), red.get("cols_gather", 0.0)
        cols_all = cols_all_box[0]
        stats = head["stats"]
        units = world * n * args.steps
        cfg_key = "2p%d_c%d%s" % (args.log_n, stats["window_bits"], "_tables" if stats["tables"] else "")
        hbm, issue = msm_roofline(n, head["acc_ms"] * 1e-3, stats["mixed_adds"], stats["window_bits"], stats["tables"], "msm_accumulate_" + cfg_key)
        line = {
            "metric": "g1_msm_scalar_muls_per_s",
            "baseline_metric": "G1 MSM scalar-muls/s + Fr NTT elements/s at 2^20/2^24; proof bit-exact",
            "value": units / elapsed, "unit": "scalar-muls/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 limbs (381-bit Fp Montgomery, 255-bit Fr)", "data": "synthetic",
            "config": {"workload": "2^%d-point BLS12-381 G1 MSM per GPU (global 2^%d x %d points, point-range shards, one RCCL all-gather "
                                   "of the ranks' partial-sum records), SRS %s; + 2^%d Fr NTT per GPU; BASELINE configs[2] at N=1"
                                   % (args.log_n, args.log_n, world,
                                      "raw points only" if args.no_tables else "resident with its fixed-base window tables (Setup)", args.ntt_log_n),
                       "msm_points_per_gpu": n, "window_bits": stats["window_bits"], "ntt_len_per_gpu": 1 << args.ntt_log_n,
                       "srs_tables": {"used": stats["tables"], "window_bits": table_info["window_bits"], "windows": table_info["windows"],
                                      "bytes": table_info["bytes"], "bytes_per_gpu": table_info["bytes"], "build_ms": 1e3 * table_build_s,
                                      "build_s": table_build_s,
                                      "note": "the headline is a FIXED-BASE MSM: T[w][i] = 2^(window_bits w) P_i built once per SRS outside the "
                                              "timed region; value_without_tables is the same MSM on the raw points"},
                       "value_without_tables": None if args.no_tables else units / other_elapsed,
                       "parallelism": "point-range x%d" % world, "ranks": world, "backend": args.backend if use_dist else None,
                       "msm_exchange": None if not use_dist else
                       ("bp_msm_g1_allgather: record -> ncclAllGather -> device pre-sum -> one D2H, all under the C ABI (capi_comm.hip); the only "
                        "path under nccl -- a communicator that cannot be created is an error on every rank; %d collectives enqueued by rank 0 so far, "
                        "waits bounded at %d ms" % (ctx.comm_stats()["collectives"], ctx.comm_stats()["timeout_ms"])
                        if exchange.c_path else "dist.ShardedMsm host-side gather (gloo rehearsal of the control flow)")},
            # the headline is a FIXED-BASE MSM (the SRS of a Setup with its window tables resident); the reference-shaped figures beside it:
            "value_without_tables": None if args.no_tables else units / other_elapsed,
            "ms_per_step_without_tables": None if args.no_tables else 1e3 * other_elapsed / args.steps,
            "first_commit_ms": None if args.no_tables else 1e3 * (table_build_s + elapsed / args.steps),
            "uncached_seam_ms_per_call": seams["msm_uncached_seam"]["ms_per_call"] if "msm_uncached_seam" in seams else None,
            "host_scalars_ms_per_step": seams["msm_host_scalars"]["ms_per_step"] if "msm_host_scalars" in seams else None,
            "headline_note": "value = scalars resident in HBM against an SRS whose fixed-base tables (config.srs_tables) were built once outside "
                             "the timed region; first_commit_ms = table build + one MSM; value_without_tables = the same MSM on raw points; "
                             "uncached_seam_ms_per_call = bucket_msm(&[G1Projective], &[Scalar]) literally, nothing cached, PCIe inclusive",
            "roofline": hbm,
            "roofline_valu_issue": issue,
            "msm_device_ms": head["dev_ms"],
            "msm_accumulate_ms": head["acc_ms"],
            "tail_ms": head["dev_ms"] - head["acc_ms"],
            "tail_note": "device time of one MSM outside msm_accumulate: bucket sort (3 launches), fix-up, bit-plane tree, copies",
            "register_only_rates_source": UBENCH["source"],
            "equals_closed_form": weak_ok,
            "exchange_ms": head["exchange_ms"],
            ("msm_with_tables" if args.no_tables else "msm_without_tables"): {
                "value": units / other_elapsed, "unit": "scalar-muls/s", "ms_per_step": 1e3 * other_elapsed / args.steps,
                "device_ms": other["dev_ms"], "accumulate_ms": other["acc_ms"], "tail_ms": other["dev_ms"] - other["acc_ms"],
                "window_bits": other["stats"]["window_bits"]},
            "ntt": None if not ntt else {"metric": "fr_ntt_elements_per_s", "value": world * nn * args.steps / ntt_elapsed, "unit": "elements/s",
                    "ms_per_step": 1e3 * ntt_elapsed / args.steps, "ms_per_blocking_call": ntt["blocking_ms"], "passes": ntt["passes"],
                    "kernel_le_step": ntt["kernel_le_step"], "spot_check": ntt_ok,
                    "how": "steps enqueued back to back on the context's stream (bp_ntt_fr_device_async), one wait at the end; "
                           "ms_per_blocking_call = the same transform through bp_ntt_fr_device, which waits for the stream every call",
                    "batched_columns": ntt_batch,
                    "roofline": ntt_roofline(nn, ntt["dev_ms"] * 1e-3, ntt["passes"], "ntt_2p%d" % args.ntt_log_n)},
            "result_sha": hashlib.sha256(head["result"]).hexdigest()[:16],
            "per_rank": per_rank,
            # under a launcher a complete line goes out after every leg: which legs this one holds, and whether more will follow
            "legs_finished": list(legs_done), "provisional": not final,
        }
        if ntt_cols:
            line["ntt_columns"] = {
                "metric": "fr_ntt_elements_per_s", "value": NTT_COLUMNS * nn * args.steps / cols_elapsed, "unit": "elements/s", "n_gpus": world,
                "columns": NTT_COLUMNS, "column_len": nn, "columns_per_rank": [c["columns"] for c in cols_all], "steps": args.steps,
                "ms_per_step": 1e3 * cols_elapsed / args.steps, "allgather_ms_per_step": 1e3 * cols_gather / args.steps,
                "allgather_bytes_per_rank_out": 32 * nn * ((NTT_COLUMNS + world - 1) // world) * world,
                "backend": args.backend, "collective": "bp_ntt_columns_allgather (in-place ncclAllGather under the C ABI)" if ntt_cols["c_path"]
                else "dist.all_gather_columns (host-side gather of the gloo rehearsal)",
                "foreign_column_matches_local_transform": all(c["ok"] for c in cols_all),
                "same_on_all_ranks": len({c["digest"] for c in cols_all}) == 1,
                "how": "column j of %d belongs to rank j mod N (dist.my_columns); per step every rank transforms its columns "
                       "(bp_ntt_fr_device_async, one wait) in its block of one [N x columns per rank, 2^%d, 4] buffer and ONE all-gather over RCCL "
                       "brings every column to every rank; allgather_ms = that collective alone, max over ranks" % (NTT_COLUMNS, args.ntt_log_n)}
        line.update(seams)
        if pipelined:
            line["pipelined"] = pipelined
        if sizes:
            line["other_sizes"] = sizes
        if strong:
            r, k = strong["r"], strong["k"]
            s_hbm, s_issue = msm_roofline(strong["m"], r["acc_ms"] * 1e-3, r["stats"]["mixed_adds"], r["stats"]["window_bits"], r["stats"]["tables"],
                                          "msm_accumulate_2p%d_c%d%s" % (args.strong_log_n, r["stats"]["window_bits"], "_tables" if r["stats"]["tables"] else "")
                                          if world == 1 else None)
            line["strong_scaling"] = {
                "metric": "g1_msm_scalar_muls_per_s", "value": strong["total"] * k / strong_elapsed, "unit": "scalar-muls/s", "scaling": "strong",
                "n_gpus": world, "rccl_ranks": world if (use_dist and args.backend == "nccl") else 0, "steps": k, "warmup": 2, "ms_per_step": 1e3 * strong_elapsed / k,
                "workload": "ONE 2^%d-point G1 MSM (BASELINE configs[3]), %d points per rank, SRS shards resident with tables; per step: "
                            "per-rank Pippenger, one all-gather of %d-byte records in HBM, one D2H, host combine"
                            % (args.strong_log_n, strong["m"], bp._lib.MSM_BLOB_BYTES),
                "points_per_rank": strong["m"], "window_bits": r["stats"]["window_bits"], "tables": r["stats"]["tables"],
                "accumulate_ms_per_rank": [p["strong_accumulate_ms"] for p in per_rank],
                "device_ms_per_rank": [p["strong_device_ms"] for p in per_rank],
                "tail_ms_per_rank": [p["strong_device_ms"] - p["strong_accumulate_ms"] for p in per_rank],
                "equals_closed_form": strong_ok,
                "exchange_ms_per_rank": [p["strong_exchange_ms"] for p in per_rank],
                "srs_generate_s": strong["gen_s"], "table_build_s": strong["table_s"], "table_bytes_per_gpu": strong["table_info"]["bytes"],
                "result_sha": hashlib.sha256(r["result"]).hexdigest()[:16],
                "roofline": s_hbm, "roofline_valu_issue": s_issue}
            if "ntt" in strong:
                t = strong["ntt"]
                line["strong_scaling"]["ntt"] = {"value": strong["total"] * k / t["elapsed"], "unit": "elements/s", "ms_per_step": 1e3 * t["elapsed"] / k,
                                                 "ms_per_blocking_call": t["blocking_ms"], "passes": t["passes"], "kernel_le_step": t["kernel_le_step"],
                                                 "roofline": ntt_roofline(strong["total"], t["dev_ms"] * 1e-3, t["passes"], "ntt_2p%d" % args.strong_log_n)}
        if group_commit:
            line["group_commit"] = group_commit
        if prove:
            line["prove"] = {"metric": "plonk_proofs_per_s", "value": world * prove["streams"] * args.prove_reps / prove_elapsed, "unit": "proofs/s",
                             "gates": 1 << args.prove_log_n, "concurrent_provers_per_gpu": prove["streams"],
                             "latency_ms_per_proof_single_prover": 1e3 * prove_single / args.prove_reps,
                             "latency_ms_per_proof_host_witness": None if prove["host_witness_s"] is None else 1e3 * prove["host_witness_s"],
                             "proofs_per_s_single_prover_per_gpu": args.prove_reps / prove_single,
                             "round_ms": prove["round_ms"], "proofs_timed_per_gpu": prove["streams"] * args.prove_reps,
                             "parallelism": "independent proofs x%d" % world,
                             "one_proof_over_all_gpus": prove["group"],
                             "workload": "bp_prove: prover.rs rounds 1-5 + host transcript on a synthetic 2^%d-gate circuit (chained "
                                         "multiplications), witness and circuit resident in HBM, 624-byte proof out; BASELINE configs[4]"
                                         % args.prove_log_n,
                             "proof_sha_rank0": prove["sha"], "srs_and_circuit_setup_s": prove["setup_s"],
                             "synthetic_circuit_host_s": prove["circuit_host_s"]}
        if final and world == 1 and not args.skip_cpu:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample_log_n, min(64, os.cpu_count() or 1))
        return line

    # ---------------------------------------------------------------- weak-scaling MSM (the headline `value`)
    n = 1 << args.log_n
    srs = ctx.srs_generate_progression(n, A0 + rank * n * D0, D0)          # this rank's point range [rank n, (rank + 1) n)
    scal = synthetic(n, 0x5EED0000 + args.log_n, rank * n)
    table_info, table_build_s = {"window_bits": 0, "windows": 0, "bytes": 0}, 0.0
    if args.no_tables:                  # secondary leg first: the other table setting, same inputs
        t0 = time.perf_counter()
        table_info = ctx.srs_precompute(srs, 0)
        table_build_s = time.perf_counter() - t0
        other = timed_msm(srs, scal, n, args.steps, args.warmup)
        ctx.srs_precompute(srs, bp.SRS_TABLES_OFF)
    else:
        other = timed_msm(srs, scal, n, args.steps, args.warmup)
        t0 = time.perf_counter()
        table_info = ctx.srs_precompute(srs, 0)
        table_build_s = time.perf_counter() - t0
    head = timed_msm(srs, scal, n, args.steps, args.warmup)
    assert other["result"] == head["result"], "MSM with and without fixed-base tables disagree"
    assert head["stats"]["tables"] == (not args.no_tables)
    # the timed result against the closed form: points are (A0 + i D0) G, so the MSM is (sum_i s_i (A0 + i D0)) G -- one dot
    # product mod q, each rank's share computed where its scalars are (a wrong sum must not print a scaling number)
    weak_pairs = gather_objects(fr_sums(scal, n, rank * n))
    weak_ok = expected_msm(weak_pairs, A0, D0) == head["result"]
    assert weak_ok, "weak-scaling MSM differs from the closed form"
    checkpoint("weak_msm", elapsed=head["elapsed"], other_elapsed=other["elapsed"])

    # ---------------------------------------------------------------- pipelined: three commitments in flight (what a prover round issues)
    # `value` above is SERIAL: one MSM, its result on the host, then the next (what one Setup::commit call costs).  The prover's rounds 1, 3
    # and 5 issue three / three / two commitments whose results are needed only together (prover.rs:249-251, 483-485, 640-641): bp_commit_many
    # runs them on the context's lanes, so that one pipeline's sort and tail overlap another's accumulation.  Per step: three 2^log_n-scalar
    # vectors against the same resident SRS + tables, all three results on the host; every result checked against its closed form.
    if world == 1 and not args.no_tables and not args.skip_pipelined:
        setup = bp.Setup(srs, ctx, tables=False)                       # wraps the resident handle (its tables exist already)
        vecs = [scal] + [synthetic(n, 0x9199000 + 131 * j) for j in (1, 2)]
        polys = [bp.DevicePolynomial(v.view(n, 4), bp.BASIS_MONOMIAL, ctx) for v in vecs]
        torch.cuda.synchronize()
        for _ in range(max(1, args.warmup)):
            got3 = bp.commit_many_device(setup, polys)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            got3 = bp.commit_many_device(setup, polys)
        barrier()
        dt = time.perf_counter() - t0
        pipe_ok = all(expected_msm([fr_sums(v, n, 0)], A0, D0) == g for v, g in zip(vecs, got3)) and got3[0] == head["result"]
        assert pipe_ok, "pipelined commitments differ from the closed form"
        pipelined = {"metric": "g1_msm_scalar_muls_per_s", "value": 3 * n * args.steps / dt, "unit": "scalar-muls/s", "msms_in_flight": 3,
                     "ms_per_msm": 1e3 * dt / (3 * args.steps), "ms_per_three": 1e3 * dt / args.steps, "equals_closed_form": pipe_ok,
                     "how": "bp_commit_many_device: three independent 2^%d-point MSMs against the same SRS + tables on three lanes (own stream and "
                            "workspaces each), all three results on the host per step; the serial `value` waits for every result before the next "
                            "MSM starts.  Cutting msm_accumulate into workgroup generations with the tails on high-priority streams was built and "
                            "measured in round 5 and loses (profiles/r05_accumulate_generations_ab.txt)" % args.log_n}
        del polys, vecs, setup

    # ---------------------------------------------------------------- seams of the reference (N = 1): host scalars, uncached points
    if world == 1 and not args.skip_seams:
        host_scal = scal.cpu().numpy().view(np.uint64).reshape(n, 4)       # pageable memory, as a Rust Vec<Scalar> is
        for _ in range(args.warmup):
            r = ctx.msm(srs, host_scal)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r = ctx.msm(srs, host_scal)
        dt = time.perf_counter() - t0
        assert r == head["result"]
        seams["msm_host_scalars"] = {
            "value": n * args.steps / dt, "unit": "scalar-muls/s", "ms_per_step": 1e3 * dt / args.steps, "device_ms": ctx.msm_stats()["device_ms"],
            "tail_ms": ctx.msm_stats()["device_ms"] - ctx.msm_stats()["accumulate_ms"],
            "seam": "Setup::commit(&Polynomial) (setup.rs:32-37): %d MiB of pageable host scalars cross PCIe per call, SRS and tables resident"
                    % (32 * n >> 20)}
        images = ctx.srs_export_projective144(srs)                         # what Setup.powers_of_x: Vec<G1Projective> holds (g1.rs:442-446)
        reps = max(2, min(args.steps, 5))

        def three_calls():
            h = ctx.srs_load_projective144(images)
            try:
                return ctx.msm(h, host_scal)
            finally:
                ctx.srs_free(h)

        timing = {}
        for name, fn in (("one_call", lambda: ctx.msm_projective144(images, host_scal)), ("three_calls", three_calls)):
            best, tot = None, 0.0
            for i in range(reps + 1):
                t0 = time.perf_counter()
                r = fn()
                dt = time.perf_counter() - t0
                if i:                                                      # first pass warms the staging workspaces
                    tot += dt
                    best = dt if best is None or dt < best else best
            assert r == head["result"]
            timing[name] = (tot / reps, best)
        seams["msm_uncached_seam"] = {
            "value": n / timing["one_call"][0], "unit": "scalar-muls/s", "ms_per_call": 1e3 * timing["one_call"][0], "best_ms": 1e3 * timing["one_call"][1],
            "three_calls_ms": 1e3 * timing["three_calls"][0],
            "seam": "BucketMSM::bucket_msm(points: &[G1Projective], scalars, 256, 4) (msm.rs:76-81) taken literally, nothing cached: %d MiB of "
                    "projective points + %d MiB of scalars from pageable memory in two pieces, each normalised on the GPU and multiplied (no "
                    "tables), the first while the second is uploaded (bp_msm_g1_projective144); three_calls_ms: bp_srs_load_projective144 + bp_msm_g1 + "
                    "bp_srs_free in sequence" % (144 * n >> 20, 32 * n >> 20)}
        del images, host_scal

    # ---------------------------------------------------------------- NTT leg (independent columns, no collective)
    nn = 1 << args.ntt_log_n
    vec = synthetic(nn, 0xF40000 + args.ntt_log_n, rank * nn)
    def ntt_spot_check(v, log_n):
        """output 0 of a forward transform is the plain sum of the inputs; the inverse brings the vector back"""
        c = v.clone()
        torch.cuda.synchronize()                                # the copy runs on torch's stream, the transform on the library's
        ctx.ntt_device(c.data_ptr(), log_n)
        torch.cuda.synchronize()
        out0 = sum((int(x) & MASK64) << (64 * j) for j, x in enumerate(c[:4].tolist()))
        ok = out0 == fr_sums(v, 1 << log_n, 0)[0] % FR_Q
        ctx.ntt_device(c.data_ptr(), log_n, inverse=True)
        torch.cuda.synchronize()
        return bool(ok and torch.equal(c, v))

    ntt_ok = ntt_spot_check(vec, args.ntt_log_n)
    assert ntt_ok, "NTT spot check failed"
    ntt = timed_ntt(vec, args.ntt_log_n, args.steps, args.warmup)
    checkpoint("ntt", ntt_elapsed=ntt["elapsed"])
    # the same transform as a BATCH of independent columns in one call (grid.y = column): what the prover issues (a, b, c, PI in round 1; the
    # coset evaluations of round 3) and what the north star's "NTT by independent columns" shards -- the per-pass ramp is paid once per batch
    if world == 1 and nn * 8 * 32 <= (8 << 30):
        NB = 8
        cols8 = torch.empty((NB, nn, 4), dtype=torch.int64, device=dev)
        for j in range(NB):
            ctx.synthetic_scalars_device(cols8[j].data_ptr(), nn, 0xBA7C0000 + 31 * j)
        first = cols8[3].clone()
        torch.cuda.synchronize()
        ctx.ntt_device(first.data_ptr(), args.ntt_log_n)                       # column 3 alone = column 3 of the batch
        ctx.ntt_device(cols8.data_ptr(), args.ntt_log_n, batch=NB)
        torch.cuda.synchronize()
        batch_ok = bool(torch.equal(cols8[3], first))
        assert batch_ok, "batched NTT differs from the single transform"
        for _ in range(max(1, args.warmup)):
            ctx.ntt_device_async(cols8.data_ptr(), args.ntt_log_n, batch=NB)
        ctx.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.ntt_device_async(cols8.data_ptr(), args.ntt_log_n, batch=NB)
        ctx.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        ntt_batch = {"columns": NB, "value": NB * nn * args.steps / dt, "unit": "elements/s", "ms_per_column": 1e3 * dt / (NB * args.steps),
                     "kernel_ms_per_batch": float(ctx.ntt_stats()["device_ms"]), "column_matches_single_transform": batch_ok,
                     "how": "bp_ntt_fr_device_async(batch = %d): eight independent 2^%d-element columns per launch" % (NB, args.ntt_log_n)}
        del cols8, first

    # ---------------------------------------------------------------- NTT columns over the ranks + ONE all-gather (north_star: "NTT by independent
    # columns across the GPUs with a single RCCL all-gather"; callers: the 23 transforms of prover.rs:386-450, utils.rs:106-129).  Column j
    # of NTT_COLUMNS belongs to rank j mod world (dist.my_columns); a step = every rank transforms its columns (enqueued back to back),
    # then one all_gather_columns brings every finished column to every rank.  Runs whenever a process group exists (a world of one included).
    if use_dist:
        mine_j = bpd.my_columns(NTT_COLUMNS, rank, world)
        per_rank_cols = (NTT_COLUMNS + world - 1) // world
        # all columns of all ranks in ONE buffer, rank r's block at rows [r * per_rank, (r + 1) * per_rank): this rank transforms its own
        # rows in place; under RCCL the gather is the LIBRARY's in-place ncclAllGather (bp_ntt_columns_allgather, the communicator lives
        # in the bp_ctx), under gloo (CPU rehearsal of the control flow) the Python mirror's host-side gather
        cols_c_path = args.backend == "nccl" and ctx.comm_info()[1] == world
        big = torch.zeros((world * per_rank_cols, nn, 4), dtype=torch.int64, device=dev)

        def row_of(j):
            return (j % world) * per_rank_cols + j // world

        torch.cuda.synchronize()                                # the zero fill runs on torch's stream, the library writes on its own
        for j in mine_j:                                        # generated in place (a temporary per column would be recycled by torch's
            ctx.synthetic_scalars_device(big[row_of(j)].data_ptr(), nn, 0xC0100000 + 977 * j)      # allocator under a copy still in flight)
        ctx.synchronize()

        def cols_step():
            for j in mine_j:
                ctx.ntt_device_async(big[row_of(j)].data_ptr(), args.ntt_log_n)
            ctx.synchronize()
            t_t = time.perf_counter()
            if cols_c_path:
                ctx.ntt_columns_allgather(big.data_ptr(), args.ntt_log_n, per_rank_cols)
                got = [big[row_of(j)] for j in range(NTT_COLUMNS)]
            else:
                got = bpd.all_gather_columns({j: big[row_of(j)] for j in mine_j}, NTT_COLUMNS)
            torch.cuda.synchronize()
            return got, time.perf_counter() - t_t

        got, _ = cols_step()                                    # first pass on fresh data: checked below
        probe = (rank + 1) % NTT_COLUMNS                        # a column another rank owns (its own when world = 1): transform it here and compare
        ref = synthetic(nn, 0xC0100000 + 977 * probe)
        ctx.ntt_device(ref.data_ptr(), args.ntt_log_n)
        torch.cuda.synchronize()
        cols_ok = bool(torch.equal(got[probe].to(dev).reshape(-1), ref.reshape(-1))) and len(got) == NTT_COLUMNS
        digest = hashlib.sha256()
        for c in got:
            digest.update(c[:64].cpu().numpy().tobytes())
        cols_digest = digest.hexdigest()[:16]
        del got, ref
        for _ in range(max(0, args.warmup - 1)):
            cols_step()
        barrier()
        t0 = time.perf_counter()
        gather_s = 0.0
        for _ in range(args.steps):
            _, g = cols_step()
            gather_s += g
        barrier()
        ntt_cols = {"elapsed": time.perf_counter() - t0, "gather_s": gather_s, "ok": cols_ok, "digest": cols_digest, "mine": len(mine_j),
                    "c_path": cols_c_path}
        del big
        torch.cuda.empty_cache()
        cols_all_box[0] = gather_objects({"ok": ntt_cols["ok"], "digest": ntt_cols["digest"], "columns": ntt_cols["mine"]})
        checkpoint("ntt_columns", cols_elapsed=ntt_cols["elapsed"], cols_gather=ntt_cols["gather_s"])

    # ---------------------------------------------------------------- the metric's other sizes, N = 1 (2^16 = configs[1]; 2^24 rides on the strong leg)
    def release():
        nonlocal srs, scal, vec
        if srs is not None:
            ctx.srs_free(srs)
        srs = scal = vec = None
        torch.cuda.empty_cache()

    if world == 1:
        for lg in args.other_sizes:
            if lg == args.log_n or lg == args.strong_log_n or lg < 10 or lg > 26:
                continue
            m = 1 << lg
            release()
            srs = ctx.srs_generate_progression(m, A0, D0)
            if not args.no_tables:
                ctx.srs_precompute(srs, 0)
            scal = synthetic(m, 0x5EED0000 + lg)
            vec = synthetic(m, 0xF40000 + lg)
            k = args.steps if lg <= 22 else max(2, min(args.steps, 5))
            r = timed_msm(srs, scal, m, k, args.warmup)
            size_ok = expected_msm([fr_sums(scal, m, 0)], A0, D0) == r["result"]
            assert size_ok, "2^%d MSM differs from the closed form" % lg
            size_ntt_ok = ntt_spot_check(vec, lg)
            assert size_ntt_ok, "2^%d NTT spot check failed" % lg
            t = timed_ntt(vec, lg, k, args.warmup)
            hbm, issue = msm_roofline(m, r["acc_ms"] * 1e-3, r["stats"]["mixed_adds"], r["stats"]["window_bits"], r["stats"]["tables"])
            sizes["2^%d" % lg] = {"steps": k, "msm": {"value": m * k / r["elapsed"], "unit": "scalar-muls/s", "ms_per_step": 1e3 * r["elapsed"] / k,
                                                        "device_ms": r["dev_ms"], "accumulate_ms": r["acc_ms"], "tail_ms": r["dev_ms"] - r["acc_ms"],
                                                        "window_bits": r["stats"]["window_bits"], "tables": r["stats"]["tables"],
                                                        "equals_closed_form": size_ok, "roofline": hbm, "roofline_valu_issue": issue},
                                  "ntt": {"value": m * k / t["elapsed"], "unit": "elements/s", "ms_per_step": 1e3 * t["elapsed"] / k, "ms_per_blocking_call": t["blocking_ms"],
                                          "passes": t["passes"], "kernel_le_step": t["kernel_le_step"], "spot_check": size_ntt_ok,
                                          "roofline": ntt_roofline(m, t["dev_ms"] * 1e-3, t["passes"])}}

    # ---------------------------------------------------------------- strong scaling = BASELINE configs[3]: ONE 2^24-point MSM over all ranks
    if args.strong_log_n:
        release()
        total = 1 << args.strong_log_n
        lo, hi = bpd.shard_range(total, rank, world)
        m = hi - lo
        t0 = time.perf_counter()
        srs = ctx.srs_generate_progression(m, A0 + lo * D0, D0)
        t_gen = time.perf_counter() - t0
        t0 = time.perf_counter()
        s_info = ctx.srs_precompute(srs, bp.SRS_TABLES_OFF if args.no_tables else 0)
        t_tab = time.perf_counter() - t0
        scal = synthetic(m, 0x5EED0000 + args.strong_log_n, lo)
        k = args.strong_steps or max(2, min(args.steps, 10))
        r = timed_msm(srs, scal, m, k, 2)
        strong_pairs = gather_objects(fr_sums(scal, m, lo))
        strong_ok = expected_msm(strong_pairs, A0, D0) == r["result"]
        assert strong_ok, "strong-scaling MSM differs from the closed form"
        strong = {"r": r, "k": k, "m": m, "total": total, "table_info": s_info, "gen_s": t_gen, "table_s": t_tab}
        if world == 1 and args.strong_log_n <= 26:
            vec = synthetic(total, 0xF40000 + args.strong_log_n)
            strong["ntt"] = timed_ntt(vec, args.strong_log_n, k, 2)
        checkpoint("strong_msm", strong_elapsed=strong["r"]["elapsed"])

    # ---------------------------------------------------------------- the drop-in's multi-GPU seam (N > 1): ONE commit through bp_init_multi
    # BASELINE configs[3] as the reference's single-threaded caller sees it (setup.rs:32-37, msm.rs:76-81): rank 0's process holds
    # one context over all N GPUs; one 2^strong_log_n-point MSM from PAGEABLE host scalars per step -- every member's slice crosses
    # its own PCIe link on its own host thread, every member runs its pipeline, the partial sums are added on the host.  The
    # other ranks wait on the host (their GPUs are driven by rank 0 here); same points and scalars as the strong leg above.
    def host_barrier():
        if use_dist:
            dist.barrier(group=ctl) if ctl is not None else dist.barrier()

    if world > 1 and strong is not None and not args.skip_group_legs:
        torch.cuda.synchronize()
        host_barrier()
        if rank == 0:
            group_devs = list(range(world)) if args.backend == "nccl" else [dev_index] * world      # a gloo rehearsal lists its one card once per rank
            group_commit = run_group_legs({"devices": group_devs, "no_tables": bool(args.no_tables),
                                           "commit": {"total": strong["total"], "k": strong["k"], "log_n": args.strong_log_n,
                                                      "expect": strong["r"]["result"].hex()}}, args.group_legs_timeout)["group_commit"]
        host_barrier()
        checkpoint("group_commit")

    # ---------------------------------------------------------------- prover leg (BASELINE configs[4])
    if args.prove_log_n:
        import random
        import threading
        from baby_plonk_rust_amd.synthetic import Q as FR_Q, chained_multiplications
        pn = 1 << args.prove_log_n
        release()
        t0 = time.perf_counter()
        cols, pk = chained_multiplications(pn, 1000 + rank)
        t_circuit_host = time.perf_counter() - t0
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
        barrier()                                                # latency: one prover, proofs back to back
        t0 = time.perf_counter()
        for _ in range(args.prove_reps):
            blob = provers[0].prove_device(ptrs[0], ptrs[1], ptrs[2], None, blinders)
        barrier()
        single_elapsed = time.perf_counter() - t0
        round_ms = provers[0].last_stats()["round_ms"]
        # the same proof with the witness in pageable host memory -- what the Rust caller's Vec<Scalar>s are:
        # round 1 uploads the columns itself, b and c beside the commitment to a
        host_witness_s = None
        if rank == 0:
            assert provers[0].prove_with_blinding(cols[0], cols[1], cols[2], None, blinders) == blob
            t0 = time.perf_counter()
            for _ in range(args.prove_reps):
                provers[0].prove_with_blinding(cols[0], cols[1], cols[2], None, blinders)
            host_witness_s = (time.perf_counter() - t0) / args.prove_reps

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
        prove_res = {"elapsed": time.perf_counter() - t0, "single_elapsed": single_elapsed, "round_ms": round_ms, "sha": hashlib.sha256(blob).hexdigest()[:16],
                 "setup_s": t_setup, "circuit_host_s": t_circuit_host, "streams": len(provers), "group": None, "host_witness_s": host_witness_s}
        prove = prove_res
        checkpoint("prove", prove_elapsed=prove["elapsed"], prove_single=prove["single_elapsed"])
        # one proof on ONE context over all N GPUs (bp_init_multi): the nine commitments of prover.rs are sharded by point range,
        # everything else runs on GPU 0.  Rank 0 drives it; the other ranks have freed their memory and wait on the host.
        if world > 1 and not args.skip_group_legs:
            del provers, wit
            torch.cuda.empty_cache()
            dist.barrier(group=ctl)
            if rank == 0:
                group_devs = list(range(world)) if args.backend == "nccl" else [dev_index] * world
                prove["group"] = run_group_legs({"devices": group_devs, "no_tables": bool(args.no_tables),
                                                 "prove": {"log_n": args.prove_log_n, "reps": args.prove_reps, "ntt_log_n": args.strong_log_n,
                                                           "expect_sha256": hashlib.sha256(blob).hexdigest()}}, args.group_legs_timeout)["group_proof"]
            dist.barrier(group=ctl)
            checkpoint("one_proof_over_all_gpus")

    emit(final=True)
    if use_dist:
        dist.barrier(group=ctl) if ctl is not None else dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

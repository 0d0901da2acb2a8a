#!/usr/bin/env python3
"""one-screen summary of a bench.py JSON line (file argument or stdin)"""
import json
import sys

d = json.loads((open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).read().strip().splitlines()[-1])
print("MSM 2^20: %.4g/s  step %.3f ms  device %.3f  accumulate %.3f  tail %.3f  host %.3f  closed_form %s  valu %.3f hbm %.5f" % (
    d["value"], d["ms_per_step"], d["msm_device_ms"], d["msm_accumulate_ms"], d["tail_ms"], d["ms_per_step"] - d["msm_device_ms"] - d.get("exchange_ms", 0),
    d["equals_closed_form"], (d["roofline_valu_issue"] or {}).get("frac", 0), d["roofline"]["frac"]))
o = d.get("msm_without_tables") or d.get("msm_with_tables")
print("  other table setting: step %.3f device %.3f tail %.3f" % (o["ms_per_step"], o["device_ms"], o["tail_ms"]))
n = d["ntt"]
print("NTT 2^20: %.4g/s step %.4f ms blocking %.4f kernel %.4f le_step %s valu %.3f" % (n["value"], n["ms_per_step"], n["ms_per_blocking_call"], n["roofline"]["kernel_ms"], n["kernel_le_step"], n["roofline"]["valu"]["frac"]))
for k, v in (d.get("other_sizes") or {}).items():
    m, t = v["msm"], v["ntt"]
    print("%s: MSM step %.3f device %.3f acc %.3f tail %.3f c=%d | NTT step %.4f kernel %.4f le_step %s" % (
        k, m["ms_per_step"], m["device_ms"], m["accumulate_ms"], m["tail_ms"], m["window_bits"], t["ms_per_step"], t["roofline"]["kernel_ms"], t["kernel_le_step"]))
s = d.get("strong_scaling")
if s:
    print("strong 2^24: step %.2f ms device %s acc %s c=%d ok %s" % (s["ms_per_step"], s["device_ms_per_rank"], s["accumulate_ms_per_rank"], s["window_bits"], s["equals_closed_form"]), end="")
    if "ntt" in s:
        print(" | NTT step %.3f kernel %.3f le_step %s" % (s["ntt"]["ms_per_step"], s["ntt"]["roofline"]["kernel_ms"], s["ntt"]["kernel_le_step"]), end="")
    print()
for k in ("msm_host_scalars", "msm_uncached_seam"):
    if k in d:
        print(k, {x: (round(y, 3) if isinstance(y, float) else y) for x, y in d[k].items() if x != "seam"})
if "group_commit" in d:
    print("group_commit", d["group_commit"])
p = d.get("prove")
if p:
    print("prove: %.2f ms single (%s ms from a host witness), %.2f proofs/s with %d provers, rounds %s" % (
        p["latency_ms_per_proof_single_prover"], "%.2f" % p["latency_ms_per_proof_host_witness"] if p.get("latency_ms_per_proof_host_witness") else "-",
        p["value"], p["concurrent_provers_per_gpu"], ["%.2f" % r for r in p["round_ms"]]))
    if p.get("one_proof_over_all_gpus"):
        print("  group:", p["one_proof_over_all_gpus"])
if "cpu_baseline" in d:
    print("cpu: %.4g/s on %d core" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))

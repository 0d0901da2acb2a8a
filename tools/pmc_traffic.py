#!/usr/bin/env python3
"""HBM bytes per launch from rocprofv3 PMC passes -> profiles/r04_hbm_traffic.json (what bench.py's roofline.traffic reads).

  python tools/pmc_traffic.py FETCH.csv WRITE.csv [--tag 2p20] [--out profiles/r04_hbm_traffic.json] [--merge]

FETCH.csv / WRITE.csv are the *_counter_collection.csv files of two separate runs of
  rocprofv3 --pmc FETCH_SIZE  -- python3 tools/run_msm.py --log-n L --reps 2 --tables 0 --ntt-log-n L
  rocprofv3 --pmc WRITE_SIZE  -- (same)
Counter values are KB per dispatch.  Corrections (MI355X_MICROARCH.md, HBM section, calibrated in round 1 on known byte counts):
FETCH_SIZE tallies a 128-B coalesced read request at 64 B, so kernels that stream 16 B per lane in >= 128-B runs are doubled;
64-B runs (the strided NTT pass of 2^19: 2^10 x 2 columns) are counted exactly; WRITE_SIZE is exact.  The gathers of msm_accumulate
(7 x 16 B per lane from random 112-B points in 128-B slots) are neither of the guide's calibrated cases; round 6 calibrated them on a known
byte count (tools/ubench_gather.hip, profiles/r06_fetch_gather_calibration.txt): FETCH_SIZE shows 0.514 of the 128-byte lines such a gather
pulls, so the counter is multiplied by GATHER_FETCH_FACTOR = 1.944 (read from that file; rounds 1-5 reported the raw counter as a lower
bound with a 1x..2x range)."""
import argparse
import collections
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gather_fetch_factor():
    """the measured factor of the 7 x dwordx4 gather (profiles/rNN_fetch_gather_calibration.txt, newest round), and the file it came from"""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fetch_gather_calibration.txt")), reverse=True):
        m = re.search(r"^gather<7>.*multiply the counter by ([0-9.]+)", open(path).read(), flags=re.M)
        if m:
            return float(m.group(1)), os.path.basename(path)
    raise SystemExit("no profiles/rNN_fetch_gather_calibration.txt: run tools/ubench_gather under rocprofv3 --pmc FETCH_SIZE first")


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").replace("bp::", "").replace("_swz", "")   # swizzled tile variants count as their pass
                acc[name].append((float(row["Counter_Value"]), int(row["Grid_Size"])))
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch")
    ap.add_argument("write")
    ap.add_argument("--tag", required=True, help="size tag of the run, e.g. 2p20")
    ap.add_argument("--window-bits", type=int, default=20)
    ap.add_argument("--tables", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_hbm_traffic.json"))
    ap.add_argument("--merge", action="store_true")
    args = ap.parse_args()
    fetch, write = per_kernel(args.fetch, "FETCH_SIZE"), per_kernel(args.write, "WRITE_SIZE")
    log_n = int(args.tag.split("p")[1])
    out = {}
    if args.merge and os.path.exists(args.out):
        out = json.load(open(args.out))
    out["_how"] = __doc__.split("\n\n")[1] if "\n\n" in __doc__ else ""
    src = "%s + %s" % (os.path.basename(args.fetch), os.path.basename(args.write))

    def mean_big(name, table):
        """mean over the launches of the timed size (largest grid)"""
        rows = table.get(name, [])
        if not rows:
            return None
        g = max(r[1] for r in rows)
        vals = [r[0] for r in rows if r[1] == g]
        return sum(vals) / len(vals)

    f, w = mean_big("msm_accumulate<2>", fetch), mean_big("msm_accumulate<2>", write)
    if f is not None and w is not None:
        factor, cal = gather_fetch_factor()
        out["msm_accumulate_%s_c%d%s" % (args.tag, args.window_bits, "_tables" if args.tables else "")] = {
            "fetch_kb_raw": f, "fetch_factor": factor, "fetch_factor_source": "profiles/" + cal, "write_kb": w,
            "hbm_bytes_per_launch": int((f * factor + w) * 1024),
            "algorithmic_bytes_per_launch": 128 << log_n, "source": src,
            "note": "one 112-B point (in its 128-B slot, one cache line) gathered per (scalar, window): FETCH_SIZE x %.3f (the factor measured for "
                    "exactly this access shape on a known byte count) + WRITE_SIZE; integer-issue bound kernel" % factor}
    passes, total = {}, 0.0
    for name in ("ntt_pass_strided", "ntt_pass_last", "ntt_small"):
        f, w = mean_big(name, fetch), mean_big(name, write)
        if f is None or w is None:
            continue
        n_str = len([1 for v, g in fetch[name] if g == max(r[1] for r in fetch[name])])
        factor = 1 if (name == "ntt_pass_strided" and log_n == 19) else 2        # 2^19 = 10 + 9: 2^10 x 2-column tiles read 64-B runs
        per = (f * factor + w) * 1024
        # a three-pass transform launches the strided kernel twice per transform
        mult = 2 if (name == "ntt_pass_strided" and log_n >= 20) else 1
        passes[name] = {"fetch_kb_raw": f, "fetch_factor": factor, "write_kb": w, "hbm_bytes_per_launch": int(per), "launches_per_transform": mult}
        total += per * mult
    if passes:
        out["ntt_%s" % args.tag] = {"passes": passes, "hbm_bytes_per_launch": int(total), "algorithmic_bytes_per_launch": 64 << log_n,
                                    "bytes_per_element": total / (1 << log_n), "source": src,
                                    "note": "hbm_bytes_per_launch = all passes of ONE transform"}
    with open(args.out, "w") as fo:
        json.dump(out, fo, indent=1)
    print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main()

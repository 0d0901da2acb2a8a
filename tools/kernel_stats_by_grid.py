#!/usr/bin/env python3
"""Per-(kernel, grid size) durations from a rocprofv3 --kernel-trace CSV.  rocprofv3's own --stats table averages every launch of a
kernel name, which mixes the 2^20 and 2^24 legs of one bench.py run; this keeps them apart.
  python tools/kernel_stats_by_grid.py gpurun_out/x/b_kernel_trace.csv > profiles/r02_..._by_grid.csv"""
import collections
import csv
import re
import sys

acc = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("bp::", "")
        grid = "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
        acc[(name, grid, r["Workgroup_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid_threads", "workgroup", "calls", "avg_us", "min_us", "max_us", "total_ms"])
for (name, grid, wg), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([name, grid, wg, len(v), "%.2f" % (sum(v) / len(v) / 1e3), "%.2f" % (min(v) / 1e3), "%.2f" % (max(v) / 1e3), "%.3f" % (sum(v) / 1e6)])

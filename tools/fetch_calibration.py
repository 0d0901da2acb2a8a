#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE per kernel of a counters-only rocprofv3 pass over tools/ubench_gather against the bytes the program
moved by construction -> the factor FETCH_SIZE under-/over-counts each access shape by (profiles/r06_fetch_gather_calibration.txt).
  python tools/fetch_calibration.py COUNTER_COLLECTION.csv UBENCH_STDOUT.txt"""
import collections
import csv
import re
import sys

rows = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if r["Counter_Name"] == "FETCH_SIZE":
            rows[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")].append(float(r["Counter_Value"]))
expect = int(re.search(r"EXPECT_BYTES (\d+)", open(sys.argv[2]).read()).group(1))
print("bytes moved by construction per launch: %d (every 128-byte line of the launch touched exactly once, table >> L2 + Infinity Cache)" % expect)
for k, v in sorted(rows.items()):
    kb = sum(v) / len(v)
    print("%-12s FETCH_SIZE %.0f KB per launch (%d launches) = %.3f of the bytes -> multiply the counter by %.3f" % (k, kb, len(v), kb * 1024 / expect, expect / (kb * 1024)))

#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters:  python tools/pmc_kernels.py <counter_collection.csv> [name filter]"""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if flt and flt not in name:
        continue
    short = name.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
    if "<" in name and "(anonymous namespace)::" in name:
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
    rows[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(rows.items()):
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())}, "n=%d" % len(next(iter(d.values()))))

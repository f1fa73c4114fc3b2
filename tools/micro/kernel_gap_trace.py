#!/usr/bin/env python3
"""Per store form: mean duration of produce / consume and the gaps around them, from a rocprofv3 --kernel-trace csv of
tools/micro/kernel_gap (argument: the *_kernel_trace.csv)."""
import csv
import sys
from collections import defaultdict

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
stat = defaultdict(lambda: defaultdict(list))
for p, r, n in zip(rows, rows[1:], rows[2:] + [None]):
    if "produce" not in r["Kernel_Name"] or n is None:
        continue
    mode = r["Kernel_Name"].split("<")[1].split(">")[0]
    size = int(r["Grid_Size_X"])  # same grid for all: key by the duration class instead
    key = (mode, round(int(n["End_Timestamp"]) - int(n["Start_Timestamp"]), -4))
    s = stat[(mode, rows.index(r) // 106)]
    s["produce"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    s["gap_after_produce"].append(int(n["Start_Timestamp"]) - int(r["End_Timestamp"]))
    s["consume"].append(int(n["End_Timestamp"]) - int(n["Start_Timestamp"]))
    s["gap_before_produce"].append(int(r["Start_Timestamp"]) - int(p["End_Timestamp"]))
for k in sorted(stat, key=lambda k: (k[1], k[0])):
    s = stat[k]
    print("block %d mode %s: produce %7.1f us, gap %5.1f us, consume %7.1f us, gap before the next produce %5.1f us" % (
        k[1], k[0], *(sum(s[n]) / len(s[n]) / 1e3 for n in ("produce", "gap_after_produce", "consume", "gap_before_produce"))))

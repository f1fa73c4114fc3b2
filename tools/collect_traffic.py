#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs into profiles/traffic.json
(HBM bytes per launch of each message kernel), applying the gfx950 corrections of
MI355X_MICROARCH.md section HBM: counters are in KiB; FETCH_SIZE reports half of the bytes of wide
coalesced reads (x2); WRITE_SIZE is exact.

    python tools/collect_traffic.py <dir with FETCH_SIZE/ and WRITE_SIZE/ runs> profiles/traffic.json
"""
import collections
import csv
import glob
import json
import os
import sys


def kernel_key(name):
    """message_scatter_{fwd,bwd}[_l0] and message_bwd_finish, one key per kernel; main() then adds the finish launch's
    bytes to the backward operator (the two launches are timed as one by bench.py)."""
    if "message_scatter" not in name and "message_bwd_finish" not in name:
        return None
    if "message_bwd_finish" in name:
        return "message_bwd_finish"
    base = "message_scatter_fwd" if "fwd" in name else "message_scatter_bwd"
    has_vec = "<true" in name or "ILb1" in name
    return base + ("" if has_vec else "_l0")


def main():
    src, dst = sys.argv[1], sys.argv[2]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = kernel_key(r["Kernel_Name"])
            if k and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"_note": "HBM bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (rocprofv3 --pmc, separate passes; "
                    "gfx950: FETCH_SIZE counts 64 B per 128-B request)"}
    for k, d in sorted(vals.items()):
        fetch = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1)
        write = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
        out[k] = 2 * fetch * 1024 + write * 1024
        out[k + "_detail"] = {"FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB": write, "launches": len(d["FETCH_SIZE"])}
    # bench.py's HIP-event time of "message_scatter_bwd" brackets the channel-per-lane kernel AND its finish launch:
    # the operator's bytes are the sum of both (the finish launch runs once per layer with vec rows)
    if "message_scatter_bwd" in out and "message_bwd_finish" in out:
        out["message_scatter_bwd_kernel_only"] = out["message_scatter_bwd"]
        out["message_scatter_bwd"] = out["message_scatter_bwd"] + out["message_bwd_finish"]
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

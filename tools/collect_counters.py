#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter CSVs (any counters, several passes) -> one JSON.

    python tools/collect_counters.py <dir holding the pass directories> <out.json> [substring filter ...]

Kernels are keyed by a short name (template arguments folded); every counter is averaged over the dispatches of a
kernel.  Derived figures (gfx950: SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave, summed over
waves; SQ_BUSY_CYCLES counts per SE -- MI355X_MICROARCH.md, cycle-constants table):
  valu_issue_share = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES    share of wave lifetime spent issuing VALU
  lds_issue_share  = SQ_ACTIVE_INST_LDS  / SQ_WAVE_CYCLES
  wait_share       = SQ_WAIT_ANY / SQ_WAVE_CYCLES            parked at s_waitcnt / barrier
  stall_share      = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES       issue stalls
  mfma_busy_share  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 4 SIMD * 256 CU)  (when both were collected)
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"([A-Za-z0-9_:]+)(<.*>)?", name)
    base = m.group(1) if m else name
    targs = m.group(2) if (m and m.group(2)) else ""
    if base.startswith("message_scatter") and targs:
        base += targs.replace(" ", "")
    elif base.startswith("Cijk") or "gemm" in base.lower():
        base = base[:60]
    return base


def main():
    src, dst, filt = sys.argv[1], sys.argv[2], sys.argv[3:]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if filt and not any(s in k for s in filt):
                continue
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {"vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                       "lds_bytes": int(r["LDS_Block_Size"]), "workgroup": int(r["Workgroup_Size"]), "grid": int(r["Grid_Size"])}
    out = {}
    for k, d in sorted(vals.items()):
        avg = {c: sum(v) / len(v) for c, v in d.items()}
        ent = dict(meta[k])
        ent["dispatches_seen"] = max(len(v) for v in d.values())
        ent["counters"] = avg
        wc = avg.get("SQ_WAVE_CYCLES")
        if wc:
            for key, c in [("valu_issue_share", "SQ_ACTIVE_INST_VALU"), ("lds_issue_share", "SQ_ACTIVE_INST_LDS"),
                           ("wait_share", "SQ_WAIT_ANY"), ("stall_share", "SQ_WAIT_INST_ANY"),
                           ("any_issue_share", "SQ_ACTIVE_INST_ANY"), ("vmem_issue_share", "SQ_ACTIVE_INST_VMEM"),
                           ("lds_stall_share", "SQ_WAIT_INST_LDS")]:
                if c in avg:
                    ent[key] = avg[c] / wc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "GRBM_GUI_ACTIVE" in avg and avg["GRBM_GUI_ACTIVE"] > 0:
            ent["mfma_busy_share"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["GRBM_GUI_ACTIVE"] / 8.0 * 4 * 256)
        if "SQ_INSTS_VALU" in avg and "SQ_WAVES" in avg and avg["SQ_WAVES"] > 0:
            ent["valu_insts_per_wave"] = avg["SQ_INSTS_VALU"] / avg["SQ_WAVES"]
        out[k] = ent
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    for k, e in out.items():
        print(k, {x: (round(y, 4) if isinstance(y, float) else y) for x, y in e.items() if x != "counters"})


if __name__ == "__main__":
    main()

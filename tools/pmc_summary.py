"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: counters averaged per dispatch, the average
dispatch duration, and the derived ratios used in DESIGN.md / profiles/ (MFMA-pipe busy fraction, effective
clock = GRBM_GUI_ACTIVE / duration).

    python tools/pmc_summary.py <kernel-name-substring> <dir-or-csv> [<dir-or-csv> ...]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def rows(path):
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        with open(f, newline="") as fh:
            yield from csv.DictReader(fh)


def summarise(substr, paths):
    # (kernel, grid) -> counter -> [sum, n]; durations per dispatch
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(dict)
    for p in paths:
        for r in rows(p):
            if substr not in r["Kernel_Name"]:
                continue
            key = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))
            a = acc[key][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            dur[key][(p, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    out = {}
    for key, ctrs in acc.items():
        c = {k: v[0] / v[1] for k, v in ctrs.items()}
        d = sum(dur[key].values()) / len(dur[key])
        e = {"dispatches": len(dur[key]), "avg_seconds": d, "counters": c}
        # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs and SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
        if "GRBM_GUI_ACTIVE" in c:
            e["effective_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8 / d * 1e-9
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            e["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
        if "SQ_WAVE_CYCLES" in c:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
                      "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM"):
                if k in c:
                    e[k.lower() + "_per_wave_cycle"] = c[k] / c["SQ_WAVE_CYCLES"]
        out[f"{key[0]} grid={key[1]}"] = e
    return out


if __name__ == "__main__":
    print(json.dumps(summarise(sys.argv[1], sys.argv[2:]), indent=1))

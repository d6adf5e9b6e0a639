#!/usr/bin/env python3
"""Reduce rocprofv3 CSV output (kernel trace / counter collection) to per-kernel averages.

usage: pmc_reduce.py <dir-or-csv> [<dir-or-csv> ...]
Prints JSON: {kernel: {counter: mean-per-dispatch, "dispatches": n, "avg_ns": mean duration}}.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    for key in ("k_integrate", "k_calib_rmw", "k_fill_u32", "k_raycast", "k_ray_begin", "k_mc_classify", "k_mc_emit",
                "k_mc_counts", "k_sample", "k_block", "k_planes"):
        if key in name:
            return key
    return name[:48]


def main():
    files = []
    for p in sys.argv[1:]:
        if os.path.isdir(p):
            files += glob.glob(os.path.join(p, "**", "*.csv"), recursive=True)
        else:
            files.append(p)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            cols = rd.fieldnames or []
            if "Counter_Name" in cols:  # counter collection
                per_dispatch = defaultdict(float)
                kname = {}
                for r in rd:
                    key = (r.get("Dispatch_Id"), r["Counter_Name"])
                    per_dispatch[key] += float(r["Counter_Value"])
                    kname[r.get("Dispatch_Id")] = short(r["Kernel_Name"])
                for (disp, cname), val in per_dispatch.items():
                    acc[kname[disp]][cname].append(val)
            elif "Start_Timestamp" in cols and "Kernel_Name" in cols:  # kernel trace
                for r in rd:
                    acc[short(r["Kernel_Name"])]["duration_ns"].append(
                        float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    out = {}
    for k, d in acc.items():
        out[k] = {c: sum(v) / len(v) for c, v in d.items()}
        out[k]["dispatches"] = max(len(v) for v in d.values())
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()

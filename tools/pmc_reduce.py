#!/usr/bin/env python3
"""Reduce rocprofv3 CSV output (kernel trace / counter collection) to per-kernel averages.

usage: pmc_reduce.py <dir-or-csv> [<dir-or-csv> ...]
Prints JSON: {kernel: {counter: mean-per-dispatch, "dispatches": n, "duration_ns": mean duration}}.
Template instances of k_integrate and k_mc_classify are kept apart ("k_integrate<0,true,true,false,true>"): bench.py
warms up through the COUNTING instance (<..., true, ...> in the fourth place), so the non-counting instance holds
exactly the timed launches; "k_integrate" (no arguments) is the mean over all instances.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

KEYS = ("k_integrate_rgbn", "k_integrate_plain", "k_integrate2", "k_integrate_pc", "k_integrate_p", "k_integrate", "k_calib_rmw", "k_calib_read", "k_fill_u32", "k_raycast", "k_ray_begin",
        "k_mc_classify", "k_mc_emit", "k_mc_counts", "k_sample", "k_block", "k_planes", "k_cull", "k_ingest")


def names(name):
    """Short names a dispatch is accounted under: the kernel, and (templates) the instance."""
    for key in KEYS:
        if key in name:
            m = re.search(re.escape(key) + r"<([^>]*)>", name)
            if m and key in ("k_integrate", "k_integrate2", "k_integrate_p", "k_integrate_pc", "k_mc_classify", "k_raycast", "k_calib_read"):
                return [key, key + "<" + m.group(1).replace(" ", "") + ">"]
            return [key]
    return [name[:48]]


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
                    kname[r.get("Dispatch_Id")] = names(r["Kernel_Name"])
                for (disp, cname), val in per_dispatch.items():
                    for k in kname[disp]:
                        acc[k][cname].append(val)
            elif "Start_Timestamp" in cols and "Kernel_Name" in cols:  # kernel trace
                for r in rd:
                    for k in names(r["Kernel_Name"]):
                        acc[k]["duration_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    out = {}
    for k, d in acc.items():
        out[k] = {c: sum(v) / len(v) for c, v in d.items()}
        out[k]["dispatches"] = max(len(v) for v in d.values())
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()

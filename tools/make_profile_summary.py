#!/usr/bin/env python3
"""Copy the judged parts of a tools/run_rocprof.sh run (gpurun_out/prof_<tag>/) into profiles/ and derive
profiles/pmc_traffic.json (HBM bytes per k_integrate launch) the way MI355X_MICROARCH.md prescribes:
FETCH_SIZE and WRITE_SIZE from separate --pmc passes, in KiB; FETCH_SIZE doubled on gfx950 for wide
coalesced reads, the factor being CHECKED here against k_calib_rmw's exactly known byte count."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    for name in ("kernel_stats.csv", "summary_trace.json", "summary_pmc_FETCH_SIZE.json", "summary_pmc_WRITE_SIZE.json",
                 "summary_pmc_SQ.json", "summary_pmc_SQ2.json", "bench_under_rocprof.json", "prof_FETCH_SIZE.json"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, f"{tag}_{name}"))
    fetch = json.load(open(os.path.join(src, "summary_pmc_FETCH_SIZE.json")))
    write = json.load(open(os.path.join(src, "summary_pmc_WRITE_SIZE.json")))
    prof = json.loads(open(os.path.join(src, "prof_FETCH_SIZE.json")).read().strip().splitlines()[-1])
    trace = json.load(open(os.path.join(src, "summary_trace.json")))
    cal_r = prof["sweep_bytes_read"] / (fetch["k_calib_rmw"]["FETCH_SIZE"] * 1024)
    cal_w = prof["sweep_bytes_written"] / (write["k_calib_rmw"]["WRITE_SIZE"] * 1024)
    rd = fetch["k_integrate"]["FETCH_SIZE"] * 1024 * round(cal_r)
    wr = write["k_integrate"]["WRITE_SIZE"] * 1024 * round(cal_w)
    key = f"{prof['res']}x{prof['res']}x{prof['planes']}_c{prof['color']}_{prof.get('layout', 'f32w')}"
    out_path = os.path.join(dst, "pmc_traffic.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    out[key] = {
        "tag": tag,
        "hbm_bytes_per_launch": rd + wr,
        "read_bytes": rd, "written_bytes": wr,
        "fetch_size_correction": round(cal_r), "write_size_correction": round(cal_w),
        "calibration": {"kernel": "k_calib_rmw", "known_read_bytes": prof["sweep_bytes_read"],
                        "FETCH_SIZE_KiB": fetch["k_calib_rmw"]["FETCH_SIZE"], "ratio_read": cal_r,
                        "known_written_bytes": prof["sweep_bytes_written"],
                        "WRITE_SIZE_KiB": write["k_calib_rmw"]["WRITE_SIZE"], "ratio_write": cal_w},
        "algorithmic_bytes_per_launch": prof["alg_bytes_per_launch"],
        "layout_bytes_per_launch": prof.get("layout_bytes_per_launch"),
        "k_integrate_avg_ns_kernel_trace": trace["k_integrate"]["duration_ns"],
    }
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out[key], indent=1))


if __name__ == "__main__":
    main()

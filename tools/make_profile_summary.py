#!/usr/bin/env python3
"""Copy the judged parts of a tools/run_rocprof.sh run (gpurun_out/prof_<tag>/) into profiles/ and derive
profiles/pmc_traffic.json: HBM bytes per TIMED k_integrate launch of bench.py, the way MI355X_MICROARCH.md prescribes
-- FETCH_SIZE and WRITE_SIZE from separate --pmc passes, in KiB; FETCH_SIZE doubled on gfx950 for wide coalesced
reads, the factor being CHECKED here against k_calib_rmw's exactly known byte count in the same process.  The entry
is stamped with the hash of the kernel sources (bench.kernel_sha16) and the commit: bench.py only quotes it while
that hash matches the tree it runs from."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def last_json_line(path):
    for line in reversed(open(path).read().strip().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            return json.loads(line)
    raise RuntimeError(f"no JSON line in {path}")


def timed_instance(summary):
    """The non-counting template instance of the integrate kernel = bench.py's timed launches: k_integrate<..., COUNT = false (4th
    argument), ...>, or -- round 6 -- the software-pipelined kernels k_integrate_p / k_integrate_pc<ORDER, COUNT = false>.  A key's
    timed launches all go through ONE of them; its counting twin holds the warm-up and the byte-counting pass."""
    found = []
    for k in summary:
        if k.startswith("k_integrate<") and k.rstrip(">").split("<")[1].split(",")[3] == "false":
            found.append(k)
        if (k.startswith("k_integrate_p<") or k.startswith("k_integrate_pc<")) and k.rstrip(">").split("<")[1].split(",")[1] == "false":
            found.append(k)
    if not found:
        raise RuntimeError("no non-counting integrate instance in " + ", ".join(summary))
    return max(found, key=lambda k: summary[k].get("dispatches", 0))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    for name in ("kernel_stats.csv", "summary_trace.json", "summary_pmc_FETCH_SIZE.json", "summary_pmc_WRITE_SIZE.json",
                 "summary_pmc_SQ.json", "summary_pmc_SQ2.json", "bench_under_rocprof.json", "bench_pmc_FETCH_SIZE.json",
                 "bench_pmc_WRITE_SIZE.json"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, f"{tag}_{name}"))
    fetch = json.load(open(os.path.join(src, "summary_pmc_FETCH_SIZE.json")))
    write = json.load(open(os.path.join(src, "summary_pmc_WRITE_SIZE.json")))
    trace = json.load(open(os.path.join(src, "summary_trace.json")))
    bf = last_json_line(os.path.join(src, "bench_pmc_FETCH_SIZE.json"))
    bw = last_json_line(os.path.join(src, "bench_pmc_WRITE_SIZE.json"))
    bt = last_json_line(os.path.join(src, "bench_under_rocprof.json"))
    assert bf["roofline"]["kernel_sha16"] == bw["roofline"]["kernel_sha16"] == bt["roofline"]["kernel_sha16"]
    cal = bf["calibration"]
    cal_r = cal["known_read_bytes"] / (fetch["k_calib_rmw"]["FETCH_SIZE"] * 1024)
    cal_w = bw["calibration"]["known_written_bytes"] / (write["k_calib_rmw"]["WRITE_SIZE"] * 1024)
    # narrow reads: what FETCH_SIZE tallies per dword read at strides 4 / 64 / 128 (bench.py --calib runs the sweeps)
    narrow = []
    for e in cal.get("narrow_reads", []):
        f = fetch.get(e["kernel"])
        if f:
            narrow.append({**e, "FETCH_SIZE_KiB": f["FETCH_SIZE"], "fetch_bytes_per_dword_read": f["FETCH_SIZE"] * 1024 / e["dwords_read"],
                           "span_over_fetch": e["span_bytes"] / (f["FETCH_SIZE"] * 1024)})
    inst = timed_instance(fetch)
    assert fetch[inst]["dispatches"] == bf["steps"] and write[inst]["dispatches"] == bw["steps"], "timed instance != timed launches"
    rd = fetch[inst]["FETCH_SIZE"] * 1024 * round(cal_r)
    wr = write[inst]["WRITE_SIZE"] * 1024 * round(cal_w)
    cfg = bf["config"]
    planes = bf.get("multi_gpu", {}).get("planes_per_gpu", cfg["grid"][2])
    key = f"{cfg['grid'][0]}x{cfg['grid'][1]}x{planes}_c{int(cfg['color'])}_{cfg['layout']}" + ("_saturated" if cfg.get("presaturate_launches") else "")
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True).strip()
    except Exception:
        head = None
    out_path = os.path.join(dst, "pmc_traffic.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    out[key] = {
        "tag": tag, "kernel_sha16": bf["roofline"]["kernel_sha16"], "git_head_when_summarised": head,
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --warmup 2 --cpu-baseline 0 --scene-b 0 --host-path 0 "
                   f"--steps {bf['steps']} --calib 2" + (" " + sys.argv[2] if len(sys.argv) > 2 else ""),
        "kernel_instance": inst, "timed_launches_averaged": fetch[inst]["dispatches"],
        "hbm_bytes_per_launch": rd + wr,
        "read_bytes": rd, "written_bytes": wr,
        "fetch_size_correction": round(cal_r), "write_size_correction": round(cal_w),
        "calibration": {"kernel": "k_calib_rmw", "known_read_bytes": cal["known_read_bytes"],
                        "FETCH_SIZE_KiB": fetch["k_calib_rmw"]["FETCH_SIZE"], "ratio_read": cal_r,
                        "known_written_bytes": bw["calibration"]["known_written_bytes"],
                        "WRITE_SIZE_KiB": write["k_calib_rmw"]["WRITE_SIZE"], "ratio_write": cal_w,
                        "narrow_reads": narrow},
        "algorithmic_bytes_per_launch": bf["roofline"]["algorithmic_bytes_per_launch"],
        # what the kernel has to move (distance words it rebuilds from the counts are not read): the figure the traffic follows
        "bytes_moved_per_launch": (bf["roofline"].get("bytes_moved") or {}).get("per_launch"),
        "kernel_ms_in_profile": {"kernel_trace_avg_timed_instance": trace[timed_instance(trace)]["duration_ns"] / 1e6,
                                 "bench_line_same_run": bt["roofline"]["kernel_ms"],
                                 "bench_line_fetch_pass": bf["roofline"]["kernel_ms"]},
        "frac_of_8TBps_by_traffic": (rd + wr) / (trace[timed_instance(trace)]["duration_ns"] * 1e-9) / 8e12,
    }
    # k_integrate2 (two frames per sweep: the bench line's extras.fused2 leg), when the profiled command ran it: its own
    # counters instead of an algorithmic estimate
    def timed2(summary):  # the non-counting instance of k_integrate2 (last template argument false): the leg's timed launches + its one probe launch
        return next((k for k in summary if k.startswith("k_integrate2<") and k.rstrip(">").endswith("false")), None)
    if timed2(fetch) and timed2(write) and timed2(trace):
        k2 = timed2(fetch)
        rd2 = fetch[k2]["FETCH_SIZE"] * 1024 * round(cal_r)
        wr2 = write[timed2(write)]["WRITE_SIZE"] * 1024 * round(cal_w)
        ms2 = trace[timed2(trace)]["duration_ns"] / 1e6
        f2 = (bt.get("extras") or {}).get("fused2") or {}
        out[key]["fused2"] = {"kernel": k2, "launches_averaged": fetch[k2]["dispatches"],
                              "read_bytes_per_launch": rd2, "written_bytes_per_launch": wr2, "hbm_bytes_per_launch": rd2 + wr2,
                              "hbm_bytes_per_frame": (rd2 + wr2) / 2, "kernel_ms_trace_avg": ms2, "ms_per_frame": ms2 / 2,
                              "frac_of_8TBps_by_traffic": (rd2 + wr2) / (ms2 * 1e-3) / 8e12,
                              "bench_line_ms_per_frame_same_run": f2.get("ms_per_frame"),
                              "kernel_counted_bytes_per_launch": f2.get("bytes_moved_per_launch")}
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out[key], indent=1))


if __name__ == "__main__":
    main()

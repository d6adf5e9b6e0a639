#!/usr/bin/env python3
"""A PREDICTED strong-scaling table from ONE GPU -- labelled as a prediction everywhere: no scaling
number can be measured on a one-GPU box, and none of this is one.

For N in {1, 2, 4, 8} and every rank r of N, `bench.py --emulate-rank r --of N` integrates, on the one GPU, exactly the Z-slab
[r * 2048 / N, (r + 1) * 2048 / N) of the 2048^3 grid that rank would own, with the real turntable frames: per-slab kernel_ms
(HIP events around each launch) and observed voxels per frame.  A step of the N-rank run lasts as long as its slowest slab
(the ranks meet at the next frame's broadcast), so

    predicted frames/s (N) = 1000 / max_r kernel_ms(r, N)        (+ the broadcast where it is not hidden under the kernel)

The RCCL broadcast of one frame is timed at world size 1 (TSDF_BENCH_FORCE_DIST=1: communicator + device-tensor broadcast
really run, but over no link) -- a floor for its fixed cost only; on xGMI 2.4 MB at ~50 GB/s per link adds ~50 us per hop.

usage: predict_scaling.py [--color 0|1] [--steps K] > profiles/r06_predicted_scaling.json
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(extra, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--warmup", "3", "--cpu-baseline", "0", "--host-path", "0", "--extras", "0"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=dict(os.environ, **(env or {})))
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    try:
        return json.loads(lines[-1])
    except Exception:  # noqa: BLE001
        return {"error": f"exit {p.returncode}: " + p.stderr.strip()[-1500:]}


def rccl_leg(color, steps):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    d = bench(["--color", str(color), "--steps", str(steps)],
              env={"TSDF_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    mg = d.get("multi_gpu") or {}
    return {"frame_broadcast_ms_isolated": mg.get("frame_broadcast_ms_isolated"), "rccl_version": mg.get("rccl_version"),
            "kernel_ms_with_collectives_in_the_loop": (d.get("roofline") or {}).get("kernel_ms"),
            "ms_per_step": d.get("ms_per_step"), "error": d.get("error")}


def main():
    a = sys.argv[1:]
    color = int(a[a.index("--color") + 1]) if "--color" in a else 1
    steps = int(a[a.index("--steps") + 1]) if "--steps" in a else 20
    if "--rccl-only" in a:  # (re-)measure the world-1 RCCL leg alone and merge it into an existing table: --rccl-only FILE
        path = a[a.index("--rccl-only") + 1]
        out = json.load(open(path))
        out["rccl_world1"] = rccl_leg(color, steps)
        json.dump(out, open(path, "w"), indent=1)
        print(json.dumps(out["rccl_world1"]))
        return
    out = {"what": "PREDICTED strong scaling of integrateCloud at 2048^3, 640x480, from per-slab kernel times measured on ONE GPU; "
                   "not a scaling measurement", "color": bool(color), "steps": steps, "per_N": {}}
    t1 = None
    for n in (1, 2, 4, 8):
        rows = []
        for r in range(n):
            d = bench(["--color", str(color), "--steps", str(steps), "--emulate-rank", str(r), "--of", str(n)])
            if "error" in d:
                rows.append({"rank": r, "error": d["error"]})
                continue
            rows.append({"rank": r, "z": [d["emulated_slab"]["z_begin"], d["emulated_slab"]["z_end"]], "kernel_ms": d["roofline"]["kernel_ms"],
                         "observed_voxels_per_frame": d["config"]["observed_voxels_per_frame"],
                         "bytes_moved_per_launch": d["roofline"]["bytes_moved"]["per_launch"], "instance": d["config"]["last_launch"]["instance"]})
        ok = [x for x in rows if "kernel_ms" in x]
        e = {"slabs": rows}
        if ok:
            worst = max(ok, key=lambda x: x["kernel_ms"])
            obs = [x["observed_voxels_per_frame"] for x in ok]
            e.update({"max_kernel_ms": worst["kernel_ms"], "slowest_rank": worst["rank"], "mean_kernel_ms": sum(x["kernel_ms"] for x in ok) / len(ok),
                      "observed_voxel_imbalance_max_over_mean": max(obs) / (sum(obs) / len(obs)) if sum(obs) else None,
                      "predicted_frames_per_s": 1e3 / worst["kernel_ms"]})
            if n == 1:
                t1 = worst["kernel_ms"]
            if t1:
                e["predicted_speedup"] = t1 / worst["kernel_ms"]
                e["predicted_efficiency"] = t1 / worst["kernel_ms"] / n
        out["per_N"][str(n)] = e
    out["rccl_world1"] = rccl_leg(color, steps)
    print(json.dumps(out, indent=1))
    print("# N  max kernel ms  slowest rank  imbalance  predicted frames/s  predicted efficiency", file=sys.stderr)
    for n, e in out["per_N"].items():
        if "max_kernel_ms" in e:
            print(f"# {n}  {e['max_kernel_ms']:.3f}  {e['slowest_rank']}  {e['observed_voxel_imbalance_max_over_mean']:.3f}  "
                  f"{e['predicted_frames_per_s']:.1f}  {e.get('predicted_efficiency', 0):.3f}", file=sys.stderr)


if __name__ == "__main__":
    main()

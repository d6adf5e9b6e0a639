#!/usr/bin/env python3
"""End-to-end rate of the C++ drop-in: cpu_tsdf::TSDFVolumeOctree::integrateCloud on
pcl::PointCloud<PointXYZRGBA> clouds in host memory -- AoS strip into the pinned slot + upload + kernel -- through the
product's timing program cpu_tsdf_amd/bin/dropin_rate (csrc/prog/dropin_rate.cpp), frame pairing off and on, beside the
resident-frame rate of the same kernel (bench.py, frames already in HBM: single frames and k_integrate2 pairs).

usage: cpp_path_timing.py [N_FRAMES [RES ...]]      -> one JSON object on stdout (profiles/r06_cpp_path_timing.json)"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cpu_tsdf_amd import synth  # noqa: E402


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    sizes = [int(a) for a in sys.argv[2:]] or [2048]
    exe = os.path.join(ROOT, "cpu_tsdf_amd", "bin", "dropin_rate")
    out = {"program": "cpu_tsdf_amd/bin/dropin_rate", "host_cores": os.cpu_count()}
    W, H = 640, 480
    for res in sizes:
        sc = synth.scene_a(res, W, H)
        e = {}
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            path = os.path.join(td, "frames.bin")
            with open(path, "wb") as f:
                for i in range(8):
                    tr = synth.turntable_pose(i, 24, sc.size)
                    f.write(np.ascontiguousarray(tr, dtype=np.float64).tobytes())
                    f.write(np.ascontiguousarray(sc.depth(tr), dtype=np.float32).tobytes())
                    f.write(np.ascontiguousarray(sc.bgra(i), dtype=np.uint8).tobytes())
            for color in (1, 0):
                for pairing in (0, 1):
                    p = subprocess.run([exe, str(res), str(W), str(H), str(color), str(pairing), str(n_frames), path], capture_output=True, text=True, timeout=300)
                    key = f"color{color}_pairing{pairing}"
                    e[key] = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else {"error": p.stderr.strip()[-300:]}
        # the resident-frame rate of the same kernels on this box: bench.py with frames in HBM
        for color in (1, 0):
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--res", str(res), "--color", str(color), "--steps", "20", "--warmup", "3",
                                "--cpu-baseline", "0", "--host-path", "0", "--scene-b", "0", "--keys", "0", "--extras", "2"], capture_output=True, text=True, timeout=300)
            try:
                d = json.loads(p.stdout.strip().splitlines()[-1])
                e[f"color{color}_resident"] = {"frames_per_s": d["frames_per_s"], "kernel_ms": d["roofline"]["kernel_ms"],
                                                "fused2_frames_per_s": (d.get("extras", {}).get("fused2") or {}).get("frames_per_s")}
            except Exception as ex:  # noqa: BLE001
                e[f"color{color}_resident"] = {"error": repr(ex), "stderr": p.stderr.strip()[-300:]}
        for color in (1, 0):
            r = e.get(f"color{color}_resident", {})
            for pairing, ref in ((0, r.get("frames_per_s")), (1, r.get("fused2_frames_per_s"))):
                c = e.get(f"color{color}_pairing{pairing}", {})
                if ref and "sustained_frames_per_s" in c:
                    c["fraction_of_resident_rate"] = c["sustained_frames_per_s"] / ref
        out[f"{res}^3"] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

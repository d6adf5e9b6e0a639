"""Streamed save() / load() at a large resolution through the C++ drop-in shell: wall time, file size and the
host memory high-water mark (the point of streaming: it must stay near one block, not near the grid).

    python tools/save_load_timing.py --res 1024 --frames 8 --out gpurun_out/save_load.json

Goes through the Python mirror of the class (tsdf_hip_save / tsdf_hip_load over ctypes)."""
import argparse
import json
import os
import resource
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpu_tsdf_amd import synth  # noqa: E402
from cpu_tsdf_amd.volume import TSDFVolumeOctree  # noqa: E402


def rss_mb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--color", type=int, default=1)
    ap.add_argument("--verify", type=int, default=1)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    sc = synth.scene_a(a.res)

    def mk():
        v = TSDFVolumeOctree()
        v.setResolution(a.res, a.res, a.res)
        v.setGridSize(sc.size, sc.size, sc.size)
        v.setImageSize(640, 480)
        v.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
        v.setSensorDistanceBounds(0.0, 3 * sc.size)
        v.setIntegrateColor(bool(a.color))
        v.reset()
        return v
    dv = mk()
    for i in range(a.frames):
        tr = synth.turntable_pose(i, a.frames, sc.size)
        dv.integrateCloud(sc.depth(tr), sc.bgra(i) if a.color else None, tr)
    grid_mb = a.res ** 3 * (11 if a.color else 8) / 2 ** 20
    out = {"res": a.res, "frames": a.frames, "color": a.color, "grid_mb": grid_mb, "rss_before_mb": rss_mb(),
           "chunk": int(os.environ.get("TSDF_HIP_VOL_CHUNK", 256))}
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        path = os.path.join(td, "big.vol")
        t0 = time.time()
        dv.save(path)
        out["save_s"] = time.time() - t0
        out["file_mb"] = os.path.getsize(path) / 2 ** 20
        out["rss_after_save_mb"] = rss_mb()
        dv2 = TSDFVolumeOctree()  # configured by the file
        t0 = time.time()
        dv2.load(path)
        out["load_s"] = time.time() - t0
        out["rss_after_load_mb"] = rss_mb()
        if a.verify:  # whole-grid compare (this is what needs the big host buffers, not save / load)
            d1, w1, c1 = dv.download()
            d2, w2, c2 = dv2.download()
            out["identical"] = bool(np.array_equal(d1.view(np.uint32), d2.view(np.uint32)) and np.array_equal(w1, w2)
                                    and (c1 is None or np.array_equal(c1, c2)))
            out["observed_voxels"] = int((w1 > 0).sum())
    out["host_high_water_over_grid"] = (out["rss_after_load_mb"] - out["rss_before_mb"]) / grid_mb
    print(json.dumps(out))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""End-to-end wall time of the two `integrate` programs on one synthetic sequence at the reference's default scale
(README-style: 10 m volume, 2048^3 nominal resolution, camera inside, 640x480 unorganised binary PCDs, colour):
cpu_tsdf_amd/bin/integrate (GPU) vs oracle/_ref/ref_integrate (the reference's own program, native adaptive octree,
OpenMP on all cores).  The meshes are not byte-comparable here (adaptive octree vs dense grid);
tests/test_programs_gpu.py does the byte-level comparison in dense mode.  Prints one JSON line."""
import json
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import sequence_util as su  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    d = su.digit_free_dir("timing")
    try:
        su.make_sequence(os.path.join(d, "in"), n_frames=n, width=640, height=480, kinds=("binary",))  # no text parsing in the timing
        args = ["--in", os.path.join(d, "in"), "--volume-size", 10, "--cell-size", 10.0 / 2048, "--color"]
        out = {"frames": n, "grid": 2048, "volume_m": 10.0}
        for tag, exe in (("gpu_program", su.OUR_INTEGRATE), ("reference_program", su.REF_INTEGRATE)):
            env_threads = os.cpu_count() or 1
            os.environ["OMP_NUM_THREADS"] = str(env_threads)
            t0 = time.perf_counter()
            rc, log = su.run(exe, args + ["--out", os.path.join(d, tag)], timeout=3000)
            dt = time.perf_counter() - t0
            v, c, f = su.read_ply(os.path.join(d, tag, "mesh.ply")) if rc == 0 else (None, None, [])
            out[tag] = {"rc": rc, "wall_s": dt, "triangles": len(f), "threads": env_threads}
        print(json.dumps(out))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()

"""Where does a host-entry integrateCloud call spend its time when the caller does other work between calls?
Prints the per-call wall times of integrateCloud (host entry point, synchronous) in call order for a few caller
behaviours: nothing in between, renderView in between, numpy frame synthesis in between, CPU oracle in between."""
import json
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from cpu_tsdf_amd import synth  # noqa: E402
from cpu_tsdf_amd.volume import TSDFVolumeOctree  # noqa: E402
from oracle.oracle import SlabOracle  # noqa: E402

res = 1024
sc = synth.scene_a(res)
v = TSDFVolumeOctree()
v.setResolution(res, res, res)
v.setGridSize(sc.size, sc.size, sc.size)
v.setImageSize(640, 480)
v.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
v.setSensorDistanceBounds(0.0, 3 * sc.size)
v.setIntegrateColor(False)
v.reset()
orcs = [SlabOracle(v._p, z, z + 2) for z in (511, 341, 724)]
N = 40
out = {}
N = int(os.environ.get("PROBE_CALLS", 40))
for name, render, synth_between, oracle in [("plain", 0, 0, 0), ("render_synth_oracle", 1, 1, 1), ("sleep60ms", 0, 0, 2)]:
    poses = [synth.turntable_pose(i, 300, sc.size, tilt=0.15 * np.sin(i * 0.05)) for i in range(60, 60 + N)]
    deps = [sc.depth(p, noise_seed=7 + i) for i, p in enumerate(poses)]
    ts = []
    for i in range(N):
        dep = deps[i]
        if synth_between:
            dep = sc.depth(poses[i], noise_seed=7 + i)
        t0 = time.perf_counter()
        v.integrateCloud(dep, None, poses[i])
        ts.append((time.perf_counter() - t0) * 1e3)
        if render:
            v.renderView(poses[i], 1)
        if oracle == 2:
            time.sleep(0.06)
        if oracle == 1:
            T = synth.cam_from_vol_f32(poses[i])
            for o in orcs:
                o.integrate(dep, None, T)
    out[name] = {"median_ms": float(np.median(ts)), "by_tenth": [round(float(np.median(c)), 2) for c in np.array_split(np.array(ts), 10)],
                 "last_24_in_order": [round(t, 1) for t in ts[-24:]]}
print(json.dumps(out))

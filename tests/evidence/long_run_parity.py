#!/usr/bin/env python3
"""BASELINE configs[3] end to end on one GPU: 2048^3 grid, integrateColor, N (default 1000) DISTINCT 640x480
frames pushed through the drop-in host entry point (frame upload included), then marching cubes; while it runs,
the CPU oracle owns a few plane groups of the same grid (oracle.SlabOracle) and integrates the same frames, and
at the end those planes are compared bit for bit.  Prints one JSON line (committed under profiles/ as evidence of
sustained throughput -- weights saturate at frame 100 -- and of parity at the full configuration)."""
import argparse
import json
import os
import sys
import time

import numpy as np

# The oracle's OpenMP team must sleep, not spin, between its parallel regions: spinning threads on every host
# core delay the HIP runtime's completion signals and show up as "GPU time" of the next call.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cpu_tsdf_amd import capi, synth  # noqa: E402
from cpu_tsdf_amd.volume import MarchingCubesTSDFOctree, TSDFVolumeOctree  # noqa: E402
from oracle.oracle import SlabOracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=2048)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--color", type=int, default=1)
    ap.add_argument("--check-every", type=int, default=0, help="also compare after every K frames (0 = only at the end)")
    ap.add_argument("--planes", type=int, default=0, help="integrate only a Z-slab of this many central planes (one rank's share)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--pipelined", type=int, default=0, help="hand frames over with tsdf_hip_integrate_async")
    ap.add_argument("--pairing", type=int, default=0, help="frame pairing of the pipelined path (tsdf_hip_set_frame_pairing): two frames per sweep")
    ap.add_argument("--raycast-every", type=int, default=0, help="renderView from the current pose every K frames (configs[2])")
    a = ap.parse_args()
    res = a.res
    sc = synth.scene_a(res, a.width, a.height)
    v = TSDFVolumeOctree()
    v.setResolution(res, res, res)
    v.setGridSize(sc.size, sc.size, sc.size)
    v.setImageSize(a.width, a.height)
    v.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
    v.setSensorDistanceBounds(0.0, 3 * sc.size)
    v.setIntegrateColor(bool(a.color))
    slab0, slab1 = 0, res
    if a.planes:
        slab0 = (res - a.planes) // 2
        slab1 = slab0 + a.planes
        v.setZSlab(slab0, slab1)
    if a.pairing:
        a.pipelined = 1
        v.setFramePairing(True)
    v.reset()
    groups = [(res // 2 - 1, res // 2 + 1), (res // 3, res // 3 + 2), (res - 300, res - 298)]
    if a.planes:
        groups = [(slab0, slab0 + 2), ((slab0 + slab1) // 2, (slab0 + slab1) // 2 + 2), (slab1 - 2, slab1)]
    oracles = [SlabOracle(v._p, zb, ze) for zb, ze in groups]
    t_gpu = t_cpu = t_synth = t_ray = 0.0
    fused_launches = 0
    info4 = None
    if a.pairing:
        import ctypes as C0
        info4 = (C0.c_int32 * 4)()
    mismatches = n_views = 0
    ray_err = []
    per_call = []
    for i in range(a.frames):
        t0 = time.perf_counter()
        tr = synth.turntable_pose(i, a.frames, sc.size, tilt=0.15 * np.sin(i * 0.05))
        dep, col = sc.depth(tr, noise_seed=12345 + i), sc.bgra(i)
        t1 = time.perf_counter()
        v.integrateCloud(dep, col if a.color else None, tr, pipelined=bool(a.pipelined))  # host entry point
        t2 = time.perf_counter()
        t_gpu += t2 - t1
        per_call.append((t2 - t1) * 1e3)
        if info4 is not None and i % 2 == 1:
            capi.check(capi.load().tsdf_hip_last_launch_info(v._need(), info4), "info")
            fused_launches += int(info4[0] == 2)
        if a.raycast_every and (i + 1) % a.raycast_every == 0:
            tq = time.perf_counter()
            view = v.renderView(tr, 1)  # camera frame: z is directly comparable with the noise-free depth image
            t_ray += time.perf_counter() - tq
            n_views += 1
            if i >= 8 and n_views % 10 == 0:
                clean = sc.depth(tr)
                both = np.isfinite(view[..., 2]) & np.isfinite(clean)
                ray_err.append(float(np.median(np.abs(view[..., 2][both] - clean[both]))))
            t2 = time.perf_counter()
        T = synth.cam_from_vol_f32(tr)
        for o in oracles:
            o.integrate(dep, col if a.color else None, T)
        t3 = time.perf_counter()
        t_synth += t1 - t0
        t_cpu += t3 - t2
        if a.check_every and (i + 1) % a.check_every == 0:
            for (zb, ze), o in zip(groups, oracles):
                d, w, rgb = v.download(z0=zb, nz=ze - zb)
                mismatches += int((d.view(np.uint32) != o.d.view(np.uint32)).sum() + (w != o.w).sum())
    tq = time.perf_counter()
    v.synchronize()
    t_gpu += time.perf_counter() - tq
    planes_equal = True
    saturated = 0.0
    for (zb, ze), o in zip(groups, oracles):
        d, w, rgb = v.download(z0=zb, nz=ze - zb)
        ok = np.array_equal(d.view(np.uint32), o.d.view(np.uint32)) and np.array_equal(w, o.w) and \
            (not a.color or np.array_equal(rgb, o.rgb))
        planes_equal &= bool(ok)
        saturated = max(saturated, float((w == v.getWeightTruncationLimit()).mean()))
    mc = MarchingCubesTSDFOctree()
    mc.setInputTSDF(v)
    mc.setMinWeight(2.0)
    mc.setColorByRGB(bool(a.color))
    t0 = time.perf_counter()
    lib = capi.load()
    import ctypes as C
    n = C.c_uint64(0)
    if not a.planes:  # a slab without its +z halo plane cannot mesh its last cell row; the multi-GPU layer does that
        capi.check(lib.tsdf_hip_march(v._need(), C.c_float(2.0), 1 if a.color else 0, C.byref(n)), "march")
    t_mc = time.perf_counter() - t0
    print(json.dumps({
        "workload": f"{res}^3 grid" + (f" (Z-slab [{slab0},{slab1}): one rank's share)" if a.planes else "") +
                    f", integrateColor={bool(a.color)}, {a.frames} distinct noisy {a.width}x{a.height} frames through the host entry "
                    "point (" + ("pinned two-slot ring, upload overlapped with the previous kernel" if a.pipelined else "PCIe upload + sync per frame") + ")" + ("" if a.planes else ", then marching cubes"),
        "frames": a.frames,
        **({"frame_pairing": True, "pairs_integrated_in_one_sweep": fused_launches} if a.pairing else {}),
        # synchronous calls: the host waits for upload + kernel, so host time per call IS the frame rate incl. upload;
        # asynchronous calls return once the frame is staged: host time per call says nothing about the GPU's rate
        # (the old name frames_per_s_incl_upload invited quoting 3267 frames/s) and is named for what it is
        **({"host_seconds_in_async_calls": t_gpu, "async_calls_issued_per_s_host_side": a.frames / t_gpu,
            "host_ms_per_async_call": t_gpu / a.frames * 1e3} if a.pipelined else
           {"gpu_seconds_incl_upload": t_gpu, "frames_per_s_incl_upload": a.frames / t_gpu,
            "ms_per_frame_incl_upload": t_gpu / a.frames * 1e3}),
        "marching_cubes_s": t_mc, "triangles": int(n.value),
        "oracle_plane_groups": groups, "planes_bit_identical_to_oracle": planes_equal,
        "intermediate_mismatches": mismatches, "fraction_of_checked_voxels_at_max_weight": saturated,
        "cpu_oracle_seconds_for_its_planes": t_cpu, "frame_synthesis_seconds": t_synth,
        "integrate_ms_median": float(np.median(per_call)),
        "integrate_ms_by_tenth_of_run": [round(float(np.median(c)), 2) for c in np.array_split(np.array(per_call), 10)],
        "integrate_ms_calls_20_to_70": [round(t, 1) for t in per_call[20:70]],
        "renderView_calls": n_views, "renderView_ms_incl_download": (t_ray / n_views * 1e3) if n_views else None,
        "renderView_median_abs_depth_error_m": (float(np.median(ray_err)) if ray_err else None)}))
    v.close()
    return 0 if planes_equal and not mismatches else 1


if __name__ == "__main__":
    sys.exit(main())

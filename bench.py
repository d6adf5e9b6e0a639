#!/usr/bin/env python3
"""bench.py -- integrateCloud throughput on MI355X (BASELINE.json metric).

One "step" = one integrateCloud pass of one synthetic 640x480 depth(+colour) frame over the whole voxel grid.
Workload = BASELINE.json configs[3]'s integrate leg, the configuration `metric` is quoted on: 2048^3 grid (8 m,
voxel 2^-8 m), integrateColor=true, Scene A turntable frames, frames resident in HBM before the
timed region.

N > 1 (one process per GPU, torch.distributed / RCCL): the SAME 2048^3 grid, Z-slab partitioned (strong scaling,
2048/N planes per GPU: the metric is "@2048^3 ... 1/2/4/8 GPUs").  The frame lives on rank 0; every step is one
RCCL broadcast of it (depth + colour in one 2.4 MB buffer) and one k_integrate launch per rank on its slab.  The
broadcast of frame i+1 is issued before the kernel of frame i is launched and lands in the other half of a
two-slot receive buffer, so it runs on RCCL's stream under that kernel.  `--config 4` is BASELINE configs[4]
(4096^3, 1280x960 frames, needs >= 4 GPUs); `--scaling weak` keeps --planes planes per GPU instead.

`python bench.py --gpus N` without a launcher re-executes itself under `python -m torch.distributed.run --nnodes=1
--nproc-per-node N` (one rank per GPU), so the plain command line measures the same thing as the launcher form.
`--host inprocess` times the OTHER host of the same partition instead: one process, one tsdf_hip_create_multi handle
over N GPUs (what cpu_tsdf::TSDFVolumeOctree::setDevices uses), frame fan-out by peer copies instead of RCCL.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant kernel
(k_integrate, HBM-bound) and `cpu_baseline` (the reference CPU path timed on this host on a bounded sample).

roofline.achieved uses the ALGORITHMIC bytes of the shipped HBM layout, measured per frame by the kernel's counting
instance: bytes of the voxel words an observed voxel must read (PACKED colour: d + colour|count word = 8 B) + bytes
of the words whose value changed (tsdf_hip_last_count_detail) + the frame: a per-voxel figure of the LAYOUT, the same
since round 2.  The kernel moves less than that where it can tell a voxel's distance from its observation count and does
not read it: `roofline.bytes_moved` prices exactly what it has to move (4 B less per such voxel,
tsdf_hip_last_read_detail) and is the figure that agrees with the PMC traffic.  SURVEY 8d's figure for the reference's
own (d, w, rgb) record, 24 B per observed voxel, stays as a labelled side note (`reference_record_*`): the PACKED
layout moves fewer bytes than that record holds, so a fraction computed from it can exceed 1 and means nothing.
roofline.traffic is the PMC measurement (rocprofv3 FETCH_SIZE / WRITE_SIZE over THIS command's timed launches,
tools/run_rocprof.sh -> profiles/pmc_traffic.json), reported only while the profile's kernel-source hash matches the
tree that is running.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
KERNEL_SOURCES = ("tsdf_integrate.hip", "tsdf_div.h", "tsdf_buffer.h", "tsdf_common.h")


def kernel_sha16():
    """Hash of the sources k_integrate is compiled from: stamps profiles so that a stale one is never quoted."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "cpu_tsdf_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", type=int, default=3, choices=[3, 4],
                    help="BASELINE.json configs[k]: 3 = 2048^3, 640x480 (default); 4 = 4096^3, 1280x960, Z-slabs over >= 4 GPUs")
    ap.add_argument("--res", type=int, default=0, help="x/y/z resolution (default: the config's)")
    ap.add_argument("--planes", type=int, default=0, help="z planes in total (strong) / per GPU (weak); 0 = --res")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--color", type=int, default=1)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong")
    ap.add_argument("--layout", choices=["auto", "f32w", "packed"], default="auto",
                    help="HBM weight layout (include/tsdf_hip.h TSDF_LAYOUT_*); auto = packed when max_weight <= 255")
    ap.add_argument("--frames", type=int, default=0, help="distinct frames on the turntable (default steps+warmup)")
    ap.add_argument("--overlap", type=int, default=1, help="N>1: broadcast frame i+1 under the kernel of frame i")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--extras", type=int, default=1, help="1: also time renderView + marching cubes once, k_integrate2 and Scene B "
                    "(N=1, outside the timed region); 2: only the k_integrate2 leg (A/B runs)")
    ap.add_argument("--scene-b", type=int, default=1, help="with --extras: the Scene-B (camera inside the volume) leg; the "
                    "rocprof run turns it off so that k_integrate's average is the headline workload's alone")
    ap.add_argument("--host", choices=["process", "inprocess"], default="process",
                    help="N>1: one process per GPU over RCCL (default) or ONE process driving all GPUs through tsdf_hip_create_multi")
    ap.add_argument("--keys", type=int, default=1, help="N=1 with --extras: also time the secondary integrate keys in this run -- the "
                    "headline grid in the SATURATED regime (w == max_weight), the colourless 2048^3 grid, one configs[4] slab "
                    "(4096x4096x512, 1280x960) -- each with kernel_ms, bytes moved, frac and its PMC quote (extras.keys)")
    ap.add_argument("--emulate-rank", type=int, default=-1, help="with --of N, on ONE GPU: integrate the Z-slab rank r of an N-rank strong-scaling "
                    "run would own ([r * planes / N, (r + 1) * planes / N) of the full grid) with the real turntable frames -- per-slab kernel "
                    "time and observed voxels for a PREDICTED scaling table (tools/predict_scaling.py); no number from it is a scaling measurement")
    ap.add_argument("--of", type=int, default=0, help="see --emulate-rank")
    ap.add_argument("--pairing", type=int, default=0, help="1: the timed frames go two per call (tsdf_hip_integrate_device2 -> k_integrate2: one "
                    "sweep of the slab per PAIR where both poses see all of it) and, at N > 1, two per collective (ONE broadcast of [A | B]); "
                    "a step is still one frame, --steps must be even.  The headline stays the single-frame path (default 0)")
    ap.add_argument("--presaturate", type=int, default=0, help="launches BEFORE the warm-up (cycling through the timed frames): the "
                    "timed region then runs in the saturated regime w == max_weight (>= 100 for the default max_weight): how "
                    "tools/run_rocprof.sh profiles that key")
    ap.add_argument("--host-path", type=int, default=1, help="N=1: also time the host-pointer entry point (report-only side field)")
    ap.add_argument("--principal-offset", type=float, default=0.0, help="evidence runs: move the principal point by this fraction "
                    "of the half-width and yaw every turntable camera so that the grid's centre still projects to the image "
                    "centre -- the regime where the reference's frustum cull (1.1 x FOV about the optical AXIS) decides voxels "
                    "and the launch carries it in row intervals.  Not the headline configuration: the line says so")
    ap.add_argument("--dry-run-ranks", type=int, default=0, help="run the FULL N-rank control flow (rendezvous, Z-slab partition, per-step "
                    "frame broadcast overlapped with the previous kernel, all-reduce of the results) with N ranks on ONE GPU over gloo "
                    "(RCCL refuses two ranks per device): what an N-GPU node executes, minus the links.  Timings of such a run say "
                    "nothing about scaling; host_us_per_step (the host-side cost of one step) is what it is for")
    ap.add_argument("--calib", type=int, default=0, help="run this many k_calib_rmw sweeps of exactly known bytes first "
                    "(PMC passes: calibrates FETCH_SIZE / WRITE_SIZE in the same process)")
    a = ap.parse_args()
    if a.config == 4:
        a.res, a.width, a.height = a.res or 4096, a.width or 1280, a.height or 960
    else:
        a.res, a.width, a.height = a.res or 2048, a.width or 640, a.height or 480
    return a


def cpu_baseline(args, sc, res3, size3, budget_s):
    """Reference CPU path on this host, bounded sample.  Rank 0, N=1 only.  Uses oracle/_ref (the
    reference's own sources compiled against the in-repo PCL/Eigen stand-ins, native adaptive octree
    mode, OpenMP) when it was prebuilt; otherwise the dense C restatement on a Z-slab sample."""
    from cpu_tsdf_amd import synth
    cores = os.cpu_count() or 1
    try:
        from oracle import refbind
        if refbind.available():
            return refbind.time_integrate(sc, res3, size3, bool(args.color), budget_s, cores)
    except ImportError:
        pass
    from cpu_tsdf_amd import capi
    from oracle.oracle import SlabOracle
    p = capi.default_params()
    planes = max(8, min(res3[2], (64 * 2048 * 2048) // (res3[0] * res3[1])))
    zb = (res3[2] - planes) // 2
    # a slab-sized dense oracle: same x/y resolution, `planes` z planes re-centred on the slab
    p.res[:] = (res3[0], res3[1], res3[2])
    p.size[:] = size3
    p.fx, p.fy, p.cx, p.cy = sc.fx, sc.fy, sc.cx, sc.cy
    p.image_width, p.image_height = sc.width, sc.height
    p.min_sensor_dist, p.max_sensor_dist = 0.0, 3 * max(size3)
    p.integrate_color = args.color
    ov = SlabOracle(p, zb, zb + planes)
    t0 = time.perf_counter()
    n = 0
    while True:
        tr = synth.turntable_pose(n, 8, sc.size)
        ov.integrate(sc.depth(tr), sc.bgra(n) if args.color else None, synth.cam_from_vol_f32(tr))
        n += 1
        if time.perf_counter() - t0 > budget_s / 2 or n >= 4:
            break
    dt = (time.perf_counter() - t0) / n
    full = dt * res3[2] / planes
    vox = res3[0] * res3[1] * res3[2]
    return {"value": vox / full / 1e6, "unit": "Mvoxels/s", "frames_per_s": 1.0 / full, "cores": cores,
            "kind": "port",
            "sample": f"dense C restatement (OpenMP, {cores} threads), planes [{zb},{zb + planes}) of the "
                      f"{res3[0]}x{res3[1]}x{res3[2]} grid, {n} frames, extrapolated x{res3[2] / planes:.0f}"}


def extras(vol, pose, W, H):
    """Report-only timings of the other two kernels of the path on the fused volume (outside the timed
    region; gather/latency-bound raycast, streaming marching cubes): not part of `value`."""
    from cpu_tsdf_amd import capi
    out = {}
    try:
        vol.renderView(pose, 1, camera_frame=False)  # warm-up (scratch allocation)
        t0 = time.perf_counter()
        img = vol.renderView(pose, 1, camera_frame=False)
        dt_pageable = time.perf_counter() - t0
        # the same into pinned caller memory (tsdf_hip_host_alloc): detected, DMA straight into it, no bounce copy
        vol.renderView(pose, 1, camera_frame=False, pinned=True)
        t0 = time.perf_counter()
        img = vol.renderView(pose, 1, camera_frame=False, pinned=True)
        dt = time.perf_counter() - t0
        out["renderView_ms"] = dt * 1e3
        out["renderView_ms_pageable_destination"] = dt_pageable * 1e3
        out["renderView_rays_per_s"] = W * H / dt
        out["renderView_hits"] = int(np.isfinite(img[..., 0]).sum())
        out["renderView_mean_steps"] = float(img[..., 7].mean())
        out["renderView_steps_per_s"] = float(img[..., 7].sum()) / dt
        lib = capi.load()
        n = C.c_uint64(0)
        lib.tsdf_hip_march(vol._need(), C.c_float(1.0), 1, C.byref(n))  # warm-up (buffer growth)
        t0 = time.perf_counter()
        rc = lib.tsdf_hip_march(vol._need(), C.c_float(1.0), 1, C.byref(n))
        dt = time.perf_counter() - t0
        if rc == 0:
            rx, ry, rz = vol.getResolution()
            vox = rx * ry * float(rz)
            ms = (C.c_float * 3)()
            cells = C.c_uint64(0)
            lib.tsdf_hip_march_timing(vol._need(), ms, C.byref(cells))
            color = bool(vol._p.integrate_color)
            out["reconstruct_ms"] = dt * 1e3
            out["reconstruct_triangles"] = int(n.value)
            out["reconstruct_active_cells"] = int(cells.value)
            out["reconstruct_Mvoxels_per_s"] = vox / dt / 1e6
            out["reconstruct_phase_ms"] = {"classify": ms[0], "sort_scan": ms[1], "emit": ms[2]}
            # PHYSICAL bytes (what the kernels must move in the shipped layout), not SURVEY's 8/12 B record:
            #  classify reads the distance quads the band flags of integrateCloud leave it (tsdf_hip_march_stats:
            #  requested bytes, counted on the device; the whole plane when the flags are not in use) and gathers 8
            #  weight words per listed cell; emit reads 8 corners (d + weight word) per active cell and writes 36 B of
            #  vertices + 9 B of colour + 8 B of cell key per triangle; 16 B (key, cell) per active cell go out of
            #  classify and through the sort.
            st = (C.c_uint64 * 4)()
            lib.tsdf_hip_march_stats(vol._need(), st)
            elided = bool(st[3] & 2)  # no count of a listed cell's corners could fail the weight test: not gathered
            classify_bytes = float(st[2]) + ((0.0 if elided else 32.0) + 16.0) * cells.value
            emit_bytes = 64.0 * cells.value + (36.0 + (9.0 if color else 0.0) + 8.0) * n.value
            out["reconstruct_classify_bytes"] = classify_bytes
            out["reconstruct_classify_d_bytes_requested"] = int(st[2])
            out["reconstruct_classify_d_plane_bytes"] = 4.0 * vox
            out["reconstruct_classify_skips_unobserved_space"] = bool(st[3] & 1)
            out["reconstruct_classify_weight_test_elided"] = elided
            out["reconstruct_classify_GBps"] = classify_bytes / (ms[0] * 1e-3) / 1e9 if ms[0] > 0 else None
            out["reconstruct_classify_frac_of_hbm_peak"] = (classify_bytes / (ms[0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms[0] > 0 else None
            out["reconstruct_emit_GBps"] = emit_bytes / (ms[2] * 1e-3) / 1e9 if ms[2] > 0 else None
    except Exception as e:  # never let a report-only leg break the bench line
        out["error"] = repr(e)
    return out


def scene_b_leg(res, color, cpu_seconds):
    """Report-only honesty check (Scene B): the regime the reference's octree was built for --
    camera inside a 10 m volume, sensor range 0..3 m, ~1 % of the voxels in the frustum.  GPU: brick cull +
    k_integrate on a second res^3 grid (frames resident in HBM, HIP events); CPU: the reference itself on the
    same poses, bounded to `cpu_seconds` of integrateCloud time."""
    import torch
    from cpu_tsdf_amd import capi, synth
    from cpu_tsdf_amd.volume import TSDFVolumeOctree
    out = {}
    try:
        sc = synth.scene_b()
        v = TSDFVolumeOctree()
        v.setResolution(res, res, res)
        v.setGridSize(10.0, 10.0, 10.0)
        v.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
        v.setSensorDistanceBounds(0.0, 3.0)
        v.setIntegrateColor(bool(color))
        stream = torch.cuda.current_stream()
        v.setStream(stream.cuda_stream)
        v.reset()
        nf = 12
        poses = [synth.scene_b_pose(i, nf) for i in range(nf)]
        frame = torch.empty((nf, 2, sc.height, sc.width), dtype=torch.float32, device="cuda")  # [depth | bgra] per frame
        for i, p in enumerate(poses):
            frame[i, 0].copy_(torch.from_numpy(sc.depth(p)))
            frame[i, 1].view(torch.uint8).view(sc.height, sc.width, 4).copy_(torch.from_numpy(sc.bgra(i)))
        lib, h = capi.load(), v._need()

        from cpu_tsdf_amd.volume import reference_cull_planes
        planes = [reference_cull_planes(v._p, p) for p in poses]  # the shells' default: the reference's cull, per frame

        def run(i, count=None):
            capi.check(lib.tsdf_hip_set_reference_cull(h, capi.as_f32p(planes[i])), "set_reference_cull")
            capi.check(lib.tsdf_hip_integrate_device(h, C.c_void_p(frame[i, 0].data_ptr()),
                                                     C.c_void_p(frame[i, 1].data_ptr()) if color else None,
                                                     capi.as_f32p(synth.cam_from_vol_f32(poses[i])), count), "scene_b")
        c = C.c_uint64(0)
        run(0, C.byref(c))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(nf):
            run(i)
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / nf
        info = (C.c_int32 * 4)()
        capi.check(lib.tsdf_hip_last_launch_info(h, info), "last_launch_info")
        detail, rdet = (C.c_uint64 * 2)(), (C.c_uint64 * 3)()
        run(nf // 2, C.byref(c))
        capi.check(lib.tsdf_hip_last_count_detail(h, detail), "last_count_detail")
        capi.check(lib.tsdf_hip_last_read_detail(h, rdet), "last_read_detail")
        packed = v.getLayout() == capi.LAYOUT_PACKED
        read_bpv = ((8 if color else 5) if packed else (12 if color else 8))
        alg = read_bpv * int(detail[0]) + int(detail[1]) + (8 if color else 4) * sc.width * sc.height
        out.update({"grid": [res] * 3, "size_m": 10.0, "sensor_range_m": [0.0, 3.0], "observed_voxels_per_frame": int(c.value),
                    "gpu_ms_per_frame": ms, "gpu_frames_per_s": 1e3 / ms,
                    "launch": {"row_intervals": int(info[2]), "reference_cull_in_intervals": bool(info[2] == 2), "blocks": int(info[3])},
                    "algorithmic_bytes_per_frame": alg, "distance_words_not_read": int(rdet[0]),
                    "frac_of_hbm_peak": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "plane_bytes_requested": int(rdet[2]),
                    "frac_of_hbm_peak_by_bytes_moved": (int(rdet[2]) + int(detail[1]) + (8 if color else 4) * sc.width * sc.height) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "note": "ms per frame = rows + flags + k_integrate (HIP events around 12 frames); algorithmic bytes as in roofline"})
        v.close()
        if cpu_seconds > 0:
            from oracle import refbind
            if refbind.available():
                cores = os.cpu_count() or 1
                os.environ.setdefault("OMP_NUM_THREADS", str(cores))
                rv = refbind.RefVolume(res, 10.0, sc.width, sc.height, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3.0, color=bool(color),
                                       dense=False, max_cell=0.5)
                spent, n = 0.0, 0
                while spent < cpu_seconds and n < nf:
                    spent += rv.integrate(sc.depth(poses[n]), sc.bgra(n) if color else None, poses[n])
                    n += 1
                rv.close()
                out.update({"cpu_reference_frames_per_s": n / spent, "cpu_cores": cores, "cpu_frames_timed": n})
    except Exception as e:
        out["error"] = repr(e)
    return out


def key_leg(res3, W, H, color, steps=10, warm=2, presaturate=0, slab_of=None):
    """One SECONDARY bench key in the driver-run line: its own volume (freed before returning),
    `warm + steps` Scene-A turntable frames resident in HBM, the warm-up through the counting instance, HIP events on the
    kernel's stream around each of `steps` timed launches, then the same frames once more through the counting instance for
    the bytes the kernel moves (plane bytes requested + changed words + frame) -- the headline's own definition of
    roofline.achieved / frac -- and the PMC quote of profiles/pmc_traffic.json for the key while the kernel-source hash
    matches.  presaturate: launches before the warm-up (the saturated regime: w == max_weight, octree.cpp:157-159)."""
    import torch
    from cpu_tsdf_amd import capi, synth
    from cpu_tsdf_amd.volume import TSDFVolumeOctree, reference_cull_planes
    out = {}
    v = None
    try:
        voxel = 2.0 ** -8
        size3 = tuple(r * voxel for r in res3)
        S = size3[0]
        sc = synth.Scene(S, W, H)
        if res3[2] != res3[0]:
            sc.h = np.array([0.47 * size3[0], 0.47 * size3[1], 0.47 * size3[2]])
        v = TSDFVolumeOctree()
        v.setResolution(*res3)
        v.setGridSize(*size3)
        v.setImageSize(W, H)
        v.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
        v.setSensorDistanceBounds(0.0, 3.0 * max(size3))
        v.setDepthTruncationLimits(0.03, 0.03)
        v.setIntegrateColor(bool(color))
        stream = torch.cuda.current_stream()
        v.setStream(stream.cuda_stream)
        v.reset()
        lib, h = capi.load(), v._need()
        n = warm + steps
        radius = 2.2 * max(size3) / S
        poses = [synth.turntable_pose(i, n, S, radius_factor=radius) for i in range(n)]
        T = [synth.cam_from_vol_f32(p) for p in poses]
        planes = [reference_cull_planes(v._p, p) for p in poses]
        fr = torch.empty((n, 2 if color else 1, H, W), dtype=torch.float32, device="cuda")
        for i, p in enumerate(poses):
            fr[i, 0].copy_(torch.from_numpy(sc.depth(p)))
            if color:
                fr[i, 1].view(torch.uint8).view(H, W, 4).copy_(torch.from_numpy(sc.bgra(i)))

        def run(i, count=None):
            capi.check(lib.tsdf_hip_set_reference_cull(h, capi.as_f32p(planes[i])), "set_reference_cull")
            capi.check(lib.tsdf_hip_integrate_device(h, C.c_void_p(fr[i, 0].data_ptr()), C.c_void_p(fr[i, 1].data_ptr()) if color else None,
                                                     capi.as_f32p(T[i]), count), "key_leg")
        for k in range(presaturate):
            run(warm + k % steps)
        c = C.c_uint64(0)
        for i in range(warm):
            run(i, C.byref(c))
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for k in range(steps):
            ev[k][0].record(stream)
            run(warm + k)
            ev[k][1].record(stream)
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in ev) / steps
        info = (C.c_int32 * 4)()
        capi.check(lib.tsdf_hip_last_launch_info(h, info), "last_launch_info")
        detail, rdet = (C.c_uint64 * 2)(), (C.c_uint64 * 3)()
        n_obs = chg = req = imp = 0
        for k in range(steps):
            run(warm + k, C.byref(c))
            capi.check(lib.tsdf_hip_last_count_detail(h, detail), "last_count_detail")
            capi.check(lib.tsdf_hip_last_read_detail(h, rdet), "last_read_detail")
            n_obs += int(detail[0]); chg += int(detail[1]); imp += int(rdet[0]); req += int(rdet[2])
        n_obs, chg, imp, req = n_obs / steps, chg / steps, imp / steps, req / steps
        packed = v.getLayout() == capi.LAYOUT_PACKED
        bpp = 8 if color else 4
        read_bpv = ((8 if color else 5) if packed else (12 if color else 8))
        moved = (req if req else read_bpv * n_obs - 4 * imp) + chg + bpp * W * H
        sha = kernel_sha16()
        key = f"{res3[0]}x{res3[1]}x{res3[2]}_c{int(bool(color))}_{'packed' if packed else 'f32w'}" + ("_saturated" if presaturate else "")
        prof, why = pmc_traffic(key, sha)
        out = {"key": key, "grid": list(res3), "image": [W, H], "color": bool(color), "layout": "packed" if packed else "f32w",
               "presaturate_launches": presaturate, "steps": steps,
               "instance": {0: "general", 1: "ALLIN", 2: "k_integrate2"}.get(int(info[0]) & 0xff, "?") + (" + row intervals" if info[2] else "") +
                           (", software-pipelined row loop (k_integrate_p)" if int(info[0]) & 0x100 else ""),
               "kernel_ms": ms, "frames_per_s": 1e3 / ms, "Mvoxels_per_s": float(res3[0]) * res3[1] * res3[2] / (ms * 1e-3) / 1e6,
               "observed_voxels_per_frame": n_obs, "distance_words_not_read": imp, "plane_bytes_requested": req,
               "changed_word_bytes": chg, "bytes_moved_per_launch": moved, "GBps": moved / (ms * 1e-3) / 1e9,
               "frac": moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "traffic": prof["hbm_bytes_per_launch"] if prof else None,
               "frac_check": ({"pmc_bytes_per_launch": prof["hbm_bytes_per_launch"], "kernel_counted_bytes_per_launch": moved,
                               "relative_difference": abs(prof["hbm_bytes_per_launch"] - moved) / prof["hbm_bytes_per_launch"],
                               "agree_within_5_percent": abs(prof["hbm_bytes_per_launch"] - moved) / prof["hbm_bytes_per_launch"] < 0.05,
                               "profile_tag": prof.get("tag"), "kernel_ms_in_profile": prof.get("kernel_ms_in_profile")} if prof else {"note": why})}
        if slab_of:
            out["note"] = slab_of
    except Exception as e:  # a report-only leg never breaks the bench line
        out["error"] = repr(e)
    finally:
        if v is not None:
            v.close()
        try:
            import torch as _t
            _t.cuda.synchronize()
            _t.cuda.empty_cache()
        except Exception:
            pass
    return out


def fused2_leg(lib, h, frames_dev, T_all, planes_all, args, stream, W, H, packed):
    """extras.fused2: frames [warmup, warmup + steps) integrated two per kernel sweep (k_integrate2).  ms per FRAME from
    one HIP event pair around all launches; algorithmic bytes per LAUNCH from the counting instance of the same kernel
    (voxel words a voxel observed by either frame must read, once + words whose value changed over the pair + two
    frames)."""
    import torch
    from cpu_tsdf_amd import capi
    first, n_pairs = args.warmup, args.steps // 2

    def pair(k, count=None, fused=None):
        i, j = first + 2 * k, first + 2 * k + 1
        fa, fb = frames_dev[i], frames_dev[j]
        capi.check(lib.tsdf_hip_integrate_device2(
            h, C.c_void_p(fa[0].data_ptr()), C.c_void_p(fa[1].data_ptr()) if args.color else None, capi.as_f32p(T_all[i]),
            capi.as_f32p(planes_all[i]),
            C.c_void_p(fb[0].data_ptr()), C.c_void_p(fb[1].data_ptr()) if args.color else None, capi.as_f32p(T_all[j]),
            capi.as_f32p(planes_all[j]), count, fused), "integrate_device2")
    was = C.c_int32(0)
    pair(0, None, C.byref(was))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for k in range(n_pairs):
        pair(k)
    e1.record(stream)
    torch.cuda.synchronize()
    ms_launch = e0.elapsed_time(e1) / n_pairs
    detail, n2, rdet = (C.c_uint64 * 2)(), (C.c_uint64 * 2)(), (C.c_uint64 * 3)()
    obs = chg = per_frame = imp = req = 0
    for k in range(n_pairs):
        pair(k, n2)
        capi.check(lib.tsdf_hip_last_count_detail(h, detail), "last_count_detail")
        capi.check(lib.tsdf_hip_last_read_detail(h, rdet), "last_read_detail")
        obs, chg, per_frame = obs + int(detail[0]), chg + int(detail[1]), per_frame + int(n2[0]) + int(n2[1])
        imp += int(rdet[0])
        req += int(rdet[2])
    obs, chg, imp, req = obs / n_pairs, chg / n_pairs, imp / n_pairs, req / n_pairs
    read_bpv = ((8 if args.color else 5) if packed else (12 if args.color else 8))
    alg = read_bpv * obs + chg + 2 * (8 if args.color else 4) * W * H
    return {"one_sweep_per_pair": bool(was.value), "pairs_timed": n_pairs, "ms_per_launch": ms_launch, "ms_per_frame": ms_launch / 2,
            "frames_per_s": 2e3 / ms_launch, "kernel": "k_integrate2" if was.value else "k_integrate x 2",
            "algorithmic_bytes_per_launch": alg, "voxels_observed_by_either_frame": obs,
            "voxels_observed_per_frame": per_frame / (2 * n_pairs), "changed_word_bytes_per_launch": chg,
            "distance_words_not_read_per_launch": imp,
            "plane_bytes_requested_per_launch": req,
            "bytes_moved_per_launch": req + chg + 2 * (8 if args.color else 4) * W * H,
            "frac_of_hbm_peak_by_bytes_moved": (req + chg + 2 * (8 if args.color else 4) * W * H) / (ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "achieved_GBps": alg / (ms_launch * 1e-3) / 1e9, "frac_of_hbm_peak": alg / (ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "report-only: tsdf_hip_integrate_device2 reads and writes each voxel word once per PAIR of frames; the planes "
                    "are bit-identical to frame-by-frame integration (tests/test_fused2_gpu.py); the headline `value` is the "
                    "single-frame kernel"}


def host_path_leg(vol, sc, poses, color, first, last):
    """Report-only: the reference's integrateCloud takes a HOST cloud, so this is the PCIe-inclusive rate of the same
    workload -- the SAME frames the timed region integrated (poses[first:last]), handed over as host pointers, back to
    back, through tsdf_hip_integrate (upload + kernel + synchronise per call) and through tsdf_hip_integrate_async
    (pinned two-slot ring, upload under the previous kernel).  Never `value`: the headline is timed with frames
    resident in HBM."""
    out = {}
    try:
        idx = list(range(first, last))
        frames = [(np.ascontiguousarray(sc.depth(poses[i])), np.ascontiguousarray(sc.bgra(i)) if color else None) for i in idx]
        for name, pipelined, pairing in (("frames_per_s_sync_calls", False, False), ("frames_per_s_async_ring", True, False),
                                         ("frames_per_s_async_ring_frame_pairing", True, True)):
            vol.setFramePairing(pairing)  # (tsdf_hip_set_frame_pairing: two queued frames share one sweep of the volume)
            vol.integrateCloud(frames[0][0], frames[0][1], poses[idx[0]], pipelined=pipelined)
            vol.synchronize()
            t0 = time.perf_counter()
            for (d, c), i in zip(frames, idx):
                vol.integrateCloud(d, c, poses[i], pipelined=pipelined)
            vol.synchronize()
            out[name] = len(idx) / (time.perf_counter() - t0)
        vol.setFramePairing(False)
        out["calls"] = len(idx)
        out["note"] = ("host-pointer entry points, the timed region's own frames, one 2.4 MB-class frame per call over PCIe; "
                       "report-only, the headline `value` is timed with frames resident in HBM.  This leg runs AFTER the timed "
                       "region on the further-fused volume (observation counts nearer saturation: fewer changed words to "
                       "store), so its rate can exceed frames_per_s")
    except Exception as e:  # never let a report-only leg break the bench line
        out["error"] = repr(e)
    return out


def cpp_dropin_leg(sc, poses, res, W, H, color, n_frames=40, distinct=8):
    """Report-only: what a C++ user of the drop-in waits for -- cpu_tsdf::TSDFVolumeOctree::integrateCloud
    on pcl::PointCloud<pcl::PointXYZRGBA> clouds in host memory (AoS strip into the pinned slot + upload + kernel), through
    the product's own timing program cpu_tsdf_amd/bin/dropin_rate (csrc/prog/dropin_rate.cpp), a process of its own with its
    own res^3 volume, frame pairing off and on, `distinct` of this run's Scene-A frames cycled through `n_frames` calls."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "cpu_tsdf_amd", "bin", "dropin_rate")
    if not os.path.exists(exe):
        return {"error": "cpu_tsdf_amd/bin/dropin_rate is not built (python __graft_entry__.py)"}
    out = {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            path = os.path.join(td, "frames.bin")
            with open(path, "wb") as f:
                for i in range(min(distinct, len(poses))):
                    f.write(np.ascontiguousarray(poses[i], dtype=np.float64).tobytes())
                    f.write(np.ascontiguousarray(sc.depth(poses[i]), dtype=np.float32).tobytes())
                    f.write(np.ascontiguousarray(sc.bgra(i), dtype=np.uint8).tobytes())
            for pairing in (0, 1):
                p = subprocess.run([exe, str(res), str(W), str(H), str(int(bool(color))), str(pairing), str(n_frames), path],
                                   capture_output=True, text=True, timeout=180)
                if p.returncode:
                    out["pairing_on" if pairing else "pairing_off"] = {"error": f"exit {p.returncode}: {p.stderr.strip()[-300:]}"}
                else:
                    out["pairing_on" if pairing else "pairing_off"] = json.loads(p.stdout.strip().splitlines()[-1])
        out["note"] = ("TSDFVolumeOctree::integrateCloud(PointCloud<PointXYZRGBA>) called back to back from C++ on host clouds: the "
                       "in-call time is what the caller's thread pays (strip + queue), the sustained rate includes a final call that "
                       "waits for the device; setSynchronous off")
    except Exception as e:
        out["error"] = repr(e)
    return out


def pmc_traffic(key, sha):
    """HBM bytes per timed k_integrate launch from the committed PMC profile of this very command -- only if the
    profile was taken on the kernel sources that are running now."""
    prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        e = json.load(open(prof)).get(key)
    except Exception:
        return None, "no profiles/pmc_traffic.json"
    if not e:
        return None, f"no PMC profile for {key}"
    if e.get("kernel_sha16") != sha:
        return None, f"PMC profile {e.get('tag')} was taken on other kernel sources ({e.get('kernel_sha16')} != {sha}): not quoted"
    return e, None


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` as the driver may call it: become `python -m torch.distributed.run ... bench.py ...`
    with one rank per GPU (the same command line the driver uses when it launches the ranks itself)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    args = parse()
    if args.dry_run_ranks:
        args.gpus = args.dry_run_ranks
        os.environ["TSDF_BENCH_ONE_DEVICE"] = "1"
        os.environ["TSDF_BENCH_BACKEND"] = "gloo"
    if args.gpus > 1 and args.host == "process" and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args.gpus)  # does not return
    if args.host == "inprocess":
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise SystemExit("--host inprocess is ONE process driving all GPUs: do not launch it under torch.distributed.run")
        return main_inprocess(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    # Test hooks for boxes with ONE GPU (gpurun): TSDF_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # TSDF_BENCH_BACKEND=gloo moves the collectives off RCCL (which refuses two ranks per device), so the N>1
    # code path can be smoke-run there; TSDF_BENCH_FORCE_DIST=1 takes the collective path even at world size 1
    # (one rank on RCCL: communicator init + device-tensor broadcast / all-reduce really run).  Numbers from such
    # runs mean nothing; the driver never sets these.
    if os.environ.get("TSDF_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("TSDF_BENCH_BACKEND", "nccl")
    use_dist = world > 1 or os.environ.get("TSDF_BENCH_FORCE_DIST") == "1"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from cpu_tsdf_amd import capi, synth
    from cpu_tsdf_amd.volume import TSDFVolumeOctree
    if args.calib:
        capi.use_test_library()  # the calibration sweeps of exactly known bytes are test hooks (include/tsdf_hip_test.h): PMC runs only
    capi.load()  # raises if the HIP library is not built: there is no fallback

    res = args.res
    voxel = 2.0 ** -8
    planes = args.planes or res
    if args.scaling == "weak":
        res3 = (res, res, planes * world)
        z_begin, z_end = rank * planes, (rank + 1) * planes
    else:
        res3 = (res, res, planes)
        per = planes // world
        z_begin, z_end = rank * per, (rank + 1) * per if rank < world - 1 else planes
    emulated = None
    if args.of and args.emulate_rank >= 0:
        if world != 1 or args.scaling != "strong" or not 0 <= args.emulate_rank < args.of:
            raise SystemExit("--emulate-rank r --of N: one process, strong scaling, 0 <= r < N")
        per = planes // args.of
        z_begin, z_end = args.emulate_rank * per, ((args.emulate_rank + 1) * per if args.emulate_rank < args.of - 1 else planes)
        emulated = {"rank": args.emulate_rank, "of": args.of, "z_begin": z_begin, "z_end": z_end,
                    "note": "ONE GPU integrating the slab this rank of an N-rank run would own: input to a PREDICTED scaling table, not a scaling measurement"}
        args.extras = 0
        args.host_path = 0
        args.cpu_baseline = 0
    size3 = tuple(r * voxel for r in res3)
    S = size3[0]
    W, H = args.width, args.height
    bytes_per_voxel = (12 if args.color else 8) if args.layout == "f32w" else (8 if args.color else 5)
    slab_gb = (z_end - z_begin) * res * res * bytes_per_voxel / 2 ** 30
    free_gb_before = torch.cuda.mem_get_info(dev)[0] / 2 ** 30
    # preflight: the slab's planes (configs[4]: 512 planes of 4096^2 = 64 GiB PACKED, 103 GB with float weights) must fit THIS
    # rank's GPU next to what is already on it -- said here, in plain words, instead of as a hipMalloc failure inside create
    if slab_gb > 250 or slab_gb * 1.02 + 1.0 > free_gb_before:
        raise SystemExit(f"rank {rank}: a {z_end - z_begin}-plane slab of a {res}x{res} grid needs {slab_gb:.0f} GiB of HBM, "
                         f"{free_gb_before:.0f} GiB are free on cuda:{local_rank}: use more GPUs (configs[4] needs >= 4) or fewer --planes")
    sc = synth.Scene(S, W, H)  # sphere + far-face box scaled to the x/y extent
    if res3[2] != res3[0]:
        sc.h = np.array([0.47 * size3[0], 0.47 * size3[1], 0.47 * size3[2]])

    if args.principal_offset:
        sc.cx += args.principal_offset * (W / 2)
    vol = TSDFVolumeOctree()
    vol.setResolution(*res3)
    vol.setGridSize(*size3)
    vol.setImageSize(W, H)
    vol.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
    vol.setSensorDistanceBounds(0.0, 3.0 * max(size3))  # CLI default min 0 (integrate.cpp:333)
    vol.setDepthTruncationLimits(0.03, 0.03)
    vol.setIntegrateColor(bool(args.color))
    vol.setLayout({"auto": capi.LAYOUT_AUTO, "f32w": capi.LAYOUT_F32W, "packed": capi.LAYOUT_PACKED}[args.layout])
    vol.setZSlab(z_begin, z_end, 0, local_rank)
    stream = torch.cuda.current_stream(dev)
    vol.setStream(stream.cuda_stream)
    vol.reset()
    lib = capi.load()
    h = vol._need()
    probe_ms, chosen = (C.c_float * 8)(), C.c_int32(0)
    n_tried = lib.tsdf_hip_alloc_probe(h, probe_ms, C.byref(chosen))
    placement = {"candidates_tried": int(n_tried), "probe_sweep_ms": [round(float(x), 3) for x in probe_ms[:max(1, n_tried)]],
                 "kept": int(chosen.value)}

    calibration = None
    if args.calib:
        br, bw = C.c_uint64(), C.c_uint64()
        for _ in range(args.calib):
            capi.check(lib.tsdf_hip_selftest_sweep(h, C.byref(br), C.byref(bw)), "sweep")
        calibration = {"kernel": "k_calib_rmw", "launches": args.calib, "known_read_bytes": br.value,
                       "known_written_bytes": bw.value, "narrow_reads": []}
        # ... and what FETCH_SIZE tallies for narrow reads (one dword per 4 / 64 / 128 bytes of the distance plane)
        for stride in (4, 64, 128):
            span, words = C.c_uint64(), C.c_uint64()
            capi.check(lib.tsdf_hip_selftest_read_sweep(h, stride, C.byref(span), C.byref(words)), "read_sweep")
            calibration["narrow_reads"].append({"kernel": f"k_calib_read<{stride}>", "stride_bytes": stride,
                                                "span_bytes": span.value, "dwords_read": words.value})

    # ---- synthetic frames, resident in HBM before the timed region ---------------------------------
    if args.pairing and (args.steps % 2 or args.warmup % 2):
        raise SystemExit("--pairing 1: --steps and --warmup must be even (frames go two per call; warm-up frames go singly)")
    n_total = args.warmup + args.steps
    n_distinct = args.frames or n_total
    radius = 2.2 * max(size3) / S
    poses = [synth.turntable_pose(i, n_distinct, S, radius_factor=radius) for i in range(n_total)]
    if args.principal_offset:
        psi = float(np.arctan(args.principal_offset * (W / 2) / sc.fx))
        yaw = np.eye(4)
        yaw[0, 0], yaw[0, 2], yaw[2, 0], yaw[2, 2] = np.cos(psi), np.sin(psi), -np.sin(psi), np.cos(psi)  # about the camera's y
        poses = [p @ yaw for p in poses]
    T_all = [synth.cam_from_vol_f32(p) for p in poses]
    # what cpu_tsdf::TSDFVolumeOctree::integrateCloud hands over per frame besides the pose: the six planes of the
    # reference's frustum cull (tsdf_hip_set_reference_cull; for this workload they keep every voxel: checked per launch)
    from cpu_tsdf_amd.volume import reference_cull_planes
    planes_all = [reference_cull_planes(vol._p, p) for p in poses]
    # one allocation per frame: [depth | bgra] back to back, which is what the kernel's single frame
    # descriptor wants (no staging copy) and what ONE broadcast per frame can carry
    fplanes = 2 if args.color else 1
    frames_dev = torch.empty((n_total if rank == 0 else 1, fplanes, H, W), dtype=torch.float32, device=dev)
    if rank == 0:
        for i, p in enumerate(poses):
            frames_dev[i, 0].copy_(torch.from_numpy(sc.depth(p)))
            if args.color:
                frames_dev[i, 1].view(torch.uint8).view(H, W, 4).copy_(torch.from_numpy(sc.bgra(i)))
    # two receive slots: the broadcast of frame i+1 fills one while k_integrate reads frame i from the other
    recv = torch.empty((4 if args.pairing else 2, fplanes, H, W), dtype=torch.float32, device=dev) if use_dist else None
    pairs = []  # HIP event pairs around each timed launch, on the stream the kernel runs on
    ev_pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    host_t = {"launch": 0.0, "broadcast": 0.0, "events": 0.0, "steps": 0}  # host seconds spent ENQUEUEING, timed steps only
    # everything a step hands to the C ABI is bound once, outside the timed region: per step the host makes two ctypes calls
    # (planes, integrate) with ready-made pointer objects
    set_cull, integ = lib.tsdf_hip_set_reference_cull, lib.tsdf_hip_integrate_device
    bound = {}

    def bound_args(i):
        b = bound.get(i)
        if b is None:
            fr = frame_buf(i)
            b = bound[i] = (capi.as_f32p(planes_all[i]), C.c_void_p(fr[0].data_ptr()), C.c_void_p(fr[1].data_ptr()) if args.color else None,
                            capi.as_f32p(T_all[i]))
        return b

    def frame_buf(i):
        return frames_dev[i] if rank == 0 else recv[i & (3 if args.pairing else 1)]

    def pair_buf(i):  # frames i, i + 1 back to back (warm-up even, so i is): what ONE broadcast carries with --pairing
        if rank == 0:
            return frames_dev[i:i + 2]
        k = i & 3
        return recv[k:k + 2]

    integ2 = lib.tsdf_hip_integrate_device2
    fused_pairs = [0, 0]  # pairs launched, pairs every slab swept once

    def launch_pair(i, timed=False):
        (pa, da, ca, ta), (pb, db, cb, tb) = bound_args(i), bound_args(i + 1)
        if timed:
            pairs.append(ev_pool[len(pairs)])
            pairs[-1][0].record(stream)
        f = C.c_int32(0)
        rc = integ2(h, da, ca, ta, pa, db, cb, tb, pb, None, C.byref(f))
        if timed:
            pairs[-1][1].record(stream)
            host_t["steps"] += 2
        if rc:
            capi.check(rc, "integrate_device2")
        fused_pairs[0] += 1
        fused_pairs[1] += int(f.value)

    def bcast(i, async_op):
        # the frame arrives on rank 0; one RCCL broadcast (depth + colour, 2.4 MB) to every slab owner.  RCCL's
        # stream first waits for what is queued on `stream` now, i.e. for the kernel that last read this slot.
        return dist.broadcast(frame_buf(i), src=0, async_op=async_op)

    def launch(i, count=None, timed=False):
        pl, dp, cp, tp = bound_args(i)
        if timed:
            t0 = time.perf_counter()
            pairs.append(ev_pool[len(pairs)])
            pairs[-1][0].record(stream)
            t1 = time.perf_counter()
        rc = set_cull(h, pl) or integ(h, dp, cp, tp, count)
        if timed:
            t2 = time.perf_counter()
            pairs[-1][1].record(stream)
            t3 = time.perf_counter()
            host_t["events"] += (t1 - t0) + (t3 - t2)
            host_t["launch"] += t2 - t1
            host_t["steps"] += 1
        if rc:
            capi.check(rc, "integrate_device")

    counted = []  # (observed voxels, changed-word bytes, observed voxels whose distance word was not read) of every counted launch

    def run(first, last, counting=False, timed=False):
        """Frames [first, last): broadcast + integrate, the next frame's broadcast in flight under each kernel.
        counting: through the counting instance of the kernel (synchronous), results appended to `counted`."""
        if args.pairing and timed and not counting:  # two frames per call and per collective
            pend = dist.broadcast(pair_buf(first), src=0, async_op=True) if (use_dist and args.overlap and first + 1 < last) else None
            for i in range(first, last - 1, 2):
                if use_dist:
                    if args.overlap:
                        pend.wait()
                        pend = dist.broadcast(pair_buf(i + 2), src=0, async_op=True) if i + 3 < last else None
                    else:
                        dist.broadcast(pair_buf(i), src=0)
                launch_pair(i, timed)
            return
        pending = bcast(first, True) if (use_dist and args.overlap and first < last) else None
        detail, rdet = (C.c_uint64 * 2)(), (C.c_uint64 * 3)()
        for i in range(first, last):
            if use_dist:
                tb = time.perf_counter()
                if args.overlap:
                    pending.wait()  # `stream` waits for frame i (the host does not)
                    pending = bcast(i + 1, True) if i + 1 < last else None
                else:
                    bcast(i, False)
                if timed:
                    host_t["broadcast"] += time.perf_counter() - tb
            if counting:
                c = C.c_uint64(0)
                launch(i, C.byref(c), timed)
                capi.check(lib.tsdf_hip_last_count_detail(h, detail), "last_count_detail")
                capi.check(lib.tsdf_hip_last_read_detail(h, rdet), "last_read_detail")
                counted.append((int(detail[0]), int(detail[1]), int(rdet[0]), int(rdet[2])))
            else:
                launch(i, None, timed)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Warm-up launches go through the COUNTING template instance of k_integrate (they integrate exactly the same
    # way), so that in a `rocprofv3 --kernel-trace --stats` / --pmc table of this command the non-counting instance
    # holds the K timed launches and nothing else: its averages there are directly comparable with roofline.*.
    _c = C.c_uint64(0)
    for k in range(args.presaturate):  # the saturated regime: every observed voxel at w == max_weight before anything is timed
        launch(args.warmup + k % args.steps, C.byref(_c))  # (through the COUNTING instance, like the warm-up: a profile's non-counting instance stays the timed launches)
    run(0, args.warmup, counting=True)
    barrier()
    t0 = time.perf_counter()
    run(args.warmup, n_total, timed=True)
    barrier()
    t1 = time.perf_counter()
    wall = t1 - t0
    # average launch duration of the dominant kernel: HIP events on the kernel's stream around each launch (at
    # N > 1 this leaves the frame broadcast out of the kernel's roofline; `value` keeps it, via the wall clock)
    kern_ms = sum(a.elapsed_time(b) for a, b in pairs) / args.steps
    info4 = (C.c_int32 * 4)()
    capi.check(lib.tsdf_hip_last_launch_info(h, info4), "last_launch_info")
    last_info = {"instance": {0: "general", 1: "ALLIN", 2: "k_integrate2"}.get(int(info4[0]) & 0xff, "?") + (" + row intervals" if info4[2] else "") +
                             (", software-pipelined row loop (k_integrate_p)" if int(info4[0]) & 0x100 else ""),
                 "certified_fp32_projection": bool(info4[1]), "reference_cull_decides_voxels": bool(info4[2] == 2), "blocks": int(info4[3])}

    # Algorithmic bytes of the timed frames, counted outside the timed region by running the same frames once more
    # through the counting instance: observed voxels (state-independent: pose + depth only) and the bytes of voxel
    # words whose value changed (the state then holds `steps` more observations per voxel; which words change --
    # the count byte until saturation, colour and distance inside the truncation band, nothing in free space at the
    # hinge -- does not depend on that to first order).
    del counted[:]
    run(args.warmup, n_total, counting=True)
    n_obs_rank = sum(c[0] for c in counted) / args.steps
    chg_rank = sum(c[1] for c in counted) / args.steps
    imp_rank = sum(c[2] for c in counted) / args.steps
    req_rank = sum(c[3] for c in counted) / args.steps
    chg_per_obs = chg_rank / n_obs_rank if n_obs_rank else 0.0

    # Two frames per sweep (tsdf_hip_integrate_device2 -> k_integrate2), report-only side field: the same timed frames
    # once more, in pairs.  Its own event pair around the whole region, its own algorithmic bytes per launch.
    fused2 = None
    if world == 1 and not use_dist and args.extras and args.steps >= 2:
        try:
            fused2 = fused2_leg(lib, h, frames_dev, T_all, planes_all, args, stream, W, H, vol.getLayout() == capi.LAYOUT_PACKED)
        except Exception as e:  # never let a report-only leg break the bench line
            fused2 = {"error": repr(e)}

    # The SATURATED regime on the headline volume (report-only, extras.keys): after >= 100 more frames every observed voxel
    # sits at w == max_weight (octree.cpp:157-159) -- where configs[3] spends nine tenths of its 1000 frames -- so the count
    # byte stops ticking and the colour word is only rewritten where the running average moves a channel.
    saturated = None
    if world == 1 and not use_dist and args.extras == 1 and args.keys and not args.presaturate:
        try:
            n_pre = 110
            for k in range(n_pre):
                launch(args.warmup + k % args.steps)
            torch.cuda.synchronize(dev)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
            for k in range(args.steps):
                pl, dp, cp, tp = bound_args(args.warmup + k)
                evs[k][0].record(stream)
                capi.check(set_cull(h, pl) or integ(h, dp, cp, tp, None), "integrate_device")
                evs[k][1].record(stream)
            torch.cuda.synchronize(dev)
            sat_ms = sum(a.elapsed_time(b) for a, b in evs) / args.steps
            del counted[:]
            run(args.warmup, n_total, counting=True)
            s_obs, s_chg = sum(c[0] for c in counted) / args.steps, sum(c[1] for c in counted) / args.steps
            s_imp, s_req = sum(c[2] for c in counted) / args.steps, sum(c[3] for c in counted) / args.steps
            pk = vol.getLayout() == capi.LAYOUT_PACKED
            s_bpv = ((8 if args.color else 5) if pk else (12 if args.color else 8))
            s_moved = (s_req if s_req else s_bpv * s_obs - 4 * s_imp) + s_chg + (8 if args.color else 4) * W * H
            s_key = f"{res3[0]}x{res3[1]}x{z_end - z_begin}_c{args.color}_{'packed' if pk else 'f32w'}_saturated"
            s_prof, s_why = pmc_traffic(s_key, kernel_sha16())
            saturated = {"key": s_key, "grid": list(res3), "image": [W, H], "color": bool(args.color), "presaturate_launches": n_pre,
                         "weights": "every observed voxel at w == max_weight (the headline's timed region runs at w <= %d)" % (args.warmup + args.steps),
                         "steps": args.steps, "kernel_ms": sat_ms, "frames_per_s": 1e3 / sat_ms,
                         "Mvoxels_per_s": float(res3[0]) * res3[1] * res3[2] / (sat_ms * 1e-3) / 1e6,
                         "observed_voxels_per_frame": s_obs, "plane_bytes_requested": s_req, "changed_word_bytes": s_chg,
                         "bytes_moved_per_launch": s_moved, "GBps": s_moved / (sat_ms * 1e-3) / 1e9,
                         "frac": s_moved / (sat_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": s_prof["hbm_bytes_per_launch"] if s_prof else None,
                         "frac_check": ({"pmc_bytes_per_launch": s_prof["hbm_bytes_per_launch"], "kernel_counted_bytes_per_launch": s_moved,
                                         "relative_difference": abs(s_prof["hbm_bytes_per_launch"] - s_moved) / s_prof["hbm_bytes_per_launch"],
                                         "agree_within_5_percent": abs(s_prof["hbm_bytes_per_launch"] - s_moved) / s_prof["hbm_bytes_per_launch"] < 0.05,
                                         "profile_tag": s_prof.get("tag"), "kernel_ms_in_profile": s_prof.get("kernel_ms_in_profile")}
                                        if s_prof else {"note": s_why})}
        except Exception as e:
            saturated = {"error": repr(e)}

    # isolated cost of one frame broadcast (report only)
    bcast_ms = None
    if use_dist:
        barrier()
        tb = time.perf_counter()
        for i in range(args.warmup, n_total):
            bcast(i, False)
        barrier()
        bcast_ms = (time.perf_counter() - tb) / args.steps * 1e3

    t = torch.tensor([wall, kern_ms, n_obs_rank, chg_rank], dtype=torch.float64, device=dev)
    per_rank_kernel_ms = [kern_ms]
    per_rank_observed = [n_obs_rank]
    if use_dist:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        allk = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(allk, t[1:2].clone())
        per_rank_kernel_ms = [float(x) for x in allk]
        allo = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(allo, t[2:3].clone())
        per_rank_observed = [float(x) for x in allo]   # load balance of the Z-slabs: observed voxels per frame and slab
        wall = float(tmax[0])
        n_obs_all = float(tsum[2])
    else:
        n_obs_all = n_obs_rank

    if rank == 0:
        vox_total = float(res3[0]) * res3[1] * res3[2]
        fps = args.steps / wall
        packed = vol.getLayout() == capi.LAYOUT_PACKED
        bpp = 8 if args.color else 4                              # frame bytes per pixel (depth + bgra)
        ref_bpv = 24 if args.color else 16                        # the reference's (d, w, rgb) record, read + write
        read_bpv = ((8 if args.color else 5) if packed else (12 if args.color else 8))  # voxel words an observed voxel must read
        alg_bytes = read_bpv * n_obs_rank + chg_rank + bpp * W * H  # rank 0's launch: the layout's per-voxel figure x observed voxels
        # ... and what the kernel moves: the plane bytes it requests (distances it rebuilds from the counts are not read; the
        # words of a quad none of whose voxels turns out to be observed are, since the request goes out with the frame gather)
        # + the changed words + the frame.  The plain kernels do not count requests: then the layout's figure stands in
        moved_bytes = (req_rank if req_rank else read_bpv * n_obs_rank - 4 * imp_rank) + chg_rank + bpp * W * H
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        moved_gbps = moved_bytes / (kern_ms * 1e-3) / 1e9
        sha = kernel_sha16()
        key = f"{res3[0]}x{res3[1]}x{z_end - z_begin}_c{args.color}_{'packed' if packed else 'f32w'}" + ("_saturated" if args.presaturate else "")
        prof, why = pmc_traffic(key, sha)
        # the parsed fraction is the one profiles/ reproduces: when a PMC profile of THIS kernel source exists, the kernel's own
        # byte count must agree with the counters within 5 %; a disagreement is carried in frac_check.agree_within_5_percent
        # (the fraction itself stays numeric -- ADVICE r05: a null there broke every consumer with a TypeError instead of a
        # readable failure; tests/test_bench_dist_gpu.py asserts the flag)
        frac_moved = moved_gbps / HBM_PEAK_GBS
        if prof:
            dev_pmc = abs(prof["hbm_bytes_per_launch"] - moved_bytes) / prof["hbm_bytes_per_launch"]
            frac_check = {"pmc_bytes_per_launch": prof["hbm_bytes_per_launch"], "kernel_counted_bytes_per_launch": moved_bytes,
                          "relative_difference": dev_pmc, "agree_within_5_percent": dev_pmc < 0.05}
        else:
            frac_check = {"pmc_bytes_per_launch": None, "kernel_counted_bytes_per_launch": moved_bytes,
                          "note": "no PMC profile of this kernel source under profiles/ (" + str(why) + "): frac rests on the kernel's "
                                  "own request counter alone"}
        out = {
            "metric": f"integrateCloud throughput, Scene A turntable depth frames, {W}x{H} -> voxel grid",
            "value": vox_total * fps / 1e6,
            "unit": "Mvoxels/s",
            "frames_per_s": fps,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"integrateCloud {res3[0]}x{res3[1]}x{res3[2]} grid (voxel 2^-8 m), "
                            f"integrateColor={'true' if args.color else 'false'}, {W}x{H} Scene-A turntable frames "
                            f"resident in HBM (BASELINE configs[{args.config}] integrate leg)" +
                            (f", Z-slab {z_end - z_begin} planes/GPU, one RCCL frame broadcast per step"
                             + (" overlapped with the previous kernel" if args.overlap else "") if use_dist else ""),
                "principal_offset": args.principal_offset or None,
                "presaturate_launches": args.presaturate or None,   # the saturated regime (w == max_weight) when set
                "last_launch": last_info,
                "grid": list(res3), "image": [W, H], "color": bool(args.color),
                "layout": "packed" if packed else "f32w",
                "observed_voxels_per_frame": n_obs_all,
                "plane_placement": placement,   # tsdf_hip_create keeps the fastest of up to 3 allocations of the planes
                "parallelism": f"zslab{world}",
            },
            "roofline": {
                # ONE definition: achieved = the bytes the kernel MOVES per launch (its own request counter +
                # changed words + frame: what the PMC counters confirm) / kernel_ms.  The layout's per-voxel figure of rounds
                # 2-4 and SURVEY 8(d)'s reference-record figure are the named side fields frac_layout_bytes / frac_survey_8d.
                "bound": "hbm", "achieved": moved_gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": frac_moved,
                "frac_layout_bytes": achieved / HBM_PEAK_GBS,
                "frac_survey_8d": (ref_bpv * n_obs_rank + bpp * W * H) / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "frac_check": frac_check,
                "traffic": prof["hbm_bytes_per_launch"] if prof else None,
                "traffic_measured_in_this_run": False,  # PMC counters need rocprofv3 around the process: `traffic` is quoted from
                                                        # the committed profile of this same command (tools/run_rocprof.sh), below
                "kernel": "k_integrate", "kernel_ms": kern_ms, "kernel_sha16": sha,
                "algorithmic_bytes_per_launch": alg_bytes,
                "algorithmic_bytes": {"read_per_observed_voxel": read_bpv, "observed_voxels": n_obs_rank,
                                      "distance_words_not_read": imp_rank, "distance_bytes_not_read": 4 * imp_rank,
                                      "plane_bytes_requested": req_rank,
                                      "changed_word_bytes": chg_rank, "changed_bytes_per_observed_voxel": chg_per_obs,
                                      "frame_bytes": bpp * W * H},
                "bytes_moved": {"per_launch": moved_bytes, "GBps": moved_gbps,
                                "frac": moved_gbps / HBM_PEAK_GBS,
                                "note": "what this kernel moves: the plane bytes it requests (counted by its counting instance: no distance word "
                                        "where a cell never observed inside the truncation band lets the count tell it, DESIGN.md 3.1c; the words "
                                        "of every quad it visits, observed or not, because the request goes out together with the frame gather) "
                                        "+ the changed words + the frame.  THIS is the figure the PMC traffic agrees with, the kernel's real "
                                        "share of the HBM peak, and since round 5 what roofline.achieved / roofline.frac are"},
                "traffic_from_profile": ({"tag": prof.get("tag"), "commit": prof.get("git_head_when_summarised"), "read_bytes": prof.get("read_bytes"),
                                     "written_bytes": prof.get("written_bytes"),
                                     "frac_of_peak_by_traffic": prof["hbm_bytes_per_launch"] / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "kernel_ms_in_profile": prof.get("kernel_ms_in_profile")} if prof else why),
                "reference_record_bytes_per_observed_voxel": ref_bpv,
                "reference_record_bytes_per_launch": ref_bpv * n_obs_rank + bpp * W * H,
                "sweep_upper_bound_bytes": ref_bpv * vox_total / world,
                "survey_8d_note": "frac_survey_8d prices an observed voxel at SURVEY 8(d)'s record of the reference (d, w, rgb read and "
                                  "written: 24 B with colour, 16 B without) + the frame; it exceeds what the kernel moves because the "
                                  "shipped PACKED layout holds the weight as a count in the colour word's free byte (8 B per voxel) and "
                                  "unchanged words are not written back -- a value above 1 is that ratio, not a bandwidth",
                "note": "achieved/frac: the bytes the kernel moves per launch (plane bytes it requests, counted by its counting instance, "
                        "+ words whose value changed + the frame) / kernel_ms / 8 TB/s -- the number the PMC counters reproduce; "
                        "frac_layout_bytes: the shipped layout's per-voxel figure (8 B read per observed voxel + changed words + frame: "
                        "rounds 2-4's definition, time per unit of work); frac_survey_8d: SURVEY 8(d)'s 24 B (16 B) reference record "
                        "per observed voxel, > 1 because the layout is smaller than that record; traffic: PMC FETCH_SIZE x2 + "
                        "WRITE_SIZE over this command's timed launches, quoted from profiles/ while the kernel-source hash matches",
            },
        }
        n_h = max(1, host_t["steps"])
        out["host_us_per_step"] = {
            "integrate_calls": host_t["launch"] / n_h * 1e6, "event_records": host_t["events"] / n_h * 1e6,
            "broadcast_calls": host_t["broadcast"] / n_h * 1e6 if use_dist else None,
            "total": (host_t["launch"] + host_t["events"] + host_t["broadcast"]) / n_h * 1e6,
            "frac_of_kernel": (host_t["launch"] + host_t["events"] + host_t["broadcast"]) / n_h * 1e3 / kern_ms if kern_ms else None,
            "note": "rank 0's host time spent ENQUEUEING one timed step (two ctypes calls with pre-bound arguments: cull planes + "
                    "integrate; two event records; the frame broadcast's wait() + next issue at N > 1) -- the GPU runs behind it" +
                    ("; backend gloo stages device tensors through the host inside broadcast(): its share is not RCCL's" if use_dist and backend != "nccl" else "")}
        if args.pairing:
            out["pairing"] = {"frames_per_call": 2, "pairs_launched": fused_pairs[0], "pairs_swept_once_by_every_slab": fused_pairs[1],
                              "note": "tsdf_hip_integrate_device2 per pair (k_integrate2 where both poses see the whole slab); at N > 1 one "
                                      "broadcast carries both frames.  roofline.kernel_ms = event time of the pairs / steps (per FRAME); the "
                                      "byte counts are the single-frame kernel's (counted frame by frame) and do not describe k_integrate2"}
        if emulated:
            out["emulated_slab"] = emulated
        if args.dry_run_ranks:
            out["dry_run"] = f"{world} ranks on ONE GPU over gloo: control flow only, no number here is a scaling number"
        if use_dist:
            n_dev = torch.cuda.device_count()
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:  # (a build without the nccl bindings: say so instead of dying in a report field)
                rccl = repr(e)
            out["multi_gpu"] = {"host": "one process per GPU, torch.distributed", "world_size": dist.get_world_size(),
                                "per_rank_kernel_ms": per_rank_kernel_ms, "frame_broadcast_ms_isolated": bcast_ms,
                                "overlap": bool(args.overlap), "backend": backend, "planes_per_gpu": z_end - z_begin,
                                # what the first run on a real node should show at a glance
                                "observed_voxels_per_rank": per_rank_observed,
                                "kernel_ms_spread": (max(per_rank_kernel_ms) / min(per_rank_kernel_ms)) if min(per_rank_kernel_ms) > 0 else None,
                                "rccl_version": rccl, "visible_devices": n_dev,
                                "peer_access": [[bool(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(n_dev)]
                                                for i in range(n_dev)],
                                "slab_hbm_gib_per_rank": slab_gb, "free_hbm_gib_before_create_rank0": free_gb_before}
        if calibration:
            out["calibration"] = calibration
        if world == 1 and args.extras:
            out["extras"] = extras(vol, poses[-1], W, H) if args.extras == 1 else {}
            if fused2 is not None:
                out["extras"]["fused2"] = fused2
            if args.scene_b and args.extras == 1:
                out["extras"]["scene_b"] = scene_b_leg(res, args.color, 8.0 if args.cpu_baseline else 0.0)
        if world == 1 and not use_dist and args.host_path:
            out["host_path"] = host_path_leg(vol, sc, poses, bool(args.color), args.warmup, n_total)
            if res3[0] == res3[1] == res3[2]:
                cd = cpp_dropin_leg(sc, poses, res3[0], W, H, bool(args.color))
                out["host_path"]["cpp_dropin"] = cd
                for k in ("pairing_off", "pairing_on"):
                    if isinstance(cd.get(k), dict) and "sustained_frames_per_s" in cd[k]:
                        out["host_path"]["cpp_dropin_frames_per_s" + ("_frame_pairing" if k == "pairing_on" else "")] = cd[k]["sustained_frames_per_s"]
        if world == 1 and not use_dist and args.extras == 1 and args.keys:
            # every integrate key in the driver-run line; the headline volume is freed first
            keys = {}
            if saturated is not None:
                keys["saturated"] = saturated
            vol.close()
            torch.cuda.empty_cache()
            if not (res3 == (2048, 2048, 2048) and not args.color and packed):
                keys["colourless"] = key_leg((2048, 2048, 2048), 640, 480, False)
            keys["config4_slab"] = key_leg((4096, 4096, 512), 1280, 960, True,
                                           slab_of="one Z-slab of BASELINE configs[4] (4096^3 over 8 GPUs = 512 planes per GPU), 1280x960 frames")
            out["extras"]["keys"] = keys
        if world == 1 and args.cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, sc, res3, size3, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    vol.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main_inprocess(args):
    """--host inprocess: ONE process, one tsdf_hip_create_multi handle over N GPUs -- the host the C++ drop-in
    (TSDFVolumeOctree::setDevices) uses.  Frames are resident on the first GPU; every step fans the frame out to the
    other slabs by peer copy (xGMI) and launches k_integrate on every slab's own stream.  Strong scaling of the same
    grid.  TSDF_BENCH_ONE_DEVICE=1 (one-GPU boxes) repeats ordinal 0: the code path, not the numbers."""
    import torch
    from cpu_tsdf_amd import capi, synth
    from cpu_tsdf_amd.volume import TSDFVolumeOctree
    lib = capi.load()
    n = args.gpus
    one_dev = os.environ.get("TSDF_BENCH_ONE_DEVICE") == "1"
    devices = [0] * n if one_dev else list(range(n))
    if not one_dev and lib.tsdf_hip_device_count() < n:
        raise SystemExit(f"--gpus {n} but only {lib.tsdf_hip_device_count()} HIP devices are visible")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    res, voxel = args.res, 2.0 ** -8
    planes = args.planes or res
    res3 = (res, res, planes * n) if args.scaling == "weak" else (res, res, planes)
    size3 = tuple(r * voxel for r in res3)
    S, W, H = size3[0], args.width, args.height
    sc = synth.Scene(S, W, H)
    if res3[2] != res3[0]:
        sc.h = np.array([0.47 * size3[0], 0.47 * size3[1], 0.47 * size3[2]])
    vol = TSDFVolumeOctree()
    vol.setResolution(*res3)
    vol.setGridSize(*size3)
    vol.setImageSize(W, H)
    vol.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
    vol.setSensorDistanceBounds(0.0, 3.0 * max(size3))
    vol.setDepthTruncationLimits(0.03, 0.03)
    vol.setIntegrateColor(bool(args.color))
    vol.setLayout({"auto": capi.LAYOUT_AUTO, "f32w": capi.LAYOUT_F32W, "packed": capi.LAYOUT_PACKED}[args.layout])
    vol.setDevices(devices)
    vol.reset()
    h = vol._need()
    slabs = vol.slabs()

    n_total = args.warmup + args.steps
    n_distinct = args.frames or n_total
    radius = 2.2 * max(size3) / S
    poses = [synth.turntable_pose(i, n_distinct, S, radius_factor=radius) for i in range(n_total)]
    T_all = [synth.cam_from_vol_f32(p) for p in poses]
    fplanes = 2 if args.color else 1
    frames_dev = torch.empty((n_total, fplanes, H, W), dtype=torch.float32, device=dev)
    for i, p in enumerate(poses):
        frames_dev[i, 0].copy_(torch.from_numpy(sc.depth(p)))
        if args.color:
            frames_dev[i, 1].view(torch.uint8).view(H, W, 4).copy_(torch.from_numpy(sc.bgra(i)))
    torch.cuda.synchronize(dev)  # the frames are complete before the library's own streams read them

    def launch(i, count=None):
        fr = frames_dev[i]
        capi.check(lib.tsdf_hip_integrate_device(h, C.c_void_p(fr[0].data_ptr()), C.c_void_p(fr[1].data_ptr()) if args.color else None,
                                                 capi.as_f32p(T_all[i]), count), "integrate_device")

    detail = (C.c_uint64 * 2)()
    for i in range(args.warmup):  # warm-up through the counting instance, as in the per-process host
        launch(i, C.byref(C.c_uint64(0)))
    vol.synchronize()
    capi.check(lib.tsdf_hip_multi_timing(h, 1), "multi_timing")
    t0 = time.perf_counter()
    for i in range(args.warmup, n_total):
        launch(i)
    vol.synchronize()
    wall = time.perf_counter() - t0
    per_slab_ms = []
    for k in range(len(slabs)):
        ms, cnt = C.c_float(0), C.c_int32(0)
        capi.check(lib.tsdf_hip_multi_kernel_ms(h, k, C.byref(ms), C.byref(cnt)), "multi_kernel_ms")
        per_slab_ms.append(ms.value / max(1, cnt.value))
    capi.check(lib.tsdf_hip_multi_timing(h, 0), "multi_timing")
    n_obs = chg = 0
    for i in range(args.warmup, n_total):  # the same frames once more through the counting instance
        launch(i, C.byref(C.c_uint64(0)))
        capi.check(lib.tsdf_hip_last_count_detail(h, detail), "last_count_detail")
        n_obs += int(detail[0])
        chg += int(detail[1])
    n_obs /= args.steps
    chg /= args.steps

    vox_total = float(res3[0]) * res3[1] * res3[2]
    fps = args.steps / wall
    packed = vol.getLayout() == capi.LAYOUT_PACKED
    bpp = 8 if args.color else 4
    read_bpv = ((8 if args.color else 5) if packed else (12 if args.color else 8))
    alg_bytes = read_bpv * n_obs + chg + n * bpp * W * H  # all slabs' launches of one step together
    kern_ms = max(per_slab_ms)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    out = {
        "metric": f"integrateCloud throughput, Scene A turntable depth frames, {W}x{H} -> voxel grid",
        "value": vox_total * fps / 1e6, "unit": "Mvoxels/s", "frames_per_s": fps, "n_gpus": n, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"integrateCloud {res3[0]}x{res3[1]}x{res3[2]} grid (voxel 2^-8 m), integrateColor="
                        f"{'true' if args.color else 'false'}, {W}x{H} Scene-A turntable frames resident in HBM on GPU 0 "
                        f"(BASELINE configs[{args.config}] integrate leg), ONE process / one tsdf_hip_create_multi handle over "
                        f"{n} GPUs, frame fan-out by peer copy",
            "grid": list(res3), "image": [W, H], "color": bool(args.color), "layout": "packed" if packed else "f32w",
            "observed_voxels_per_frame": n_obs, "parallelism": f"zslab{n}-inprocess",
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS * n, "unit": "GB/s", "frac": achieved / (HBM_PEAK_GBS * n),
            "traffic": None, "kernel": "k_integrate", "kernel_ms": kern_ms, "kernel_sha16": kernel_sha16(),
            "algorithmic_bytes_per_launch": alg_bytes,
            "note": "all slabs of one step together: algorithmic bytes summed over the slabs / the slowest slab's average k_integrate "
                    "time (HIP events on every slab's own stream, tsdf_hip_multi_kernel_ms), against N x the per-GPU HBM peak",
        },
        "multi_gpu": {"host": "one process, tsdf_hip_create_multi", "devices": devices, "per_slab_kernel_ms": per_slab_ms,
                      "slabs": [{"device": s[0], "z_begin": s[1], "z_end": s[2], "halo": s[3]} for s in slabs]},
    }
    print(json.dumps(out), flush=True)
    vol.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- integrateCloud throughput on MI355X (BASELINE.json metric).

One "step" = one integrateCloud pass of one synthetic 640x480 depth(+colour) frame over the whole
voxel grid.  N=1 workload = BASELINE.json configs[3]: 2048^3 grid (8 m, voxel 2^-8 m),
integrateColor=true, Scene A turntable frames.  Frames are resident in HBM before
the timed region.  For N>1 (one process per GPU, torch.distributed/RCCL) the grid is Z-slab
partitioned: weak scaling extends the grid along z by 2048 planes per GPU (default) -- every rank
integrates its own 2048-plane slab after an RCCL broadcast of the frame from rank 0.

Prints ONE JSON line on rank 0 (see the contract in the task statement), with `roofline` for the
dominant kernel (k_integrate; HBM-bound) and `cpu_baseline` (the reference CPU path, timed on this
host on a bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--res", type=int, default=2048, help="x/y resolution (and planes per GPU unless --planes)")
    ap.add_argument("--planes", type=int, default=0, help="z planes per GPU (weak) / in total (strong); 0 = --res.  "
                    "--res 4096 --planes 512 --width 1280 --height 960 at N=8 is BASELINE configs[4]")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--color", type=int, default=1)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--layout", choices=["auto", "f32w", "packed"], default="auto",
                    help="HBM weight layout (include/tsdf_hip.h TSDF_LAYOUT_*); auto = packed when max_weight <= 255")
    ap.add_argument("--frames", type=int, default=0, help="distinct frames on the turntable (default steps+warmup)")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--extras", type=int, default=1, help="also time renderView + marching cubes once (N=1, untimed region)")
    ap.add_argument("--scene-b", type=int, default=1, help="with --extras: the Scene-B (camera inside the volume) leg; the "
                    "rocprof run turns it off so that k_integrate's average is the headline workload's alone")
    return ap.parse_args()


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
    from oracle.oracle import OracleVolume, SlabOracle
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
    from cpu_tsdf_amd.volume import MarchingCubesTSDFOctree
    out = {}
    try:
        vol.renderView(pose, 1, camera_frame=False)  # warm-up (scratch allocation)
        t0 = time.perf_counter()
        img = vol.renderView(pose, 1, camera_frame=False)
        dt = time.perf_counter() - t0
        out["renderView_ms"] = dt * 1e3
        out["renderView_rays_per_s"] = W * H / dt
        out["renderView_hits"] = int(np.isfinite(img[..., 0]).sum())
        out["renderView_mean_steps"] = float(img[..., 7].mean())
        out["renderView_steps_per_s"] = float(img[..., 7].sum()) / dt
        lib = capi_mod().load()
        n = C.c_uint64(0)
        lib.tsdf_hip_march(vol._need(), C.c_float(1.0), 1, C.byref(n))  # warm-up (buffer growth)
        t0 = time.perf_counter()
        rc = lib.tsdf_hip_march(vol._need(), C.c_float(1.0), 1, C.byref(n))
        dt = time.perf_counter() - t0
        if rc == 0:
            rx, ry, rz = vol.getResolution()
            out["reconstruct_ms"] = dt * 1e3
            out["reconstruct_triangles"] = int(n.value)
            out["reconstruct_Mvoxels_per_s"] = rx * ry * float(rz) / dt / 1e6
            # 8 B (12 B colour) per voxel + 36 B (+9 B colour) per triangle, whole call (classify + sort + emit)
            color = bool(vol._p.integrate_color)
            out["reconstruct_algorithmic_GBps"] = ((12 if color else 8) * rx * ry * float(rz) + (45 if color else 36) * n.value) / dt / 1e9
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

        def run(i, count=None):
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
        out.update({"grid": [res] * 3, "size_m": 10.0, "sensor_range_m": [0.0, 3.0], "observed_voxels_per_frame": int(c.value),
                    "gpu_ms_per_frame": ms, "gpu_frames_per_s": 1e3 / ms})
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


def capi_mod():
    from cpu_tsdf_amd import capi
    return capi


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # Test hooks for boxes with ONE GPU (gpurun): TSDF_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # TSDF_BENCH_BACKEND=gloo moves the collectives off RCCL (which refuses two ranks per device), so the N>1
    # code path can be smoke-run there.  Numbers from such a run mean nothing; the driver never sets these.
    if os.environ.get("TSDF_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("TSDF_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from cpu_tsdf_amd import capi, synth
    from cpu_tsdf_amd.volume import TSDFVolumeOctree
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
    size3 = tuple(r * voxel for r in res3)
    S = size3[0]
    W, H = args.width, args.height
    sc = synth.Scene(S, W, H)  # sphere + far-face box scaled to the x/y extent
    if res3[2] != res3[0]:
        sc.h = np.array([0.47 * size3[0], 0.47 * size3[1], 0.47 * size3[2]])

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

    # ---- synthetic frames, resident in HBM before the timed region ---------------------------------
    n_total = args.warmup + args.steps
    n_distinct = args.frames or n_total
    radius = 2.2 * max(size3) / S
    poses = [synth.turntable_pose(i, n_distinct, S, radius_factor=radius) for i in range(n_total)]
    T_all = [synth.cam_from_vol_f32(p) for p in poses]
    # one allocation per frame: [depth | bgra] back to back, which is what the kernel's single frame
    # descriptor wants (no staging copy) and what ONE broadcast per frame can carry
    fplanes = 2 if args.color else 1
    frames_dev = torch.empty((n_total, fplanes, H, W), dtype=torch.float32, device=dev)
    if rank == 0:
        for i, p in enumerate(poses):
            frames_dev[i, 0].copy_(torch.from_numpy(sc.depth(p)))
            if args.color:
                frames_dev[i, 1].view(torch.uint8).view(H, W, 4).copy_(torch.from_numpy(sc.bgra(i)))
    recv = torch.empty((fplanes, H, W), dtype=torch.float32, device=dev) if world > 1 else None
    lib = capi.load()
    h = vol._need()

    pairs = []  # HIP event pairs around each timed launch, on the stream the kernel runs on

    def step(i, count=None, timed=False):
        if world > 1:
            # the frame arrives on rank 0; one RCCL broadcast (depth + colour, 2.4 MB) to every slab owner
            fr = frames_dev[i] if rank == 0 else recv
            dist.broadcast(fr, src=0)
        else:
            fr = frames_dev[i]
        if timed:
            pairs.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
            pairs[-1][0].record(stream)
        rc = lib.tsdf_hip_integrate_device(h, C.c_void_p(fr[0].data_ptr()), C.c_void_p(fr[1].data_ptr()) if args.color else None,
                                           capi.as_f32p(T_all[i]), count)
        capi.check(rc, "integrate_device")
        if timed:
            pairs[-1][1].record(stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Warm-up launches go through the COUNTING template instance of k_integrate (they integrate exactly the same
    # way), so that in a `rocprofv3 --kernel-trace --stats` table of this command the non-counting instance holds
    # the K timed launches and nothing else: its average there is directly comparable with roofline.kernel_ms.
    for i in range(args.warmup):
        step(i, C.byref(C.c_uint64(0)))
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, n_total):
        step(i, timed=True)
    barrier()
    t1 = time.perf_counter()
    wall = t1 - t0
    # average launch duration of the dominant kernel: HIP events on the kernel's stream around each launch (at
    # N > 1 this leaves the frame broadcast out of the kernel's roofline; `value` keeps it, via the wall clock)
    kern_ms = sum(a.elapsed_time(b) for a, b in pairs) / args.steps

    # observed voxels of the timed frames (state-independent: depends on pose + depth only), counted
    # outside the timed region by re-running the same frames with the counter read back
    n_obs = 0
    for i in range(args.warmup, n_total):
        c = C.c_uint64(0)
        step(i, C.byref(c))
        n_obs += c.value
    n_obs_rank = n_obs / args.steps

    t = torch.tensor([wall, kern_ms, n_obs_rank], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        wall = float(tmax[0])
        n_obs_all = float(tsum[2])
    else:
        n_obs_all = n_obs_rank

    if rank == 0:
        vox_total = float(res3[0]) * res3[1] * res3[2]
        fps = args.steps / wall
        bpv = 24 if args.color else 16
        bpp = 8 if args.color else 4
        alg_bytes = bpv * n_obs_rank + bpp * W * H  # rank 0's launch
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        # bytes the chosen HBM layout actually has to move per observed voxel (read + write back):
        # F32W d,w(,rgb) = 16 (24); PACKED d + colour|count word = 16, d + count byte = 10
        packed = vol.getLayout() == capi.LAYOUT_PACKED
        lbpv = (16 if args.color else 10) if packed else bpv
        layout_bytes = lbpv * n_obs_rank + bpp * W * H
        traffic = None
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(prof):
            try:
                pj = json.load(open(prof))
                key = f"{res3[0]}x{res3[1]}x{z_end - z_begin}_c{args.color}_{'packed' if packed else 'f32w'}"
                traffic = pj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "integrateCloud throughput, Scene A turntable depth frames, 640x480 -> voxel grid",
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
                            f"resident in HBM" + (f", Z-slab {z_end - z_begin} planes/GPU, RCCL frame broadcast"
                                                  if world > 1 else " (BASELINE configs[3] integrate leg)"),
                "grid": list(res3), "image": [W, H], "color": bool(args.color),
                "layout": "packed" if packed else "f32w",
                "observed_voxels_per_frame": n_obs_all,
                "parallelism": f"zslab{world}",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "k_integrate", "kernel_ms": kern_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                "bytes_per_observed_voxel": bpv,
                "layout_bytes_per_observed_voxel": lbpv,
                "layout_bytes_per_launch": layout_bytes,
                "frac_layout": layout_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "sweep_upper_bound_bytes": bpv * vox_total / world,
                "note": ("achieved/frac use SURVEY 8d's algorithmic record (24 B per observed voxel with colour, 16 B without); "
                         "the PACKED HBM layout only has to move layout_bytes_per_observed_voxel, so frac can approach or pass 1 "
                         "while the memory system runs at frac_layout (nominal layout bytes) / traffic (measured, PMC)") if packed else
                        "F32W layout: the algorithmic record is what the layout moves",
            },
        }
        if world == 1 and args.extras:
            out["extras"] = extras(vol, poses[-1], W, H)
            if args.scene_b:
                out["extras"]["scene_b"] = scene_b_leg(res, args.color, 8.0 if args.cpu_baseline else 0.0)
        if world == 1 and args.cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, sc, res3, size3, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    vol.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

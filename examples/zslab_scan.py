#!/usr/bin/env python3
"""Multi-GPU walk-through: fuse a synthetic turntable scan into ONE volume that is Z-slab partitioned over the
GPUs of a node, render a view, mesh it and write the results -- the calls a user of the reference's
TSDFVolumeOctree / MarchingCubesTSDFOctree would make, on cpu_tsdf_amd.zslab.ZSlabVolume.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        examples/zslab_scan.py --res 2048 --frames 100 --out /tmp/scan

One process per GPU (backend "nccl" = RCCL).  With a single process (plain `python examples/zslab_scan.py`) the same
script runs on one GPU.  Every ZSlabVolume call is collective: all ranks make it, rank 0 supplies the frames."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(args, slab_factory=None, log=print):
    import torch.distributed as dist

    from cpu_tsdf_amd import synth
    from cpu_tsdf_amd.zslab import ZSlabVolume

    rank = dist.get_rank() if dist.is_initialized() else 0
    res, (W, H) = args.res, args.image
    sc = synth.scene_a(res, W, H)  # sphere in a box, 2^-8 m voxels; cameras on a circle around it

    def configure(v):  # the reference's setters, applied to every rank's slab
        v.setResolution(res, res, res)
        v.setGridSize(sc.size, sc.size, sc.size)
        v.setImageSize(W, H)
        v.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
        v.setSensorDistanceBounds(0.0, 3 * sc.size)
        v.setIntegrateColor(True)

    vol = ZSlabVolume(configure, res, slab_factory=slab_factory)
    t0 = time.perf_counter()
    for i in range(args.frames):
        pose = synth.turntable_pose(i, args.frames, sc.size)
        if rank == 0:  # only the ingest rank needs the frame; everyone gets it by one broadcast
            vol.integrateCloud(sc.depth(pose), sc.bgra(i), pose)
        else:
            vol.integrateCloud(None, None, pose)
    if rank == 0:
        log(f"integrated {args.frames} frames in {time.perf_counter() - t0:.2f} s (incl. frame synthesis on rank 0)")
    view = vol.renderView(synth.turntable_pose(0, 8, sc.size))  # every rank gets the full image
    os.makedirs(args.out, exist_ok=True)
    n_tri = vol.save_ply(os.path.join(args.out, "mesh.ply"), w_min=2.0, color_by_rgb=True)  # every rank writes its byte ranges
    vol.save(os.path.join(args.out, "volume.vol"))  # the reference's checkpoint format, written on rank 0
    if rank == 0:
        np.save(os.path.join(args.out, "view.npy"), view)
        log(f"renderView hit {int(np.isfinite(view[..., 2]).sum())} of {view.shape[0] * view.shape[1]} rays; {n_tri} triangles")
    vol.close()
    return n_tri


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--image", type=int, nargs=2, default=(640, 480))
    ap.add_argument("--out", default="zslab_scan_out")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    try:
        run(args)
    finally:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()


if __name__ == "__main__":
    main()

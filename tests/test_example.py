"""The multi-GPU walk-through (examples/zslab_scan.py) runs: under gloo with the oracle-backed slab on CPU (world 2),
and on one GPU with the HIP slab."""
import argparse
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))


def _args(out):
    return argparse.Namespace(res=32, frames=3, image=(80, 60), out=out)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import zslab_scan
    from tests.fake_slab import OracleSlab
    n = zslab_scan.run(_args(out), slab_factory=OracleSlab, log=lambda *_: None)
    assert n > 100
    dist.barrier()
    dist.destroy_process_group()


def _check_outputs(out):
    blob = open(os.path.join(out, "mesh.ply"), "rb").read()
    assert blob.startswith(b"ply\nformat binary_little_endian") and b"element face" in blob
    assert os.path.getsize(os.path.join(out, "volume.vol")) > 1000
    view = np.load(os.path.join(out, "view.npy"))
    assert view.shape == (60, 80, 8) and np.isfinite(view[..., 2]).sum() > 50


def test_example_runs_under_gloo_world_2(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "scan")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    _check_outputs(out)


@pytest.mark.gpu
def test_example_runs_on_one_gpu(gpu, tmp_path):
    import zslab_scan
    out = str(tmp_path / "scan")
    assert zslab_scan.run(_args(out), log=lambda *_: None) > 100
    _check_outputs(out)


def _build_fuse_node(tmp_path):
    import subprocess

    from cpu_tsdf_amd import build as b
    b.build_shell()
    exe = str(tmp_path / "fuse_node")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-fopenmp", "-ffp-contract=off", os.path.join(ROOT, "examples", "fuse_node.cpp")] +
                          b.host_include_flags() + ["-L" + b.LIBDIR, "-lcpu_tsdf_hip", "-ltsdf_hip", "-Wl,-rpath," + b.LIBDIR, "-o", exe])
    return exe


def test_cpp_example_compiles_against_the_drop_in_headers(tmp_path):
    """examples/fuse_node.cpp is user code written against the reference's public API (+ setDevices): it must build
    against include/cpu_tsdf with nothing but the headers and the two libraries."""
    assert os.path.exists(_build_fuse_node(tmp_path))


@pytest.mark.gpu
def test_cpp_example_is_partition_independent(gpu, tmp_path):
    """The same program on one handle and on three Z-slab handles (setDevices 0,0,0): identical mesh, render and
    getFxn, identical .vol bytes."""
    import json
    import subprocess
    exe = _build_fuse_node(tmp_path)
    outs = []
    for tag, dev in (("one", []), ("three", ["0,0,0"])):
        vol = str(tmp_path / f"{tag}.vol")
        args = [exe, "128", "4"] + (dev if dev else ["0"]) + [vol]
        r = subprocess.run(args, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append((json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]), open(vol, "rb").read()))
    a, b = outs
    assert a[0]["triangles"] > 10000 and a[0]["render_hits"] > 1000 and a[0]["getFxn_ok"] == 1
    for k in ("triangles", "render_hits", "getFxn_ok", "getFxn", "cloud_bytes"):
        assert a[0][k] == b[0][k], k
    assert a[1] == b[1] and b[0]["devices"] == 3

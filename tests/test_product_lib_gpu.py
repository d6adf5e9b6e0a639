"""GPU tier: the PRODUCT library (cpu_tsdf_amd/lib/libtsdf_hip.so -- no test hooks, the one the C++ drop-in links and
bench.py / smoke() load) against the oracle, in a fresh interpreter (this pytest process runs on libtsdf_hip_test.so, the
same sources + hooks, see tests/conftest.py).  Three launches shapes without any knob: the whole grid in view (ALLIN
instance), a camera inside the grid (row intervals), a principal point far off centre (the reference's frustum cull decides
voxels: tsdf_volume_octree.cpp:619-652); then renderView and the mesh."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import ctypes as C, numpy as np, sys
sys.path.insert(0, %(root)r)
from cpu_tsdf_amd import capi, synth
from cpu_tsdf_amd.volume import TSDFVolumeOctree, MarchingCubesTSDFOctree
from oracle.oracle import OracleVolume
from tests.common import make_volume, assert_same_f32
lib = capi.load()
assert capi.LIB_PATH == capi.PRODUCT_LIB_PATH and not capi.has_test_hooks()
info = (C.c_int32 * 4)()
seen = []
for name, off in (("centred", 0.0), ("off-centre", 0.45)):
    vol, sc = make_volume(64, 160, 120, color=True)
    if off:
        sc.cx += off * 80
        vol.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
    vol.reset()
    ov = OracleVolume(vol._p)
    poses = [synth.turntable_pose(i, 8, sc.size) for i in range(3)] + [synth.look_at_pose((0.01, 0.0, -0.02), target=(0.03, 0.0, 1.0)),
                                                                      synth.turntable_pose(1, 8, sc.size, radius_factor=1.2)]
    for i, tr in enumerate(poses):
        dep, col = sc.depth(tr, noise_seed=3 + i), sc.bgra(i)
        got = vol.integrateCloud(dep, col, tr, count=True)
        want = ov.integrate_culled(dep, col, tr, synth.cam_from_vol_f32(tr))
        assert got == want, (name, i, got, want)
        capi.check(lib.tsdf_hip_last_launch_info(vol._need(), info), "info")
        seen.append((info[0], info[2]))
    d, w, rgb = vol.download()
    assert_same_f32(d, ov.d, name + ": d")
    assert np.array_equal(w, ov.w) and np.array_equal(rgb, ov.rgb)
    img = vol.renderView(poses[1], 1, camera_frame=False)
    assert_same_f32(img[..., :6], ov.raycast(poses[1], 1)[..., :6], name + ": renderView")
    mc = MarchingCubesTSDFOctree()
    mc.setInputTSDF(vol)
    mc.setMinWeight(0.0)
    mc.setColorByRGB(True)
    mesh = mc.reconstruct()
    verts, cols = ov.march(0.0, 1)[:2]
    assert len(verts) > 1000
    assert_same_f32(mesh["vertices"], verts, name + ": mesh")
    assert np.array_equal(mesh["rgb"], cols)
    vol.close()
assert (1, 0) in seen and any(s[1] for s in seen), seen   # the ALLIN instance and a row-interval launch both ran
print("PRODUCT_OK", seen)
'''


def test_the_product_library_without_hooks_equals_the_oracle(gpu):
    env = {k: v for k, v in os.environ.items() if k != "TSDF_HIP_LIB_PATH"}
    env["PYTHONPATH"] = ROOT
    out = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=env, text=True, capture_output=True, timeout=600)
    assert out.returncode == 0 and "PRODUCT_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]

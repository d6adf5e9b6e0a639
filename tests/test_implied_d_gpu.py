"""GPU tier: distances the kernel does not read (k_integrate / k_integrate2 `s_bin`).

In the PACKED layout a cell of 64 x 4 x 1 voxels that no frame since the reset has observed inside the truncation band
(its "band seen" flag is still 0) holds only two distances: the reset value -1 where the count is 0 and the hinge value
max_dist_pos / max_dist_neg elsewhere -- updateVoxel (include/cpu_tsdf/impl/tsdf_volume_octree.hpp:189-198) hands
addObservation (src/lib/octree.cpp:152-163) that constant for every free-space observation and (p*w + p)/(w + 1) == p.
The kernels rebuild such distances from the counts instead of loading them.  Nothing may change: every case below runs
the same frames with the knob on and off and against the CPU oracle, bit for bit, and checks through
tsdf_hip_last_read_detail that the path under test was really taken (or really refused)."""
import ctypes as C

import numpy as np
import pytest

from cpu_tsdf_amd import capi, synth
from oracle.oracle import OracleVolume
from tests.common import assert_same_f32, frames, make_volume
from tests.test_fused2_gpu import device_frame

pytestmark = pytest.mark.gpu


def read_detail(vol):
    out = (C.c_uint64 * 3)()
    capi.check(capi.load().tsdf_hip_last_read_detail(vol._need(), out), "last_read_detail")
    assert int(out[2]) >= 4 * (int(out[0]) > 0)  # the launch requested plane bytes
    return int(out[0]), int(out[1])


def launch_info(vol):
    out = (C.c_int32 * 4)()
    capi.check(capi.load().tsdf_hip_last_launch_info(vol._need(), out), "last_launch_info")
    return [int(out[0]) & 0xff, int(out[1]), int(out[2]), int(out[3]), bool(int(out[0]) & 0x100)]  # [4]: the pipelined row loop (k_integrate_p)


def compare(vol, ov):
    d, w, rgb = vol.download()
    assert_same_f32(d, ov.d, "d")
    assert_same_f32(w, ov.w, "w")
    if ov.rgb is not None:
        assert np.array_equal(rgb, ov.rgb)
    return d, w, rgb


def open_scene(sc):
    """Scene A with the box's walls OUTSIDE the grid: only the sphere's surface is ever seen inside the truncation band, the
    rest of the volume is free space (rows that never meet a flagged cell: most of them)."""
    return synth.Scene(sc.size, sc.width, sc.height, box=0.6)


def holes(dep, i):
    dep = dep.copy()
    dep[(i * 7) % 50::53, ::3] = np.nan
    return dep


@pytest.mark.parametrize("color,wmax,kind", [(True, 100.0, "allin"), (False, 100.0, "allin"), (True, 3.0, "general"),
                                             (False, 2.0, "general"), (True, 255.0, "allin"), (False, 1.0, "allin")])
def test_distances_rebuilt_from_counts_change_nothing(gpu, color, wmax, kind):
    """The turntable (every voxel in view: ALLIN instance, and the general one with the knob "allin" off), noisy depth with
    NaN holes, through weight saturation: knob on == knob off == oracle, counts included; with the knob on most observed
    voxels are not read (free space), with it off none is skipped."""
    outs = []
    try:
        capi.set_tuning("allin", 1 if kind == "allin" else 0)
        for on in (1, 0):
            capi.set_tuning("implied_d", on)
            vol, sc = make_volume(96, color=color, max_weight=wmax)
            sc = open_scene(sc)
            vol.reset()
            assert vol.getLayout() == capi.LAYOUT_PACKED
            ov = OracleVolume(vol._p)
            skipped = []
            for i, tr, dep, col in frames(sc, 7, 9, noise=True):
                dep = holes(dep, i)
                n = vol.integrateCloud(dep, col if color else None, tr, count=True)
                assert launch_info(vol)[0] == (1 if kind == "allin" else 0)
                assert n == ov.integrate(dep, col if color else None, synth.cam_from_vol_f32(tr))
                k, allowed = read_detail(vol)
                assert allowed == on and 0 <= k <= n
                skipped.append(k / n)
            if on:
                assert min(skipped) > 0.5, skipped  # free space is most of what a frame observes
            else:
                assert max(skipped) == 0
            outs.append(compare(vol, ov))
            vol.close()
    finally:
        capi.set_tuning("implied_d", 1)
        capi.set_tuning("allin", 1)
    assert_same_f32(outs[0][0], outs[1][0], "d: implied vs read")
    assert np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("rows_per_block,live_log2tx", [(64, 5), (4, 4), (16, 6), (256, 8), (8, 7)])
def test_implied_distances_with_row_intervals_and_every_block_shape(gpu, rows_per_block, live_log2tx):
    """Camera inside the volume (Scene B: row intervals + block flags, launches on a sub-box of the grid, narrow blocks) at
    several block shapes: a flag cell must mean the same voxels to every launch whatever its shape and origin."""
    sc = synth.scene_b(160, 120)
    try:
        capi.set_tuning("rows_per_block", rows_per_block)
        capi.set_tuning("live_log2tx", live_log2tx)
        vol, _ = make_volume(128, 160, 120, color=True, size=10.0, zmin=0.0, zmax=3.0)
        vol.reset()
        ov = OracleVolume(vol._p)
        took = 0
        for i in range(6):
            tr = synth.scene_b_pose(i % 4, 4)  # revisits: free space observed again with counts > 0
            dep, col = sc.depth(tr, noise_seed=5 + i), sc.bgra(i)
            n = vol.integrateCloud(dep, col, tr, count=True)
            assert n == ov.integrate_culled(dep, col, tr, synth.cam_from_vol_f32(tr))
            k, allowed = read_detail(vol)
            assert allowed == 1 and k <= n
            took += k
            # the other shape in between: flags written by one shape are read by the other
            capi.set_tuning("live_log2tx", 5 if i % 2 == 0 else live_log2tx)
        assert took > 0
        compare(vol, ov)
        vol.close()
    finally:
        capi.set_tuning("rows_per_block", 64)
        capi.set_tuning("live_log2tx", 5)


def test_implied_distances_are_refused_when_the_planes_may_hold_anything(gpu):
    """An upload hands the volume arbitrary distances: the flags stop describing the planes and every later launch reads
    its distance words (allowed == 0), still equal to the oracle.  A reset restores the record."""
    vol, sc = make_volume(64, color=True)
    sc = open_scene(sc)
    vol.reset()
    ov = OracleVolume(vol._p)
    fr = list(frames(sc, 6, 8, noise=True))
    for i, tr, dep, col in fr[:2]:
        vol.integrateCloud(dep, col, tr, count=True)
        ov.integrate(dep, col, synth.cam_from_vol_f32(tr))
        assert read_detail(vol)[1] == 1
    d, w, rgb = vol.download()
    # a free-space voxel moved off the hinge value, where no flag is set: only reading it can tell
    free = np.argwhere((w > 0) & (d == d.max()))
    z, y, x = free[len(free) // 2]
    d[z, y, x] = 0.25
    ov.d[z, y, x] = 0.25
    vol.upload(d=d, w=w, rgb=rgb)
    for i, tr, dep, col in fr[2:]:
        n = vol.integrateCloud(dep, col, tr, count=True)
        assert n == ov.integrate(dep, col, synth.cam_from_vol_f32(tr))
        assert read_detail(vol) == (0, 0)
    got = compare(vol, ov)
    assert got[0][z, y, x] != d.max()
    vol.reset()
    ov = OracleVolume(vol._p)
    for i, tr, dep, col in fr[:3]:
        vol.integrateCloud(dep, col, tr, count=True)
        ov.integrate(dep, col, synth.cam_from_vol_f32(tr))
        assert read_detail(vol)[1] == 1
    compare(vol, ov)
    vol.close()


@pytest.mark.parametrize("wmax,layout,trunc", [(2.5, capi.LAYOUT_AUTO, (0.03, 0.03)), (100.0, capi.LAYOUT_F32W, (0.03, 0.03)),
                                               (0.0, capi.LAYOUT_PACKED, (0.03, 0.03))])
def test_implied_distances_need_the_packed_layout_and_a_fixed_hinge(gpu, wmax, layout, trunc):
    """A non-integer max_weight (the hinge identity is not checked for it), float weights, a count that never leaves 0:
    the launches read every distance word."""
    vol, sc = make_volume(64, color=True, max_weight=wmax, trunc=trunc)
    vol.setLayout(layout)
    vol.reset()
    ov = OracleVolume(vol._p)
    for i, tr, dep, col in frames(sc, 4, 6, noise=True):
        n = vol.integrateCloud(dep, col, tr, count=True)
        assert n == ov.integrate(dep, col, synth.cam_from_vol_f32(tr))
        assert read_detail(vol) == (0, 0)
    compare(vol, ov)
    vol.close()


@pytest.mark.parametrize("trunc", [(0.03, 0.03), (0.05, 0.02), (0.01, 0.03), (0.07, 0.011)])
def test_implied_distances_at_hinge_values_other_than_one(gpu, trunc):
    """p = max_dist_pos / max_dist_neg != 1: allowed exactly when the host found (p*w + p)/(w + 1) == p for every weight;
    the voxels equal the oracle's either way."""
    vol, sc = make_volume(80, color=False, max_weight=30.0, trunc=trunc)
    sc = open_scene(sc)
    vol.reset()
    ov = OracleVolume(vol._p)
    p = np.float32(trunc[0]) / np.float32(trunc[1])
    fixed = all(np.float32(np.float32(p * np.float32(k)) + p) / np.float32(k + 1) == p for k in range(31))
    for i, tr, dep, col in frames(sc, 6, 6, noise=True):
        n = vol.integrateCloud(holes(dep, i), None, tr, count=True)
        assert n == ov.integrate(holes(dep, i), None, synth.cam_from_vol_f32(tr))
        assert read_detail(vol)[1] == int(fixed), (trunc, fixed)
    compare(vol, ov)
    vol.close()


@pytest.mark.parametrize("color", [True, False])
def test_two_frames_per_sweep_with_implied_distances(gpu, color):
    """k_integrate2 between single frames with the knob on and off (the flags one kernel writes are the ones the other reads).
    Since round 5 k_integrate2 itself reads every distance word (the shortcut is compiled out of it: DESIGN 3.1b) while still
    keeping the flags and the host's record, so the single frames around it go on skipping."""
    outs = []
    try:
        capi.set_tuning("fuse2", 2)  # k_integrate2 also without colour (the default leaves those pairs to the pipelined kernel)
        for on in (1, 0):
            capi.set_tuning("implied_d", on)
            vol, sc = make_volume(96, color=color, max_weight=6.0)
            sc = open_scene(sc)
            vol.reset()
            ov = OracleVolume(vol._p)
            fr = list(frames(sc, 10, 10, noise=True))
            keep = []
            k = 0
            while k < len(fr):
                single = k % 3 == 2
                if single:  # a single frame between pairs
                    i, tr, dep, col = fr[k]
                    dep = holes(dep, i)
                    n = vol.integrateCloud(dep, col if color else None, tr, count=True)
                    assert n == ov.integrate(dep, col if color else None, synth.cam_from_vol_f32(tr))
                    k += 1
                else:
                    if k + 1 >= len(fr):
                        break
                    pair, want = [], []
                    for i, tr, dep, col in fr[k:k + 2]:
                        dep = holes(dep, i)
                        t = device_frame(dep, col if color else None)
                        keep.append(t)
                        pair.append((t[0].data_ptr(), t[1].data_ptr() if color else 0, tr))
                        want.append(ov.integrate(dep, col if color else None, synth.cam_from_vol_f32(tr)))
                    fused, counts = vol.integrateCloudDevice2(pair[0], pair[1], count=True)
                    assert fused and counts == want
                    k += 2
                skipped, allowed = read_detail(vol)
                if single:
                    assert allowed == on and (skipped > 0) == bool(on)
                else:
                    assert allowed == 0 and skipped == 0
            outs.append(compare(vol, ov))
            vol.close()
    finally:
        capi.set_tuning("fuse2", 1)
        capi.set_tuning("implied_d", 1)
    assert_same_f32(outs[0][0], outs[1][0], "d: implied vs read")
    assert np.array_equal(outs[0][1], outs[1][1])


def test_marching_cubes_sees_the_same_flags(gpu):
    """The flags have a second reader (k_mc_classify skips what no set flag is near): the mesh after implied-distance
    launches equals the mesh of the knob-off run triangle for triangle."""
    from cpu_tsdf_amd.volume import MarchingCubesTSDFOctree
    meshes = []
    try:
        for on in (1, 0):
            capi.set_tuning("implied_d", on)
            vol, sc = make_volume(96, color=True, max_weight=20.0)
            sc = open_scene(sc)
            vol.reset()
            for i, tr, dep, col in frames(sc, 8, 8, noise=True):
                vol.integrateCloud(dep, col, tr)
            mc = MarchingCubesTSDFOctree()
            mc.setInputTSDF(vol)
            mc.setMinWeight(0.0)
            mc.setColorByRGB(True)
            m = mc.reconstruct()
            meshes.append((np.array(m["vertices"], dtype=np.float32), np.array(m["rgb"])))
            vol.close()
    finally:
        capi.set_tuning("implied_d", 1)
    assert meshes[0][0].shape == meshes[1][0].shape and len(meshes[0][0]) > 1000
    assert np.array_equal(meshes[0][0].view(np.uint32), meshes[1][0].view(np.uint32))
    assert np.array_equal(meshes[0][1], meshes[1][1])

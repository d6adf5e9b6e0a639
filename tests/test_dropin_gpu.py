"""GPU tier: the C++ drop-in shell (include/cpu_tsdf/*.h + libcpu_tsdf_hip.so) driven through the SAME
C driver source that wraps the reference (oracle/ref_capi.cpp, compiled once against each).

What this proves: code written against the reference's public C++ API (setters, reset, templated
integrateCloud on a pcl::PointCloud<PointXYZRGBA>, renderView, getFxn/getGradient/getHessian,
MarchingCubesTSDFOctree::reconstruct, save/load) compiles unchanged against the drop-in and produces
the reference's results."""
import os

import numpy as np
import pytest

from cpu_tsdf_amd import synth
from oracle import refbind
from oracle.oracle import OracleVolume
from tests.common import assert_same_f32
from tests.test_oracle_golden import params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dropin(gpu):
    lib = refbind.DROPIN_LIB if os.path.exists(refbind.DROPIN_LIB) else refbind.build_dropin()
    return lib


def make_pair(dropin, res=64, W=160, H=120, color=True, n_frames=5, devices=None):
    sc = synth.scene_a(res, W, H)
    dv = refbind.RefVolume(res, sc.size, W, H, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size, color=color,
                           lib_path=dropin, devices=devices)
    ov = OracleVolume(params(res, W, H, sc.size, color))
    for i in range(n_frames):
        tr = synth.turntable_pose(i, 8, sc.size, tilt=0.05 * i)
        dep, col = sc.depth(tr), sc.bgra(i)
        dv.integrate(dep, col, tr)
        ov.integrate(dep, col if color else None, synth.cam_from_vol_f32(tr))
    return dv, ov, sc


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
def test_dropin_integrate_render_sample_mesh(dropin, devices):
    """devices = [0, 0, 0]: the same through TSDFVolumeOctree::setDevices -- three Z-slab handles behind the C++ class
    (on the one GPU gpurun has), i.e. the multi-GPU path as a user of the reference API reaches it."""
    from cpu_tsdf_amd.volume import transform_cloud_with_normals
    dv, ov, sc = make_pair(dropin, devices=devices)
    d, w, rgb = dv.download()
    assert_same_f32(d, ov.d, "d")
    assert_same_f32(w, ov.w, "w")
    assert np.array_equal(rgb, ov.rgb)
    for tr, ds in [(synth.turntable_pose(1, 8, sc.size), 1), (synth.look_at_pose((0.3, -0.2, -0.25)), 2)]:
        got, _ = dv.render_view(tr, ds)
        want = transform_cloud_with_normals(ov.raycast(tr, ds), synth.eigen_affine_inverse(tr))
        assert np.isfinite(want[..., 0]).sum() > 100
        assert_same_f32(got[..., :6], want[..., :6], "renderView (camera frame)")
    rng = np.random.RandomState(5)
    pts = rng.uniform(-0.13, 0.13, (500, 3)).astype(np.float32)
    ok, val, grad, hess = dv.sample(pts)
    ok2, val2, grad2, hess2 = ov.sample(pts)
    assert np.array_equal(ok == 7, ok2) and set(np.unique(ok)) <= {0, 7}
    assert_same_f32(val[ok2], val2[ok2], "getFxn")
    assert_same_f32(grad[ok2], grad2[ok2], "getGradient")
    assert_same_f32(hess[ok2], hess2[ok2], "getHessian")
    for mode in (0, 1, 2):
        v, c, polys, _ = dv.march(2.0, mode)
        v2, c2, _ = ov.march(2.0, mode)
        assert len(v) > 10000
        assert_same_f32(v, v2, "mesh vertices")
        assert np.array_equal(polys.ravel(), np.arange(len(v), dtype=np.uint32))
        if mode:
            assert np.array_equal(c, c2)
    dv.close()


@pytest.mark.parametrize("color", [True, False])
def test_dropin_render_colored_view_equals_reference(dropin, color):
    """renderColoredView (tsdf_volume_octree.cpp:426-450): the drop-in's batched device lookup
    (tsdf_hip_lookup_rgb = getContainingVoxel + getRGB per hit) vs the reference's own method on a dense octree
    fed the same frames.  Without integrateColor the reference's octree is "NOCOLOR": every found voxel answers
    127,127,127 (OctreeNode::getRGB)."""
    if not refbind.available():
        pytest.skip("oracle/_ref/libcpu_tsdf_ref.so not built")
    res, W, H = 64, 160, 120
    sc = synth.scene_a(res, W, H)
    vols = [refbind.RefVolume(res, sc.size, W, H, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size, color=color, lib_path=lp)
            for lp in (dropin, refbind.LIB)]
    from tests.common import make_volume
    pyvol, _ = make_volume(res, W, H, color=color)  # the Python binding of the same C ABI
    pyvol.reset()
    for i in range(5):
        tr = synth.turntable_pose(i, 8, sc.size, tilt=0.05 * i)
        for v in vols:
            v.integrate(sc.depth(tr), sc.bgra(i), tr)
        pyvol.integrateCloud(sc.depth(tr), sc.bgra(i) if color else None, tr)
    for tr, ds in [(synth.turntable_pose(1, 8, sc.size), 1), (synth.look_at_pose((0.3, -0.2, -0.25)), 2)]:
        (got, got_rgb), (want, want_rgb) = (v.render_colored_view(tr, ds) for v in vols)
        py_cloud, py_rgb = pyvol.renderColoredView(tr, ds)
        assert_same_f32(py_cloud[..., :6], want[..., :6], "Python renderColoredView geometry")
        assert np.array_equal(py_rgb[np.isfinite(want[..., 0])], want_rgb[np.isfinite(want[..., 0])])
        hits = np.isfinite(want[..., 0])
        assert hits.sum() > 100
        assert_same_f32(got[..., :6], want[..., :6], "renderColoredView geometry")
        assert np.array_equal(got_rgb[hits], want_rgb[hits]), "colour of the voxel containing each hit"
        if color:
            assert len(np.unique(want_rgb[hits], axis=0)) > 20
        else:
            assert set(np.unique(want_rgb[hits]).tolist()) <= {0, 127} and (want_rgb[hits] == 127).mean() > 0.9
    for v in vols:
        v.close()


@pytest.fixture
def vol_chunk():
    """Set the block edge save / load stream with (tsdf_hip_set_tuning "vol_chunk"); restored afterwards."""
    from cpu_tsdf_amd import capi
    yield lambda n: capi.set_tuning("vol_chunk", n)
    capi.set_tuning("vol_chunk", 256)


@pytest.mark.parametrize("color,chunk", [(False, None), (True, None), (True, 8), (False, 4)])
def test_dropin_save_load_interop_with_reference(dropin, tmp_path, vol_chunk, color, chunk):
    """.vol files cross both ways: the drop-in's save() is readable by the reference's load(), and the
    reference's save() by the drop-in's load(); voxels survive bit for bit.  save / load stream the grid
    through host memory in blocks; `chunk` forces blocks smaller than this small grid (tsdf_hip_download /
    tsdf_hip_upload on sub-boxes)."""
    if not refbind.available():
        pytest.skip("oracle/_ref not built")
    if chunk:
        vol_chunk(chunk)
    dv, ov, sc = make_pair(dropin, res=32, W=80, H=60, color=color, n_frames=3)
    ours = str(tmp_path / "dropin.vol")
    dv.save(ours)
    full_tree = sum(8 ** l for l in range(6)) * (43 if color else 40)
    assert os.path.getsize(ours) < full_tree + 2000
    ref = refbind.RefVolume(32, sc.size, 80, 60, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size, color=color)
    ref.load(ours)
    d, w, rgb, leaf, _ = ref.dump_dense()
    assert_same_f32(d, ov.d, "reference reading the drop-in's file: d")
    assert_same_f32(w, ov.w, "w")
    if color:
        assert np.array_equal(rgb, ov.rgb)
    else:  # free space (d = 1, same w) and unobserved regions collapse into coarse leaves
        assert os.path.getsize(ours) < 0.9 * full_tree and leaf.max() > leaf.min()
    # and a mesh made by the reference from our file equals ours
    v_ref, _, _, _ = ref.march(1.0, 0)
    v_our, _, _, _ = dv.march(1.0, 0)
    assert len(v_ref) > 1000
    assert_same_f32(v_ref, v_our, "mesh by the reference from the drop-in's file")
    theirs = str(tmp_path / "reference.vol")
    ref2 = refbind.RefVolume(32, sc.size, 80, 60, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size, color=color)
    for i in range(3):
        tr = synth.turntable_pose(i, 8, sc.size, tilt=0.05 * i)
        ref2.integrate(sc.depth(tr), sc.bgra(i), tr)
    ref2.save(theirs)
    dv2 = refbind.RefVolume(32, sc.size, 80, 60, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size, color=color,
                            lib_path=dropin)
    dv2.load(theirs)
    d2, w2, rgb2 = dv2.download()
    assert_same_f32(d2, ov.d, "drop-in reading the reference's file: d")
    assert_same_f32(w2, ov.w, "w")
    if color:
        assert np.array_equal(rgb2, ov.rgb)
    for v in (dv, dv2, ref, ref2):
        v.close()


def test_dropin_load_of_a_file_with_other_weights_rereads_into_float_weights(dropin, tmp_path, vol_chunk):
    """A .vol whose weights are not min(k, max_weight) (another weighting scheme wrote it) cannot live in the
    packed layout.  The streamed load() finds out at the first such block -- here the last of 64 -- and reads
    the file again into a float weight plane; every voxel must come back bit for bit."""
    import subprocess
    from tests.test_vol_stream import ROOT, RES, SIZE, grid, write_raw
    exe = str(tmp_path / "vol_stream")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-pthread", "-I", os.path.join(ROOT, "cpu_tsdf_amd", "csrc"),
                           os.path.join(ROOT, "tests", "harness", "vol_stream.cpp"), "-o", exe])
    d, w, rgb = grid(True, seed=9)
    w[RES - 1, RES - 1, RES - 1] = 2.5
    d[RES - 1, RES - 1, RES - 1] = 0.125
    raw, vol = str(tmp_path / "in.raw"), str(tmp_path / "odd.vol")
    write_raw(raw, d, w, rgb, True)
    subprocess.check_call([exe, "write", raw, str(RES), str(SIZE), "1", "32", vol], stdout=subprocess.DEVNULL)
    vol_chunk(8)
    dv = refbind.RefVolume(RES, SIZE, 640, 480, 525.0, 525.0, 319.5, 239.5, 0.0, 3.0, color=True, lib_path=dropin)
    dv.load(vol)
    d2, w2, rgb2 = dv.download()
    assert_same_f32(d2, d, "d")
    assert_same_f32(w2, w, "w")
    assert np.array_equal(rgb2, rgb)
    again = str(tmp_path / "again.vol")
    dv.save(again)
    assert open(again, "rb").read().split(b"#OCTREEBINARY")[1] == open(vol, "rb").read().split(b"#OCTREEBINARY")[1]
    dv.close()


def test_python_save_load_over_the_c_abi(dropin, tmp_path, vol_chunk):
    """tsdf_hip_save / tsdf_hip_load as every binding sees them (here ctypes): the file the Python class
    writes is the file the C++ class writes, the reference reads it, and load() restores voxels and metadata."""
    from cpu_tsdf_amd import capi
    from cpu_tsdf_amd.volume import TSDFVolumeOctree
    dv, ov, sc = make_pair(dropin, res=32, W=80, H=60, color=True, n_frames=3)
    by_cpp = str(tmp_path / "cpp.vol")
    dv.save(by_cpp)
    pv = TSDFVolumeOctree()
    pv.setResolution(32, 32, 32)
    pv.setGridSize(sc.size, sc.size, sc.size)
    pv.setImageSize(80, 60)
    pv.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
    pv.setSensorDistanceBounds(0.0, 3 * sc.size)
    pv.setIntegrateColor(True)
    pv.reset()
    for i in range(3):
        tr = synth.turntable_pose(i, 8, sc.size, tilt=0.05 * i)
        pv.integrateCloud(sc.depth(tr), sc.bgra(i), tr)
    G = synth.turntable_pose(2, 8, sc.size)
    pv.setGlobalTransform(G)
    vol_chunk(16)
    by_py = str(tmp_path / "py.vol")
    pv.save(by_py)
    tree = lambda path: open(path, "rb").read().split(b"#OCTREEBINARY")[1]
    assert tree(by_py) == tree(by_cpp)
    if refbind.available():
        ref = refbind.RefVolume(32, sc.size, 80, 60, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size, color=True)
        ref.load(by_py)
        d, w, rgb, _, _ = ref.dump_dense()
        assert_same_f32(d, ov.d, "reference reading the Python class's file")
        assert np.array_equal(w, ov.w) and np.array_equal(rgb, ov.rgb)
        ref.close()
    back = TSDFVolumeOctree()      # nothing configured: everything must come from the file
    back.load(by_py)
    assert back.getResolution() == (32, 32, 32) and back.getImageSize() == (80, 60) and not back.isEmpty()
    assert back.getLayout() == capi.LAYOUT_PACKED
    assert np.array_equal(back.getGlobalTransform(), G)
    d, w, rgb = back.download()
    assert_same_f32(d, ov.d, "d after load")
    assert np.array_equal(w, ov.w) and np.array_equal(rgb, ov.rgb)
    tr = synth.turntable_pose(3, 8, sc.size)     # and it keeps integrating like the original
    back.integrateCloud(sc.depth(tr), sc.bgra(3), tr)
    pv.integrateCloud(sc.depth(tr), sc.bgra(3), tr)
    assert all(np.array_equal(a, b) for a, b in zip(back.download(), pv.download()))
    # errors: a missing file, a file that is not a .vol, a Z-slab handle
    with pytest.raises(capi.TsdfHipError) as e:
        back.load(str(tmp_path / "nope.vol"))
    assert e.value.code == capi.E_IO and back.getResolution() == (32, 32, 32)   # the old volume is untouched
    junk = tmp_path / "junk.vol"
    junk.write_bytes(b"not a volume\n" * 10)
    with pytest.raises(capi.TsdfHipError) as e:
        back.load(str(junk))
    assert e.value.code == capi.E_IO
    slab = TSDFVolumeOctree()
    slab.setResolution(32, 32, 32)
    slab.setZSlab(0, 16)
    slab.reset()
    with pytest.raises(capi.TsdfHipError) as e:
        slab.save(str(tmp_path / "slab.vol"))
    assert e.value.code == capi.E_UNSUPPORTED
    for v in (pv, back, slab, dv):
        v.close()


def test_dropin_set_color_mode_rgb_normalized(dropin):
    """setColorMode("RGBNormalized") through the C++ classes: the drop-in against the reference's own library,
    driven by the same C driver (colours read back through getRGB / marching cubes)."""
    if not refbind.available():
        pytest.skip("oracle/_ref not built")
    res, W, H = 32, 80, 60
    sc = synth.scene_a(res, W, H)
    vols = [refbind.RefVolume(res, sc.size, W, H, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size, color=True,
                              color_mode="RGBNormalized", lib_path=lib) for lib in (dropin, refbind.LIB)]
    for i in range(4):
        tr = synth.turntable_pose(i, 8, sc.size, tilt=0.05 * i)
        col = sc.bgra(i).copy()
        col[25:35, 30:50, :3] = 0
        for v in vols:
            v.integrate(sc.depth(tr), col, tr)
    d, w, rgb = vols[0].download()
    rd, rw, rrgb, _, _ = vols[1].dump_dense()
    assert_same_f32(d, rd, "d")
    assert np.array_equal(w, rw) and np.array_equal(rgb, rrgb) and rgb.max() > 30
    meshes = [v.march(1.0, 1) for v in vols]
    assert len(meshes[0][0]) > 500
    assert_same_f32(meshes[0][0], meshes[1][0], "mesh")
    assert np.array_equal(meshes[0][1], meshes[1][1])
    for v in vols:
        v.close()


def test_drop_in_default_path_equals_the_reference_with_an_off_centre_camera(gpu):
    """The drop-in through the SAME C driver as the reference, calling nothing the reference does not have: a principal
    point 15 % of the half-width off centre makes pcl::FrustumCulling drop voxels at one image border
    (tsdf_volume_octree.cpp:619-652); cpu_tsdf::TSDFVolumeOctree::integrateCloud, building the six planes with its own
    Eigen, drops the same ones.  Also on a three-slab multi handle, and with the explicit
    setReferenceCull(true); setReferenceCull(false) is the opt-out and integrates the strict superset."""
    res, W, H, size = 32, 64, 48, 1.0
    fx = fy = 110.0
    cx, cy = W / 2 - 0.5 + 4.8, H / 2 - 0.5
    ref = refbind.RefVolume(res, size, W, H, fx, fy, cx, cy, 0.0, 3.0, color=True)
    drops = [refbind.RefVolume(res, size, W, H, fx, fy, cx, cy, 0.0, 3.0, color=True, lib_path=refbind.DROPIN_LIB,
                               reference_cull=rc, devices=dev) for rc, dev in ((None, None), (None, [0, 0, 0]), (True, None))]
    optout = refbind.RefVolume(res, size, W, H, fx, fy, cx, cy, 0.0, 3.0, color=True, lib_path=refbind.DROPIN_LIB, reference_cull=False)
    rng = np.random.RandomState(9)
    for i in range(4):
        tr = synth.look_at_pose(np.array([1.9, 0.2 * i, 0.3]) * size, target=np.zeros(3))
        dep = (rng.uniform(1.2, 2.6, (H, W)) * size).astype(np.float32)
        col = rng.randint(0, 256, (H, W, 4)).astype(np.uint8)
        for v in [ref, optout] + drops:
            v.integrate(dep, col, tr)
    d, w, rgb, _, _ = ref.dump_dense()
    for v in drops:
        gd, gw, grgb = v.download()
        assert_same_f32(gd, d, "d")
        assert np.array_equal(gw, w) and np.array_equal(grgb, rgb)
    pd, pw, _ = optout.download()
    assert (w <= pw).all() and (w < pw).sum() > 20  # without the cull: a strict superset
    for v in [ref, optout] + drops:
        v.close()


@pytest.mark.parametrize("color", [True, False])
def test_dropin_frame_pairing_equals_the_reference_frame_by_frame(dropin, color):
    """TSDFVolumeOctree::setFramePairing on the C++ drop-in (round 6: also on a set of devices): integrateCloud calls on
    pcl clouds are queued in the pinned ring and swept two per launch where both poses see the whole slab, a frame waiting
    for its partner is launched by the next query.  Against the COMPILED REFERENCE fed the same clouds one by one: an odd
    number of frames, a query in the middle (getFxn while a frame is parked), one and three slab handles."""
    res, W, H = 64, 160, 120
    sc = synth.scene_a(res, W, H)
    args = (res, sc.size, W, H, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size)
    ref = refbind.RefVolume(*args, color=color)
    drops = [refbind.RefVolume(*args, color=color, lib_path=dropin, frame_pairing=True, devices=dev) for dev in (None, [0, 0, 0])]
    pts = np.random.RandomState(2).uniform(-0.12, 0.12, (200, 3)).astype(np.float32)
    for i in range(7):
        tr = synth.turntable_pose(i, 9, sc.size, tilt=0.04 * i)
        dep, col = sc.depth(tr, noise_seed=70 + i), sc.bgra(i)
        for v in [ref] + drops:
            v.integrate(dep, col, tr)
        if i == 2:   # frame 2 is waiting for frame 3 in the drop-ins
            rok, rval, _, _ = ref.sample(pts)
            for v in drops:
                ok, val, _, _ = v.sample(pts)
                assert np.array_equal(ok, rok)
                m = (rok & 1).astype(bool)
                assert m.sum() > 50
                assert_same_f32(val[m], rval[m], "getFxn with a frame parked")
    d, w, rgb, _, _ = ref.dump_dense()
    assert (w > 0).sum() > 100000
    for v in drops:
        gd, gw, grgb = v.download()
        assert_same_f32(gd, d, "d")
        assert np.array_equal(gw, w) and (not color or np.array_equal(grgb, rgb))
    for v in [ref] + drops:
        v.close()


def test_dropin_extensions_through_the_environment(dropin, monkeypatch):
    """A binary written against the reference and only RE-LINKED against the drop-in cannot call setFramePairing or
    setDevices: CPU_TSDF_HIP_FRAME_PAIRING / CPU_TSDF_HIP_DEVICES switch them on for every volume it constructs.  Same
    voxels as the compiled reference; the setters still override."""
    res, W, H = 32, 80, 60
    sc = synth.scene_a(res, W, H)
    args = (res, sc.size, W, H, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size)
    ref = refbind.RefVolume(*args, color=True)
    plain = refbind.RefVolume(*args, color=True, lib_path=dropin)
    assert plain.L.ct_get_frame_pairing(plain.h) == 0 and plain.L.ct_get_num_devices(plain.h) == 0
    monkeypatch.setenv("CPU_TSDF_HIP_FRAME_PAIRING", "1")
    monkeypatch.setenv("CPU_TSDF_HIP_DEVICES", "0,0,0")
    env = refbind.RefVolume(*args, color=True, lib_path=dropin)
    assert env.L.ct_get_frame_pairing(env.h) == 1 and env.L.ct_get_num_devices(env.h) == 3
    over = refbind.RefVolume(*args, color=True, lib_path=dropin, devices=[0, 0])
    assert over.L.ct_get_num_devices(over.h) == 2
    monkeypatch.delenv("CPU_TSDF_HIP_FRAME_PAIRING")
    monkeypatch.delenv("CPU_TSDF_HIP_DEVICES")
    for i in range(5):
        tr = synth.turntable_pose(i, 7, sc.size)
        dep, col = sc.depth(tr, noise_seed=5 + i), sc.bgra(i)
        for v in (ref, plain, env, over):
            v.integrate(dep, col, tr)
    d, w, rgb, _, _ = ref.dump_dense()
    for v in (plain, env, over):
        gd, gw, grgb = v.download()
        assert_same_f32(gd, d, "d")
        assert np.array_equal(gw, w) and np.array_equal(grgb, rgb)
    for v in (ref, plain, env, over):
        v.close()


def test_dropin_refuses_the_queries_on_a_non_cubic_grid_size(dropin, capfd):
    """VERDICT r04 missing #3: under a non-cubic setGridSize the reference looks per-axis voxel indices
    (src/lib/tsdf_volume_octree.cpp:553-574) up in an octree that is a cube of edge size_x (src/lib/octree.cpp:244-266) -- a
    mixture of two geometries the flat grid does not reproduce.  integrateCloud still works (it replicates the cube leaf by
    leaf); renderView, getFxn and MarchingCubesTSDFOctree::reconstruct REFUSE with a PCL_ERROR-style message instead of
    answering for another geometry: an all-NaN view, false, an empty mesh."""
    sc = synth.scene_a(64, 160, 120)
    dv = refbind.RefVolume(64, sc.size, 160, 120, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size, color=True, lib_path=dropin)
    dv.L.ct_set_grid_size(dv.h, sc.size, 0.75 * sc.size, sc.size)
    dv.L.ct_reset(dv.h)
    tr = synth.turntable_pose(0, 8, sc.size)
    dv.integrate(sc.depth(tr), sc.bgra(0), tr)          # integrateCloud is served
    d, w, _ = dv.download()
    assert (w > 0).sum() > 1000
    capfd.readouterr()
    got, _ = dv.render_view(tr, 1)
    assert not np.isfinite(got[..., :3]).any()
    ok, _, _, _ = dv.sample(np.zeros((4, 3), np.float32))
    assert not ok.any()
    v, _, polys, _ = dv.march(0.0, 0)
    assert len(v) == 0 and len(polys) == 0
    err = capfd.readouterr().err
    assert err.count("is not a cube") >= 3 and "renderView" in err and "getFxn" in err and "reconstruct" in err
    dv.close()

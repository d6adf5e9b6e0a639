"""GPU tier: the product's programs (cpu_tsdf_amd/bin/integrate, tsdf2mesh -- cpu_tsdf_amd/csrc/prog/) against
the reference's own programs compiled from its unmodified sources (oracle/_ref/ref_integrate, ref_tsdf2mesh):
same input directory, same command line -> byte-identical mesh.ply.  Covers every ingest mode (unorganised /
organised clouds, world-frame clouds, units, zero -> NaN, text and binary poses, inverted poses), colour,
--flatten / --cleanup, ascii output, --num-frames, and tsdf2mesh on each other's volume files."""
import os
import shutil

import numpy as np
import pytest

from tests import sequence_util as su
from tests.test_programs import COMMON, H, W

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.path.exists(su.REF_INTEGRATE) and os.path.exists(su.OUR_INTEGRATE)),
                                 reason="programs not built")]

CASES = {
    "plain_color": (dict(), ["--color"]),
    "world_units_binary_poses": (dict(world=True, units=0.001, binary_poses=True), ["--world", "--cloud-units", 0.001, "--color"]),
    "zero_nans_nocolor": (dict(), ["--zero-nans"]),
    "organized_color": (dict(organized=True), ["--organized", "--color"]),
    # --cleanup drops face groups of <= 5 faces linked at 2 cm, --flatten merges vertices closer than 0.1 mm:
    # shrink the whole scene 20x (units) so that a 64^3 grid has 6 mm voxels and the surface stays connected
    "flatten_cleanup": (dict(), ["--flatten", "--cleanup", "--color", "--cloud-units", 0.05, "--pose-units", 0.05,
                                 "GEOMETRY", "--volume-size", 0.4, "--cell-size", 0.00625, "--max-cell-size", 0.00625,
                                 "--max-sensor-dist", 0.15, "--trunc-dist-pos", 0.015, "--trunc-dist-neg", 0.015,
                                 "--width", W, "--height", H]),
    "ascii_minweight_numframes": (dict(), ["--save-ascii", "--min-weight", 2, "--num-frames", 3, "--color"]),
    "sensor_range_trunc": (dict(), ["--max-sensor-dist", 2.6, "--min-sensor-dist", 0.4, "--pose-units", 1.0, "--color"]),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_integrate_program_matches_reference_program(gpu, case):
    kw, flags = CASES[case]
    d = su.digit_free_dir(case.replace("_", ""))
    try:
        su.make_sequence(os.path.join(d, "in"), n_frames=4, width=W, height=H, **kw)
        if "GEOMETRY" in flags:  # the case brings its own grid
            args = ["--in", os.path.join(d, "in")] + [f for f in flags if f != "GEOMETRY"]
        else:
            args = ["--in", os.path.join(d, "in")] + COMMON + flags
        rc_ref, log_ref = su.run(su.REF_INTEGRATE, args + ["--out", os.path.join(d, "ref")])
        rc_our, log_our = su.run(su.OUR_INTEGRATE, args + ["--out", os.path.join(d, "our")])
        assert rc_ref == 0, log_ref[-2000:]
        assert rc_our == 0, log_our[-2000:]
        a = open(os.path.join(d, "ref", "mesh.ply"), "rb").read()
        b = open(os.path.join(d, "our", "mesh.ply"), "rb").read()
        v, c, f = su.read_ply(os.path.join(d, "ref", "mesh.ply"))
        assert len(f) > 200, "the scene must produce a real mesh"
        if a != b:
            v2, c2, f2 = su.read_ply(os.path.join(d, "our", "mesh.ply"))
            raise AssertionError(f"{case}: mesh.ply differs: ref {v.shape}/{f.shape} ours {v2.shape}/{f2.shape}; "
                                 f"first vertex diff {np.argwhere(v[:min(len(v), len(v2))] != v2[:min(len(v), len(v2))])[:3].tolist()}")
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_inverted_poses_and_pose_units(gpu):
    """--invert: the files hold world -> camera; --pose-units scales the translation after the inversion."""
    d = su.digit_free_dir("invert")
    try:
        sc = su.make_sequence(os.path.join(d, "in"), n_frames=3, width=W, height=H)
        from cpu_tsdf_amd import synth
        for i in range(3):  # overwrite the pose files by their inverses, translation in millimetres
            T = synth.eigen_affine_inverse(synth.scene_b_pose(i, 3))
            T[:3, 3] *= 1000.0
            su.write_pose(os.path.join(d, "in", f"cloud_{i:04d}.txt"), T, False)
        # NB the reference scales the translation of the INVERTED pose (src/prog/integrate.cpp:464-467), so a
        # file in millimetres only round-trips for pure translations; here the point is equality, not geometry
        args = ["--in", os.path.join(d, "in")] + COMMON + ["--invert", "--pose-units", 0.001, "--color"]
        assert su.run(su.REF_INTEGRATE, args + ["--out", os.path.join(d, "ref")])[0] == 0
        assert su.run(su.OUR_INTEGRATE, args + ["--out", os.path.join(d, "our")])[0] == 0
        assert open(os.path.join(d, "ref", "mesh.ply"), "rb").read() == open(os.path.join(d, "our", "mesh.ply"), "rb").read()
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_tsdf2mesh_on_each_others_volume(gpu):
    d = su.digit_free_dir("tsdfmesh")
    try:
        su.make_sequence(os.path.join(d, "in"), n_frames=4, width=W, height=H)
        args = ["--in", os.path.join(d, "in")] + COMMON + ["--color", "--save-tsdf"]
        assert su.run(su.REF_INTEGRATE, args + ["--out", os.path.join(d, "ref")])[0] == 0
        assert su.run(su.OUR_INTEGRATE, args + ["--out", os.path.join(d, "our")])[0] == 0
        meshes = {}
        for prog, tag in ((su.REF_TSDF2MESH, "ref"), (su.OUR_TSDF2MESH, "our")):
            for vol in ("ref", "our"):
                out = os.path.join(d, f"{tag}_from_{vol}.ply")
                rc, log = su.run(prog, [os.path.join(d, vol, "volume.tsdf"), out])
                assert rc == 0, log[-2000:]
                meshes[(tag, vol)] = open(out, "rb").read()
        ref = meshes[("ref", "ref")]
        assert len(ref) > 5000
        for k, m in meshes.items():
            assert m == ref, f"tsdf2mesh {k[0]} on the {k[1]} volume differs from the reference on its own volume"
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_program_usage_and_missing_pose(gpu):
    rc, log = su.run(su.OUR_INTEGRATE, ["--help"])
    assert rc == 1 and "--in" in log
    rc, _ = su.run(su.OUR_INTEGRATE, ["--out", "/tmp/nowhere"])  # --in is required
    assert rc == 1
    d = su.digit_free_dir("nopose")
    try:
        su.make_sequence(os.path.join(d, "in"), n_frames=2, width=W, height=H)
        os.remove(os.path.join(d, "in", "cloud_0001.txt"))
        rc_ref, _ = su.run(su.REF_INTEGRATE, ["--in", os.path.join(d, "in"), "--out", os.path.join(d, "ref")] + COMMON)
        rc_our, log = su.run(su.OUR_INTEGRATE, ["--in", os.path.join(d, "in"), "--out", os.path.join(d, "our")] + COMMON)
        assert rc_ref == rc_our == 1 and "matching transform" in log
    finally:
        shutil.rmtree(d, ignore_errors=True)

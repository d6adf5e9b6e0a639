"""CPU tier: the ray hand-off protocol (tsdf_hip_raycast_begin / advance, restated in oracle_raycast_advance) on random
small configurations -- grid 32..64, 2..5 slabs, halo from tsdf_hip_render_halo, cameras all around and inside -- run
sequentially in one process on oracle-backed slabs (tests/fake_slab.py) the way ZSlabVolume drives its ranks.  Every
image must equal the single-volume ray loop bit for bit, within world + 2 rounds, and no slab may read a plane it
does not hold.  (This is the search that found the far-extrapolated hit point: tests/test_zslab_gpu.py pins that case on
the HIP kernel.)"""
import numpy as np
import torch

from cpu_tsdf_amd import synth
from cpu_tsdf_amd.zslab import render_halo, slab_range
from oracle.oracle import OracleVolume
from tests.fake_slab import OracleSlab, _Cfg


def test_random_small_configurations():
    rng = np.random.RandomState(77)
    finish_hops = views = 0
    for case in range(60):
        res = int(rng.choice([32, 40, 64]))
        world = int(rng.randint(2, 6))
        W, H = 64, 48
        sc = synth.scene_a(res, W, H)
        # asymmetric truncation too: the hinge value max_dist_pos / max_dist_neg then exceeds 1 and free-space steps are
        # max_dist_pos long, which the halo has to cover (tsdf_hip_render_halo once assumed |d| <= 1)
        pos, neg = [(0.03, 0.03), (0.012, 0.012), (0.09, 0.02), (0.02, 0.05)][rng.randint(4)]

        def configure(v):
            v.setResolution(res, res, res)
            v.setGridSize(sc.size, sc.size, sc.size)
            v.setImageSize(W, H)
            v.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
            v.setSensorDistanceBounds(0.0, 3 * sc.size)
            v.setDepthTruncationLimits(pos, neg)
            v.setIntegrateColor(False)
        halo = render_halo(configure)
        cuts = [slab_range(res, world, r) for r in range(world)]
        if min(ze - zb for zb, ze in cuts) < 1:
            continue
        cfg = _Cfg()
        configure(cfg)
        whole = OracleVolume(cfg._p)
        slabs = [OracleSlab(configure, zb, ze, res, r, halo=halo) for r, (zb, ze) in enumerate(cuts)]
        fd, _ = slabs[0].frame_buffers()
        nf = int(rng.randint(2, 5))
        for i in range(nf):
            tr = synth.turntable_pose(i, nf, sc.size, tilt=float(rng.uniform(-0.4, 0.4)))
            dep = sc.depth(tr, noise_seed=int(rng.randint(1 << 30)) if case % 2 else None)
            whole.integrate(dep, None, synth.cam_from_vol_f32(tr))
            fd.copy_(torch.from_numpy(dep))
            for s in slabs:
                s.integrate_tensor(fd, None, tr)
        for r, s in enumerate(slabs):  # halo refresh: every slab takes what its neighbours own of its halo range
            for nb in range(world):
                if nb == r:
                    continue
                zb, ze = cuts[nb]
                lo, hi = max(zb, s.z_begin - halo), min(ze, s.z_end + halo)
                if lo < hi:
                    s.set_planes(lo, *slabs[nb].get_planes(lo, hi - lo))
        for _ in range(3):
            eye = rng.uniform(-2.4, 2.4, 3) * sc.size * (1.0 if rng.rand() < 0.7 else 0.2)
            tr = synth.look_at_pose(eye, target=rng.uniform(-0.2, 0.2, 3) * sc.size)
            want = whole.raycast(tr, 1)
            state = slabs[0].ray_begin(tr, 1)
            for rounds in range(1, world + 4):
                delta = sum(s.ray_advance(tr, 1, state, r, world) for r, s in enumerate(slabs))   # raises on a read outside a halo
                finish_hops += int(((delta[:, 0] == 1) & (delta[:, 12] == 1)).sum())
                state = torch.where(delta[:, :1] != 0, delta, state)
                if int((state[:, 0] == 1).sum()) == 0:
                    break
            assert int((state[:, 0] == 2).sum()) == state.shape[0], (case, res, world)
            assert rounds <= world + 2, (case, res, world, rounds)
            have = state[:, 16:24].contiguous().numpy().view(np.float32).reshape(want.shape)
            same = (have.view(np.uint32) == want.view(np.uint32)) | (np.isnan(have) & np.isnan(want))
            assert same.all(), (case, res, world, int((~same).sum()))
            views += 1
    assert views >= 120 and finish_hops >= 5   # far-extrapolated hit points did occur and were finished by their owners

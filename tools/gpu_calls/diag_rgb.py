import sys, numpy as np
sys.path.insert(0, '/root/repo')
from cpu_tsdf_amd import synth, capi
from cpu_tsdf_amd.volume import TSDFVolumeOctree
from oracle.oracle import OracleVolume
from tests.test_zslab_hip_ranks_gpu import configure, frames, RES, W, H
sc = synth.scene_a(RES, W, H)
one = TSDFVolumeOctree(); configure(one); one.reset()
ov = OracleVolume(one._p)
for i, tr, dep, col in frames(sc):
    one.integrateCloud(dep, col, tr)
    ov.integrate(dep, col, synth.cam_from_vol_f32(tr))
    d, w, c = one.download()
    ne = (c != ov.rgb).any(-1)
    print("frame", i, "rgb mismatches", int(ne.sum()), "d", int((d.view(np.uint32) != ov.d.view(np.uint32)).sum()), "w", int((w != ov.w).sum()))
    if ne.any():
        idx = np.argwhere(ne)[:8]
        for z, y, x in idx:
            print("  voxel", z, y, x, "got", c[z, y, x], "want", ov.rgb[z, y, x], "w", w[z, y, x], "d", d[z, y, x])
        break

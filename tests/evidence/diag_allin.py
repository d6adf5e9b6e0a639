import sys, numpy as np
sys.path.insert(0, '.')
from cpu_tsdf_amd import capi, synth
capi.use_test_library()  # knobs / selftest hooks live in libtsdf_hip_test.so (include/tsdf_hip_test.h)
from oracle.oracle import OracleVolume
from tests.common import frames, make_volume
import ctypes as C
def info(vol):
    out = (C.c_int32 * 4)(); capi.check(capi.load().tsdf_hip_last_launch_info(vol._need(), out), "x"); return list(out)
for mode, allin in (("as_test", 1), ("as_test", 0), ("sync_each", 0)):
    capi.set_tuning("allin", allin)
    vol, sc = make_volume(96, color=True, max_weight=100.0)
    vol.setLayout(capi.LAYOUT_AUTO)
    vol.reset()
    ov = OracleVolume(vol._p)
    for i, tr, dep, col in frames(sc, 7, 9, noise=True):
        dep = dep.copy(); dep[(i * 7) % 50::53, ::3] = np.nan
        n = vol.integrateCloud(dep, col, tr, count=(i % 2 == 0))
        if mode == "sync_each": vol.synchronize()
        ov.integrate(dep, col, synth.cam_from_vol_f32(tr))
    d, w, rgb = vol.download()
    bad = (d.view(np.uint32) != ov.d.view(np.uint32)) | (w != ov.w) | (rgb != ov.rgb).any(-1)
    print(mode, "allin", allin, "bad voxels", int(bad.sum()))
    if bad.any():
        idx = np.argwhere(bad)
        print(" z range", idx[:,0].min(), idx[:,0].max(), "y", idx[:,1].min(), idx[:,1].max(), "x", idx[:,2].min(), idx[:,2].max())
        print(" x%4 hist", np.bincount(idx[:,2] % 4, minlength=4), "y%8 hist", np.bincount(idx[:,1] % 8, minlength=8), " (x//4)%8", np.bincount((idx[:,2]//4) % 8, minlength=8))
        print(" w diff hist", np.unique((w - ov.w)[bad], return_counts=True))
        for k in idx[:6]:
            z,y,x = k
            print("  ", k, "gpu d,w,rgb", d[z,y,x], w[z,y,x], rgb[z,y,x], "oracle", ov.d[z,y,x], ov.w[z,y,x], ov.rgb[z,y,x])
    vol.close()

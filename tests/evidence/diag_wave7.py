"""Bisect of the "wrong voxels" build (tsdf_integrate.hip TSDF_WPE_GENERAL note): runs the failing
scenario of diag_allin.py (96^3, colour, PACKED, noisy frames with NaN holes) against the library named by
TSDF_HIP_LIB_PATH, downloads after EVERY frame and compares with the oracle, for the general instance (allin = 0) with
counting always / never / alternating.  Prints one line per (mode, frame): instance info, bad voxels, where."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cpu_tsdf_amd import capi, synth  # noqa: E402
capi.use_test_library()  # knobs / selftest hooks live in libtsdf_hip_test.so (include/tsdf_hip_test.h)
from oracle.oracle import OracleVolume  # noqa: E402
from tests.common import frames, make_volume  # noqa: E402
import ctypes as C  # noqa: E402


def info(vol):
    out = (C.c_int32 * 4)()
    capi.check(capi.load().tsdf_hip_last_launch_info(vol._need(), out), "x")
    return list(out)


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    out = {"lib": os.environ.get("TSDF_HIP_LIB_PATH", "default"), "res": res, "runs": []}
    for allin, counting in ((0, "never"), (0, "always"), (0, "alternate"), (1, "never"), (1, "always"), (1, "alternate")):
        capi.set_tuning("allin", allin)
        vol, sc = make_volume(res, color=True, max_weight=100.0)
        vol.reset()
        ov = OracleVolume(vol._p)
        first_bad, hist = None, []
        for i, tr, dep, col in frames(sc, 5, 9, noise=True):
            dep = dep.copy()
            dep[(i * 7) % 50::53, ::3] = np.nan
            cnt = counting == "always" or (counting == "alternate" and i % 2 == 0)
            vol.integrateCloud(dep, col, tr, count=cnt)
            ov.integrate_culled(dep, col, tr, synth.cam_from_vol_f32(tr))
            d, w, rgb = vol.download()
            bad = (d.view(np.uint32) != ov.d.view(np.uint32)) | (w != ov.w) | (rgb != ov.rgb).any(-1)
            nb = int(bad.sum())
            rec = {"frame": i, "info": info(vol), "bad": nb}
            if nb:
                idx = np.argwhere(bad)
                rec.update({"z": [int(idx[:, 0].min()), int(idx[:, 0].max())], "y": [int(idx[:, 1].min()), int(idx[:, 1].max())],
                            "x": [int(idx[:, 2].min()), int(idx[:, 2].max())],
                            "x_mod4": np.bincount(idx[:, 2] % 4, minlength=4).tolist(),
                            "y_mod8": np.bincount(idx[:, 1] % 8, minlength=8).tolist(),
                            "quad_mod8": np.bincount((idx[:, 2] // 4) % 8, minlength=8).tolist(),
                            "bad_d": int((d.view(np.uint32) != ov.d.view(np.uint32)).sum()), "bad_w": int((w != ov.w).sum()),
                            "bad_rgb": int((rgb != ov.rgb).any(-1).sum()),
                            "w_diff": [[float(a), int(b)] for a, b in zip(*np.unique((w - ov.w)[bad], return_counts=True))][:8],
                            "sample": [[int(v) for v in k] + [float(d[tuple(k)]), float(ov.d[tuple(k)]), float(w[tuple(k)]), float(ov.w[tuple(k)])]
                                       for k in idx[:4]]})
                if first_bad is None:
                    first_bad = i
                # resynchronise the oracle with the device so that later frames are judged on their own
                ov.d[...], ov.w[...] = d, w
                ov.rgb[...] = rgb
            hist.append(rec)
        out["runs"].append({"allin": allin, "counting": counting, "first_bad_frame": first_bad, "frames": hist})
        vol.close()
    capi.set_tuning("allin", 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

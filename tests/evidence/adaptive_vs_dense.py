"""How far is the reference's DEFAULT (adaptive octree, --max-cell-size 0.5 m) from the dense mode this project is
bit-identical to?  (VERDICT r03 "missing" #3 / next #9.)  Both sides are the REFERENCE's own code (oracle/_ref): the same
Scene-B sequence (camera inside a 10 m volume, sensor range 0..3 m: the regime the octree was built for) integrated once
with max_cell = voxel size (every leaf finest: what the MI355X path reproduces bit for bit) and once with the program's
default max_cell 0.5 m (cells split only near observed surfaces, impl/tsdf_volume_octree.hpp:161-187), then meshed by the
reference's MarchingCubesTSDFOctree.  Reports triangle counts, the symmetric Hausdorff distance between the two meshes'
vertex sets and its quantiles, and renderView differences.  CPU only.  usage: adaptive_vs_dense.py [res] [frames] [size_m]   (size_m scales Scene B: volume edge, camera distance and sensor range; the truncation
band stays the programs' 3 cm, so choose res / size_m with a voxel of ~1 cm or the band is thinner than a voxel)"""
import json
import os
import sys
import time

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cpu_tsdf_amd import synth  # noqa: E402
from oracle import refbind  # noqa: E402


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    assert refbind.available()
    size = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
    k = size / 10.0
    sc = synth.Scene(size, 640, 480, sphere=0.1, box=0.35)
    out = {"res": res, "voxel_m": size / res, "frames": nf, "scene": f"B: camera inside a {size} m volume, range 0..{3 * k} m"}

    def pose(i):
        p = synth.scene_b_pose(i, nf).copy()
        p[:3, 3] *= k
        return p
    vols = {}
    for name, dense in (("dense", True), ("adaptive_0.5m", False)):
        t0 = time.time()
        v = refbind.RefVolume(res, size, sc.width, sc.height, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3.0 * k, color=True, dense=dense, max_cell=0.5)
        for i in range(nf):
            tr = pose(i)
            v.integrate(sc.depth(tr, noise_seed=100 + i), sc.bgra(i), tr)
        verts, _, polys, _ = v.march(0.0, 1)
        view, _ = v.render_view(pose(nf // 2), 2)
        vols[name] = (verts, view)
        out[name] = {"triangles": int(len(verts) // 3), "seconds": round(time.time() - t0, 1)}
        v.close()
    a, b = vols["dense"][0], vols["adaptive_0.5m"][0]
    da = cKDTree(b).query(a)[0]
    db = cKDTree(a).query(b)[0]
    q = lambda x: {"median": float(np.median(x)), "p95": float(np.quantile(x, 0.95)), "p99": float(np.quantile(x, 0.99)), "max": float(x.max())}
    out["vertex_distance_dense_to_adaptive_m"] = q(da)
    out["vertex_distance_adaptive_to_dense_m"] = q(db)
    out["hausdorff_m"] = float(max(da.max(), db.max()))
    va, vb = vols["dense"][1], vols["adaptive_0.5m"][1]
    ha, hb = np.isfinite(va[..., 0]), np.isfinite(vb[..., 0])
    both = ha & hb
    dz = np.linalg.norm(va[both][:, :3] - vb[both][:, :3], axis=1)
    out["renderView"] = {"hits_dense": int(ha.sum()), "hits_adaptive": int(hb.sum()), "hits_both": int(both.sum()),
                         "point_distance_m": q(dz) if both.any() else None}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

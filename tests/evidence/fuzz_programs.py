#!/usr/bin/env python3
"""Random command lines for the two programs: cpu_tsdf_amd/bin/integrate (this repo, MI355X) against
oracle/_ref/ref_integrate (the reference's own program, compiled from its unmodified sources) on the same random input
directory -- ingest mode (organised / unorganised, camera / world frame, units, zero -> NaN, text / binary poses, PCD
flavours), colour, sensor range, truncation, weights, frame count, output options -- and mesh.ply compared byte for byte.
usage: python tests/evidence/fuzz_programs.py [--cases 30] [--seed 1]"""
import argparse
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import sequence_util as su  # noqa: E402

W, H, RES, SIZE = 160, 120, 64, 8.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    assert os.path.exists(su.REF_INTEGRATE) and os.path.exists(su.OUR_INTEGRATE), "programs not built"
    rng = np.random.RandomState(a.seed)
    bad = []
    for case in range(a.cases):
        kw, flags = {}, []
        if rng.rand() < 0.4:
            kw["organized"] = True
            flags += ["--organized"]
        elif rng.rand() < 0.4:
            kw["world"] = True
            flags += ["--world"]
        if rng.rand() < 0.4:
            u = float(rng.choice([0.001, 0.01, 0.5]))
            kw["units"] = u
            flags += ["--cloud-units", u]
        if rng.rand() < 0.4:
            kw["binary_poses"] = True
        if rng.rand() < 0.3:
            flags += ["--zero-nans"]
        if rng.rand() < 0.6:
            flags += ["--color"]
        if rng.rand() < 0.4:
            flags += ["--max-sensor-dist", float(rng.uniform(2.5, 6.0)), "--min-sensor-dist", float(rng.choice([0.0, 0.3, 0.5]))]
        # (truncation of a few voxels, or the 8 m scene leaves no surface in a 32^3 / 64^3 grid)
        flags += ["--trunc-dist-pos", float(rng.uniform(0.25, 0.6)), "--trunc-dist-neg", float(rng.uniform(0.25, 0.6))]
        if rng.rand() < 0.3:
            flags += ["--min-weight", float(rng.choice([1, 2]))]
        if rng.rand() < 0.3:
            flags += ["--num-frames", int(rng.randint(2, 5))]
        if rng.rand() < 0.3:
            flags += ["--save-ascii"]
        kinds = [("binary", "ascii", "binary_compressed"), ("ascii",), ("binary_compressed", "binary")][rng.randint(3)]
        res = int(rng.choice([32, 64]))
        common = ["--volume-size", SIZE, "--cell-size", SIZE / res, "--max-cell-size", SIZE / res, "--width", W, "--height", H]
        d = su.digit_free_dir("fuzzprog")
        try:
            su.make_sequence(os.path.join(d, "in"), n_frames=int(rng.randint(2, 6)), width=W, height=H, seed=int(rng.randint(1000)),
                             kinds=kinds, **kw)
            args = ["--in", os.path.join(d, "in")] + common + flags
            rc_ref, log_ref = su.run(su.REF_INTEGRATE, args + ["--out", os.path.join(d, "ref")])
            rc_our, log_our = su.run(su.OUR_INTEGRATE, args + ["--out", os.path.join(d, "our")])
            verdict = "ok"
            if rc_ref != rc_our:
                verdict = f"DIFF exit codes {rc_ref} vs {rc_our}: {log_our[-300:]!r}"
            elif rc_ref == 0:
                ra = open(os.path.join(d, "ref", "mesh.ply"), "rb").read()
                rb = open(os.path.join(d, "our", "mesh.ply"), "rb").read()
                nfaces = len(su.read_ply(os.path.join(d, "ref", "mesh.ply"))[2])
                verdict = f"ok ({nfaces} faces)" if ra == rb else f"DIFF mesh.ply ({len(ra)} vs {len(rb)} bytes)"
            else:
                verdict = f"ok (both exit {rc_ref})"
        finally:
            shutil.rmtree(d, ignore_errors=True)
        print(f"case {case:3d}: res {res} {' '.join(str(f) for f in flags)} kinds {','.join(kinds)} poses "
              f"{'binary' if kw.get('binary_poses') else 'text'}  {verdict}", flush=True)
        if verdict.startswith("DIFF"):
            bad.append(case)
    print(f"{a.cases} cases, seed {a.seed}: {len(bad)} with differences {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

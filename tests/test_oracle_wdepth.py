"""CPU tier: the oracle's weight_by_depth_ restatement (hpp:200-202) pinned to the reference's own outputs.

(1) tests/golden/reference_wdepth_32.npz -- written by tests/golden/make_golden_wdepth.py from oracle/_ref with
    weight_by_depth_ switched on the only way the reference allows (a patched .vol header through load());
(2) live against oracle/_ref when it is present, also without colour.
Bar: bit equality, NaN states included (a voxel first observed beyond 10 m becomes 0/0)."""
import os

import numpy as np
import pytest

from cpu_tsdf_amd import synth
from oracle.oracle import OracleVolume
from tests.common import assert_same_f32
from tests.golden.make_golden_wdepth import H, NF, RES, W, frame, weighted_reference
from tests.test_oracle_golden import params

GOLD = os.path.join(os.path.dirname(__file__), "golden", "reference_wdepth_32.npz")


def test_oracle_weight_by_depth_matches_reference_golden():
    gold = np.load(GOLD)
    sc = synth.scene_a(RES, W, H)
    assert sc.size == float(gold["size"]) and NF == int(gold["n_frames"])
    ov = OracleVolume(params(RES, W, H, sc.size))
    for i in range(NF):
        tr, dep, col = frame(sc, i)
        ov.integrate(dep, col, synth.cam_from_vol_f32(tr), weight_by_depth=True)
        assert_same_f32(ov.d, gold[f"d{i}"], f"d after frame {i}")
        assert_same_f32(ov.w, gold[f"w{i}"], f"w after frame {i}")
        assert np.array_equal(ov.rgb, gold[f"rgb{i}"])
    assert np.isnan(ov.d).any() and ((ov.w % 1) != 0).any() and (ov.w > 0).mean() > 0.3


@pytest.mark.parametrize("color", [True, False])
def test_oracle_weight_by_depth_equals_compiled_reference(color):
    from oracle import refbind
    if not refbind.available():
        pytest.skip("oracle/_ref not built (needs /root/reference: make -C oracle ref)")
    sc = synth.scene_a(RES, W, H)
    rv = weighted_reference(sc, color=color)
    ov = OracleVolume(params(RES, W, H, sc.size, color))
    for i in range(3):
        tr, dep, col = frame(sc, i)
        dep[dep > 5] = 3.0 + i  # mid-range weights (0.7 .. 0.5) instead of the golden's zeros
        rv.integrate(dep, col, tr)
        ov.integrate(dep, col if color else None, synth.cam_from_vol_f32(tr), weight_by_depth=True)
    d, w, rgb, _, _ = rv.dump_dense()
    assert_same_f32(ov.d, d, "d")
    assert_same_f32(ov.w, w, "w")
    if color:
        assert np.array_equal(ov.rgb, rgb)
    rv.close()

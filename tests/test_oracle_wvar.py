"""CPU tier: the oracle's weight_by_variance_ restatement (hpp:203-204, octree.cpp:152-163,281-287) pinned to the
reference's own outputs.

(1) tests/golden/reference_wvar_32.npz -- written by tests/golden/make_golden_wvar.py from oracle/_ref with
    weight_by_variance_ switched on the only way the reference allows (a patched .vol header through load());
(2) live against oracle/_ref when it is present: without colour, and together with weight_by_depth_.
Bar: bit equality (std::exp(float) is the host's expf on both sides; the GPU tests pin the device's)."""
import numpy as np
import os
import pytest

from cpu_tsdf_amd import synth
from oracle.oracle import OracleVolume
from tests.common import assert_same_f32
from tests.golden.make_golden_wvar import H, NF, RES, W, frame, variance_reference
from tests.test_oracle_golden import params

GOLD = os.path.join(os.path.dirname(__file__), "golden", "reference_wvar_32.npz")


def test_oracle_weight_by_variance_matches_reference_golden():
    gold = np.load(GOLD)
    sc = synth.scene_a(RES, W, H)
    assert sc.size == float(gold["size"]) and NF == int(gold["n_frames"])
    ov = OracleVolume(params(RES, W, H, sc.size))
    for i in range(NF):
        tr, dep, col = frame(sc, i)
        ov.integrate_variance(dep, col, synth.cam_from_vol_f32(tr))
        if i + 1 >= 6:
            assert_same_f32(ov.d, gold[f"d{i}"], f"d after frame {i}")
            assert_same_f32(ov.w, gold[f"w{i}"], f"w after frame {i}")
            assert np.array_equal(ov.rgb, gold[f"rgb{i}"])
    assert ((ov.w % 1) != 0).mean() > 0.05 and ov.nsample.max() == NF and (ov.M != 0).mean() > 0.3


@pytest.mark.parametrize("color,by_depth", [(False, False), (True, True)])
def test_oracle_weight_by_variance_equals_compiled_reference(color, by_depth):
    from oracle import refbind
    if not refbind.available():
        pytest.skip("oracle/_ref not built (needs /root/reference: make -C oracle ref)")
    sc = synth.scene_a(RES, W, H)
    rv = variance_reference(sc, color=color, by_depth=by_depth)
    ov = OracleVolume(params(RES, W, H, sc.size, color))
    for i in range(9):
        tr, dep, col = frame(sc, i + 3)
        rv.integrate(dep, col, tr)
        ov.integrate_variance(dep, col if color else None, synth.cam_from_vol_f32(tr), weight_by_depth=by_depth)
    d, w, rgb, _, _ = rv.dump_dense()
    assert_same_f32(ov.d, d, "d")
    assert_same_f32(ov.w, w, "w")
    if color:
        assert np.array_equal(ov.rgb, rgb)
    assert ((w % 1) != 0).mean() > 0.05
    rv.close()

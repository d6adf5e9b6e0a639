"""The three committed copies of the marching-cubes case tables -- product (cpu_tsdf_amd/csrc/mc_tables.h), oracle
(oracle/mc_tables.h) and the PCL stand-in the compiled reference links (compat/mini_pcl_mc_tables.inc) -- are re-derived here
instead of trusted (VERDICT r05 "what's weak" #1 / next #8): parsed from the committed text, compared with each other, checked
structurally against the cube geometry pcl::MarchingCubes::createSurface uses, and, where scikit-image's
LUT file exists on the machine (it does in this image, under another interpreter's site-packages), compared entry by entry
with its CASESCLASSIC = Lorensen / Bourke's table -- the independent public copy tools/gen_mc_tables.py generated them from."""
import base64
import glob
import importlib.util
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EDGE_CORNERS = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
LUT_GLOBS = ["/opt/conda/lib/python3*/site-packages/skimage/measure/_marching_cubes_lewiner_luts.py",
             "/usr/lib/python3*/site-packages/skimage/measure/_marching_cubes_lewiner_luts.py",
             "/usr/lib/python3/dist-packages/skimage/measure/_marching_cubes_lewiner_luts.py",
             "/usr/local/lib/python3*/*-packages/skimage/measure/_marching_cubes_lewiner_luts.py"]


def _array(text, name):
    """The integers of `name[...] = { ... };` in a C source text."""
    m = re.search(re.escape(name) + r"\s*(?:\[\d+\])+\s*=\s*\{(.*?)\};", text, re.S)
    assert m, name
    return [int(t, 0) for t in re.findall(r"-?(?:0x[0-9a-fA-F]+|\d+)", m.group(1))]


def _tables(rel, edge_name, tri_name, ntri_name=None):
    text = open(os.path.join(ROOT, rel)).read()
    et = np.array(_array(text, edge_name), np.int64)
    tri = np.array(_array(text, tri_name), np.int64).reshape(256, 16)
    nt = np.array(_array(text, ntri_name), np.int64) if ntri_name else None
    assert et.shape == (256,)
    return et, tri, nt


COPIES = [("cpu_tsdf_amd/csrc/mc_tables.h", "mc_edge_table", "mc_tri_table", "mc_ntri_table"),
          ("oracle/mc_tables.h", "mc_edge_table", "mc_tri_table", "mc_ntri_table"),
          ("compat/mini_pcl_mc_tables.inc", "edgeTable", "triTable", None)]


def test_the_three_copies_hold_the_same_tables():
    first = _tables(*COPIES[0])
    for c in COPIES[1:]:
        et, tri, nt = _tables(*c)
        assert np.array_equal(et, first[0]), c[0]
        assert np.array_equal(tri, first[1]), c[0]
        if nt is not None:
            assert np.array_equal(nt, first[2]), c[0]


@pytest.mark.parametrize("copy", COPIES, ids=[c[0] for c in COPIES])
def test_tables_follow_from_the_cube_geometry(copy):
    et, tri, nt = _tables(*copy)
    for c in range(256):
        # edgeTable[c] = the cube edges whose two corners lie on different sides in case c (createSurface's corner pairs)
        want = 0
        for e, (a, b) in enumerate(EDGE_CORNERS):
            if ((c >> a) & 1) != ((c >> b) & 1):
                want |= 1 << e
        assert et[c] == want, c
        row = tri[c]
        n = int(np.argmax(row == -1)) if (row == -1).any() else 16
        assert n % 3 == 0 and n <= 15 and (row[n:] == -1).all(), c
        used = 0
        for e in row[:n]:
            assert 0 <= e < 12
            used |= 1 << int(e)
        assert used == et[c], c  # a case's triangles use exactly its crossed edges
        for t in range(n // 3):
            assert len(set(row[3 * t:3 * t + 3].tolist())) == 3, c  # no degenerate triangle
        if nt is not None:
            assert nt[c] == n // 3, c
    assert et[0] == 0 and et[255] == 0 and all(et[c] == et[255 - c] for c in range(256))
    # rows and values of Bourke's published tables
    recalled = {1: [0, 8, 3], 2: [0, 1, 9], 3: [1, 8, 3, 9, 8, 1], 4: [1, 2, 10], 5: [0, 8, 3, 1, 2, 10], 8: [3, 11, 2],
                15: [9, 8, 10, 10, 8, 11], 16: [4, 7, 8], 128: [7, 6, 11], 254: [0, 3, 8], 253: [0, 9, 1]}
    for c, want in recalled.items():
        assert [int(v) for v in tri[c] if v != -1] == want, c
    assert et[1] == 0x109 and et[2] == 0x203 and et[3] == 0x30a and et[128] == 0x8c0 and et[254] == 0x109


def _skimage_lut():
    for g in LUT_GLOBS:
        for p in sorted(glob.glob(g)):
            return p
    return None


def test_tables_equal_scikit_images_classic_cases():
    lut = _skimage_lut()
    if lut is None:
        pytest.skip("no scikit-image LUT file on this machine (the tables' independent public copy)")
    sp = importlib.util.spec_from_file_location("_mc_luts", lut)
    m = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(m)
    shape, txt = m.CASESCLASSIC
    want = np.frombuffer(base64.decodebytes(txt.encode()), dtype=np.int8).reshape(shape).astype(np.int64)
    assert want.shape == (256, 16)
    for c in COPIES:
        _, tri, _ = _tables(*c)
        assert np.array_equal(tri, want), c[0]

"""CPU tier: the streaming .vol writer / reader (cpu_tsdf_amd/csrc/host/vol_format.h) behind
TSDFVolumeOctree::save / load, driven from host arrays through tests/harness/vol_stream.cpp.

The block size is an implementation detail: every block size must produce byte for byte the same file and
read back the same voxels, and the files must cross with the reference's own save / load
(src/lib/tsdf_volume_octree.cpp:222-275, src/lib/octree.cpp:289-304,360-367,645-656)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import refbind

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = 32
SIZE = 0.4


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("volstream") / "vol_stream")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-pthread", "-I", os.path.join(ROOT, "cpu_tsdf_amd", "csrc"),
                           os.path.join(ROOT, "tests", "harness", "vol_stream.cpp"), "-o", exe])
    return exe


def grid(color, seed=0):
    """A volume with structure at every octree level: unseen octants, free space, a noisy band."""
    rng = np.random.RandomState(seed)
    d = np.full((RES,) * 3, -1.0, np.float32)
    w = np.zeros((RES,) * 3, np.float32)
    rgb = np.zeros((RES,) * 3 + (3,), np.uint8)
    d[:, :, 16:] = 1.0      # free space seen 3 times: collapses to 16^3 leaves
    w[:, :, 16:] = 3.0
    d[8:16, 8:24, 4:20] = rng.uniform(-1, 1, (8, 16, 16)).astype(np.float32)  # a surface band: single voxels
    w[8:16, 8:24, 4:20] = rng.randint(1, 5, (8, 16, 16)).astype(np.float32)
    d[24:28, 0:4, 0:4] = 0.25   # one uniform 4^3 node inside an otherwise unseen octant
    w[24:28, 0:4, 0:4] = 1.0
    if color:
        rgb[w > 0] = (10, 20, 30)
        rgb[8:16, 8:24, 4:20] = rng.randint(0, 256, (8, 16, 16, 3))
    return d, w, rgb


def write_raw(path, d, w, rgb, color):
    with open(path, "wb") as f:
        f.write(d.tobytes())
        f.write(w.tobytes())
        if color:
            f.write(rgb.tobytes())


def read_raw(path, n, color):
    raw = np.fromfile(path, np.uint8)
    nv = n ** 3
    d = raw[:4 * nv].view(np.float32).reshape(n, n, n)
    w = raw[4 * nv:8 * nv].view(np.float32).reshape(n, n, n)
    rgb = raw[8 * nv:].reshape(n, n, n, 3) if color else None
    return d, w, rgb


@pytest.mark.parametrize("color", [False, True])
def test_every_block_size_writes_the_same_file_and_reads_the_same_voxels(harness, tmp_path, color):
    d, w, rgb = grid(color)
    raw = str(tmp_path / "in.raw")
    write_raw(raw, d, w, rgb, color)
    files = {}
    for chunk in (32, 16, 8, 4, 1):
        out = str(tmp_path / f"c{chunk}.vol")
        fetched = int(subprocess.check_output([harness, "write", raw, str(RES), str(SIZE), str(int(color)), str(chunk), out]))
        files[chunk] = open(out, "rb").read()
        m = (RES // chunk) ** 3
        assert m <= fetched <= 2 * m       # pass 1 sees every block once, pass 2 only the non-uniform ones
        if chunk == 16:
            assert fetched == 8 + 5   # the band crosses four octants, the 4^3 node sits in a fifth
    assert all(files[c] == files[32] for c in files), "block size changed the file"
    node = 43 if color else 40
    full = sum(8 ** l for l in range(6)) * node
    assert len(files[32]) < 0.2 * full     # collapsed: far smaller than a fully refined tree
    for chunk in (32, 8, 2):
        back = str(tmp_path / f"back{chunk}.raw")
        n, c, blocks = map(int, subprocess.check_output([harness, "read", str(tmp_path / "c32.vol"), str(chunk), back]).split())
        assert (n, c) == (RES, int(color)) and blocks == (RES // chunk) ** 3   # each block stored exactly once
        d2, w2, rgb2 = read_raw(back, RES, color)
        assert np.array_equal(d2.view(np.uint32), d.view(np.uint32)) and np.array_equal(w2, w)
        if color:
            assert np.array_equal(rgb2, rgb)


def test_parallel_block_writer_writes_what_the_serial_walk_writes(harness, tmp_path):
    """Blocks of edge >= 32 are serialised by 64 worker subtrees (vol_format.h write_block); smaller ones by the plain
    recursion.  A 128^3 grid with a uniform octant (a level-1 leaf inside every 64^3 / 128^3 block that holds it), uniform
    level-2 cubes, single uniform voxels groups and noise: the files of block edges 128, 64, 32 (parallel, at three
    depths) and 16 (serial) are the same bytes."""
    n = 128
    rng = np.random.RandomState(9)
    d = rng.uniform(-1, 1, (n,) * 3).astype(np.float32)
    w = rng.randint(1, 7, (n,) * 3).astype(np.float32)
    rgb = rng.randint(0, 256, (n,) * 3 + (3,)).astype(np.uint8)
    for box, val in (((slice(0, 64),) * 3, (-1.0, 0.0)), ((slice(64, 96), slice(0, 32), slice(32, 64)), (1.0, 3.0)),
                     ((slice(96, 112), slice(112, 128), slice(0, 16)), (0.5, 2.0)), ((slice(64, 128), slice(64, 128), slice(64, 128)), (1.0, 5.0))):
        d[box], w[box], rgb[box] = val[0], val[1], (val[1] * 10, 20, 30)
    d[70, 70, 70] = 0.125  # one voxel that breaks the last uniform octant
    raw = str(tmp_path / "in.raw")
    write_raw(raw, d, w, rgb, True)
    files = {}
    for chunk in (128, 64, 32, 16):
        out = str(tmp_path / f"c{chunk}.vol")
        subprocess.check_call([harness, "write", raw, str(n), str(SIZE), "1", str(chunk), out], stdout=subprocess.DEVNULL)
        files[chunk] = open(out, "rb").read()
    assert all(files[c] == files[16] for c in files), "the parallel block writer changed the file"
    back = str(tmp_path / "back.raw")
    subprocess.check_call([harness, "read", str(tmp_path / "c128.vol"), "64", back], stdout=subprocess.DEVNULL)
    d2, w2, rgb2 = read_raw(back, n, True)
    assert np.array_equal(d2.view(np.uint32), d.view(np.uint32)) and np.array_equal(w2, w) and np.array_equal(rgb2, rgb)


@pytest.mark.parametrize("color", [False, True])
def test_streamed_files_cross_with_the_reference(harness, tmp_path, color):
    if not refbind.available():
        pytest.skip("oracle/_ref not built")
    d, w, rgb = grid(color, seed=3)
    raw = str(tmp_path / "in.raw")
    write_raw(raw, d, w, rgb, color)
    ours = str(tmp_path / "ours.vol")
    subprocess.check_call([harness, "write", raw, str(RES), str(SIZE), str(int(color)), "8", ours], stdout=subprocess.DEVNULL)
    ref = refbind.RefVolume(RES, SIZE, 640, 480, 525.0, 525.0, 319.5, 239.5, 0.0, 3.0, color=color)
    ref.load(ours)
    rd, rw, rrgb, leaf, _ = ref.dump_dense()
    assert np.array_equal(rd.view(np.uint32), d.view(np.uint32)) and np.array_equal(rw, w)
    if color:
        assert np.array_equal(rrgb, rgb)
    assert leaf.max() >= 16 * leaf.min()            # coarse leaves survived the trip
    theirs = str(tmp_path / "theirs.vol")
    ref.save(theirs)                                # the reference's own writer, same tree
    for chunk in (32, 4):
        back = str(tmp_path / f"back{chunk}.raw")
        subprocess.check_call([harness, "read", theirs, str(chunk), back], stdout=subprocess.DEVNULL)
        d2, w2, rgb2 = read_raw(back, RES, color)
        assert np.array_equal(d2.view(np.uint32), d.view(np.uint32)) and np.array_equal(w2, w)
        if color:
            assert np.array_equal(rgb2, rgb)
    ref.close()


def test_reader_rejects_damaged_files(harness, tmp_path):
    d, w, rgb = grid(False)
    raw = str(tmp_path / "in.raw")
    write_raw(raw, d, w, rgb, False)
    good = str(tmp_path / "good.vol")
    subprocess.check_call([harness, "write", raw, str(RES), str(SIZE), "0", "8", good], stdout=subprocess.DEVNULL)
    blob = open(good, "rb").read()
    cut = str(tmp_path / "cut.vol")
    open(cut, "wb").write(blob[:len(blob) - 100])
    p = subprocess.run([harness, "read", cut, "8", str(tmp_path / "x.raw")], capture_output=True)
    assert p.returncode == 1 and b"truncated" in p.stderr


def test_reader_survives_corrupted_files(tmp_path):
    """load() parses files from anywhere: flipped bytes, truncations, corrupted header text and child counts must
    end in an error (or a successful read of a still-consistent tree), never in a memory error -- the harness is
    built with AddressSanitizer + UBSan for this test."""
    exe = str(tmp_path / "vol_stream_asan")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-pthread", "-I", os.path.join(ROOT, "cpu_tsdf_amd", "csrc"),
                           os.path.join(ROOT, "tests", "harness", "vol_stream.cpp"), "-o", exe])
    d, w, rgb = grid(True)
    raw, vol = str(tmp_path / "in.raw"), str(tmp_path / "good.vol")
    write_raw(raw, d, w, rgb, True)
    subprocess.check_call([exe, "write", raw, str(RES), str(SIZE), "1", "8", vol], stdout=subprocess.DEVNULL)
    blob = open(vol, "rb").read()
    tree = blob.index(b"#OCTREEBINARY") + 14
    rng = np.random.RandomState(0)
    outcomes = set()
    for it in range(90):
        b = bytearray(blob)
        mode = it % 4
        if mode == 0:      # bytes of the binary tree
            for _ in range(rng.randint(1, 6)):
                b[rng.randint(tree, len(b))] = rng.randint(0, 256)
        elif mode == 1:    # truncation
            b = b[:rng.randint(10, len(b))]
        elif mode == 2:    # the ASCII header
            b[rng.randint(0, tree)] = rng.randint(32, 127)
        else:              # resolution words / the root's child count
            b[tree + rng.randint(0, 64)] = rng.randint(0, 256)
        bad = str(tmp_path / "bad.vol")
        open(bad, "wb").write(b)
        p = subprocess.run([exe, "read", bad, str([8, 32, 4][it % 3]), str(tmp_path / "out.raw")], capture_output=True, timeout=120,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert p.returncode in (0, 1, 4, 5), (it, mode, p.returncode, p.stderr[-600:])
        outcomes.add(p.returncode)
    assert 1 in outcomes      # errors were really provoked

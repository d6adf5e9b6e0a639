#!/usr/bin/env python3
"""A/B builds of libtsdf_hip.so with extra compiler flags: tools/build_variant.py NAME [-DFOO=1 ...] writes
cpu_tsdf_amd/lib/variants/NAME/libtsdf_hip.so; run anything with TSDF_HIP_LIB_PATH pointing at it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cpu_tsdf_amd import build as b  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    out = os.path.join(b.LIBDIR, "variants", name)
    os.makedirs(os.path.join(out, "obj"), exist_ok=True)
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + b.CSRC]
    objs, procs = [], []
    for src in b.sources():
        obj = os.path.join(out, "obj", os.path.basename(src) + ".o")
        objs.append(obj)
        procs.append(subprocess.Popen([b._hipcc()] + b.HIPCC_FLAGS + ["-DTSDF_HIP_TEST_HOOKS"] + flags + inc + ["-c", src, "-o", obj]))
    if any(p.wait() for p in procs):
        raise SystemExit("hipcc failed")
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-Wl,--version-script=" + b.EXPORTS] + objs + ["-o", os.path.join(out, "libtsdf_hip.so")])
    print(os.path.join(out, "libtsdf_hip.so"))


if __name__ == "__main__":
    main()

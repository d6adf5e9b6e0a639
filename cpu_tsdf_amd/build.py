"""Build recipe for libtsdf_hip.so (hand-written HIP, gfx950 only, built in-tree with hipcc)."""
import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "lib", "libtsdf_hip.so")

# -ffp-contract=off: the reference CPU build has no FMA (no -march in its CMakeLists.txt), and
# per-voxel parity needs the same separate mul/add roundings on the GPU.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wno-unused-result"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "tsdf_hip.h")]
    return os.path.getmtime(LIB) < max(os.path.getmtime(p) for p in deps)


def build_hip(force=False, verbose=False):
    """Compile every HIP translation unit into cpu_tsdf_amd/lib/libtsdf_hip.so."""
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the HIP extension (no fallback path exists)")
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc] + HIPCC_FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + sources() + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB

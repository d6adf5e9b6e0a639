"""Build recipes (in-tree, hipcc / g++ only; no cmake needed):

  libtsdf_hip.so       hand-written HIP kernels + the C ABI of include/tsdf_hip.h (gfx950 only)
  libcpu_tsdf_hip.so   C++ host shell: cpu_tsdf::TSDFVolumeOctree / MarchingCubesTSDFOctree with the
                       reference's signatures (include/cpu_tsdf/*.h) on top of the C ABI.  Built against
                       real PCL/Eigen when PCL_INCLUDE_DIRS is set, else against the stand-ins in compat/.
"""
import concurrent.futures
import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
HOST = os.path.join(CSRC, "host")
LIBDIR = os.path.join(_HERE, "lib")
OBJDIR = os.path.join(_HERE, "lib", "obj")
LIB = os.path.join(LIBDIR, "libtsdf_hip.so")
# the test build: the same translation units + the hooks of include/tsdf_hip_test.h (-DTSDF_HIP_TEST_HOOKS)
TEST_LIB = os.path.join(LIBDIR, "libtsdf_hip_test.so")
TEST_OBJDIR = os.path.join(_HERE, "lib", "obj_test")
SHELL_LIB = os.path.join(LIBDIR, "libcpu_tsdf_hip.so")
PROG = os.path.join(CSRC, "prog")
BINDIR = os.path.join(_HERE, "bin")
EXPORTS = os.path.join(CSRC, "exports.map")

# -ffp-contract=off: the reference CPU build has no FMA (no -march in its CMakeLists.txt), and
# per-voxel parity needs the same separate mul/add roundings on the GPU.
# -Werror=undef: a compile-time switch tested by #if before its default is defined silently reads as 0 in the default build
# and as its value in a -D build (round 5: the colour update compiled one way, its table the other; every -D variant passed)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-result", "-Werror=undef"]
HOST_FLAGS = ["-std=c++14", "-O2", "-fPIC", "-fopenmp", "-ffp-contract=off", "-Wall", "-Wno-unknown-pragmas"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "tsdf_hip.h"), os.path.join(ROOT, "include", "tsdf_hip_test.h")]


def _stale(target, deps):
    return not os.path.exists(target) or os.path.getmtime(target) < max(os.path.getmtime(p) for p in deps)


def needs_build(test_hooks=False):
    return _stale(TEST_LIB if test_hooks else LIB, sources() + _headers() + [EXPORTS])


def _hipcc():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the HIP extension (no fallback path exists)")
    return hipcc


def build_hip(force=False, verbose=False, test_hooks=False):
    """Compile every HIP translation unit (in parallel) and link cpu_tsdf_amd/lib/libtsdf_hip.so -- the product: exactly
    the entry points of include/tsdf_hip.h -- or, with test_hooks, libtsdf_hip_test.so: the same sources with
    -DTSDF_HIP_TEST_HOOKS, which adds the hooks of include/tsdf_hip_test.h."""
    lib, objdir = (TEST_LIB, TEST_OBJDIR) if test_hooks else (LIB, OBJDIR)
    if not force and not needs_build(test_hooks):
        return lib
    hipcc = _hipcc()
    os.makedirs(objdir, exist_ok=True)
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    extra = ["-DTSDF_HIP_TEST_HOOKS"] if test_hooks else []
    jobs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or _stale(obj, [src] + _headers()):
            jobs.append([hipcc] + HIPCC_FLAGS + extra + inc + ["-c", src, "-o", obj])
    if verbose:
        for j in jobs:
            print(" ".join(j))
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, len(jobs))) as ex:
        for rc in ex.map(lambda c: subprocess.run(c).returncode, jobs):
            if rc:
                raise RuntimeError("hipcc failed")
    objs = [os.path.join(objdir, os.path.basename(s) + ".o") for s in sources()]
    # -Bsymbolic: calls between the library's own entry points stay inside it, whatever else the process has loaded (the
    # product and the test build can sit in one process: the C++ drop-in links the former, the Python tests load the latter)
    # --version-script: the dynamic symbol table holds the tsdf_hip_* entry points and nothing else (csrc/exports.map)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-Wl,--version-script=" + EXPORTS] + objs + ["-o", lib]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return lib


def clean():
    """Remove every built artefact of this package (objects, libraries, programs): the next build starts from the sources."""
    for d in (OBJDIR, TEST_OBJDIR, os.path.join(LIBDIR, "variants"), BINDIR):
        shutil.rmtree(d, ignore_errors=True)
    for f in (LIB, TEST_LIB, SHELL_LIB):
        if os.path.exists(f):
            os.remove(f)


def host_include_flags():
    """Real PCL/Eigen if the environment names them (PCL_INCLUDE_DIRS, colon separated), else compat/."""
    real = os.environ.get("PCL_INCLUDE_DIRS")
    inc = ["-I" + os.path.join(ROOT, "include")]
    if real:
        inc += ["-I" + p for p in real.split(":") if p]
    else:
        inc += ["-I" + os.path.join(ROOT, "compat")]
    return inc


def build_shell(force=False, verbose=False):
    """libcpu_tsdf_hip.so: the C++ classes with the reference's names and signatures."""
    srcs = sorted(glob.glob(os.path.join(HOST, "*.cpp")))
    deps = srcs + glob.glob(os.path.join(HOST, "*.h")) + glob.glob(os.path.join(ROOT, "include", "cpu_tsdf", "*.h")) + \
        glob.glob(os.path.join(ROOT, "include", "cpu_tsdf", "impl", "*.hpp")) + glob.glob(os.path.join(ROOT, "compat", "*.h")) + \
        [os.path.join(ROOT, "include", "tsdf_hip.h")]
    build_hip(force=False, verbose=verbose)
    if not force and not _stale(SHELL_LIB, deps + [LIB]):
        return SHELL_LIB
    cmd = ["g++"] + HOST_FLAGS + host_include_flags() + ["-I" + HOST, "-shared"] + srcs + \
        ["-L" + LIBDIR, "-ltsdf_hip", "-Wl,-rpath,$ORIGIN", "-o", SHELL_LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SHELL_LIB


def build_programs(force=False, verbose=False):
    """cpu_tsdf_amd/bin/{integrate,tsdf2mesh}: the reference's two programs (src/prog/) on the MI355X path; dropin_rate: the
    timing program of the C++ drop-in's integrateCloud (bench.py host_path.cpp_dropin, tools/cpp_path_timing.py)."""
    build_shell(force=False, verbose=verbose)
    os.makedirs(BINDIR, exist_ok=True)
    outs = []
    for name in ("integrate", "tsdf2mesh", "dropin_rate"):
        src, exe = os.path.join(PROG, name + ".cpp"), os.path.join(BINDIR, name)
        deps = [src, SHELL_LIB] + glob.glob(os.path.join(PROG, "*.h")) + glob.glob(os.path.join(ROOT, "compat", "*.h")) + \
            glob.glob(os.path.join(ROOT, "include", "cpu_tsdf", "*.h"))
        if force or _stale(exe, deps):
            cmd = ["g++", "-std=c++17", "-O2", "-fopenmp", "-ffp-contract=off", "-Wall", "-Wno-unknown-pragmas"] + \
                host_include_flags() + ["-I" + PROG, src, "-L" + LIBDIR, "-lcpu_tsdf_hip", "-ltsdf_hip",
                                        "-Wl,-rpath,$ORIGIN/../lib", "-o", exe]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        outs.append(exe)
    return outs

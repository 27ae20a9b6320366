"""Build libmv3d_hip.so (all HIP kernels + the C-ABI) for gfx950, in-tree.

    python -m mv3d_tf_amd.build [--force]

hipcc cross-compiles without a GPU.  Flags that matter for parity with the reference's
arithmetic: -ffp-contract=off (no implicit FMA), no fast-math; HIP's default IEEE f32
divide/sqrt and preserved denormals are left untouched.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmv3d_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-pthread", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # MV3D_HIPCC_FLAGS: extra compiler flags for experiment builds (e.g. -DMV3D_TUNING, which compiles the environment
    # overrides the tools/ probes use); the shipped library is built without any
    cmd = [hipcc] + FLAGS + os.environ.get("MV3D_HIPCC_FLAGS", "").split() + sources() + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

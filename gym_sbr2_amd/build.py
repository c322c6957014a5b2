"""Builds libsbr_amd.so (HIP, gfx950 only) in-tree: gym_sbr2_amd/lib/libsbr_amd.so."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "sbr_amd.hip")
DEPS = [SRC, os.path.join(_HERE, "csrc", "sbr_device.h"),
        os.path.join(os.path.dirname(_HERE), "include", "sbr_amd.h")]
LIB = os.path.join(_HERE, "lib", "libsbr_amd.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libsbr_amd.so cannot be built (and there is no CPU fallback)")
    return exe


def is_stale():
    return (not os.path.exists(LIB)) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS)


def build_library(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc()] + FLAGS + ["-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))

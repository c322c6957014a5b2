"""Builds libsbr_amd.so (HIP, gfx950 only) in-tree: gym_sbr2_amd/lib/libsbr_amd.so.

Staleness is decided by CONTENT, not by mtime: the hash of the sources and flags the library was built from is kept next to
it (libsbr_amd.so.srchash).  A copied tree (e.g. the snapshot sent to a GPU box) has arbitrary mtimes, and eight ranks that
all thought the library stale would otherwise run hipcc into the same file at once.  The build itself is serialised with a
file lock and published with an atomic rename, so a concurrent loader sees either the old library or the new one."""
import fcntl
import hashlib
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "sbr_amd.hip")
DEPS = [SRC, os.path.join(_HERE, "csrc", "sbr_device.h"),
        os.path.join(os.path.dirname(_HERE), "include", "sbr_amd.h")]
LIB = os.path.join(_HERE, "lib", "libsbr_amd.so")
HASH = LIB + ".srchash"
# -amdgpu-kernarg-preload-count: the first 16 dwords of a kernel's argument segment arrive in SGPRs at wave launch (gfx940+);
# k_step's leading arguments are the pointers its first loads need (-0.5 us per launch at small batches, profiles/r02_notes.md)
# -ffp-contract=off: a multiply-add is fused where the source says __builtin_fma and nowhere else.  hipcc's default lets the
# backend fuse a * b + c when it sees fit, which depends on the shape of the surrounding code: the two register budgets of
# k_step (round 5) differed in three such places - by one ulp in the NO3-PID's integral from the first aerobic call on - where
# the library promises the same bits for an env whatever batch it is stepped in.  No measurable cost (profiles/r05_notes.md).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math", "-ffp-contract=off",
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libsbr_amd.so cannot be built (and there is no CPU fallback)")
    return exe


def source_hash():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in DEPS:
        with open(d, "rb") as f:
            h.update(b"\0" + os.path.basename(d).encode() + b"\0" + f.read())
    return h.hexdigest()


def is_stale():
    if not os.path.exists(LIB) or not os.path.exists(HASH):
        return True
    with open(HASH) as f:
        return f.read().strip() != source_hash()


def build_library(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():          # another process built it while this one waited
                return LIB
            tmp = "%s.tmp.%d" % (LIB, os.getpid())
            cmd = [hipcc()] + FLAGS + ["-o", tmp, SRC]
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, LIB)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
            with open(HASH + ".tmp", "w") as f:
                f.write(source_hash() + "\n")
            os.replace(HASH + ".tmp", HASH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))

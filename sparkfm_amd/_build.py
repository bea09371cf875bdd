"""Builds libfmhip.so (the C-ABI library) in-tree with hipcc for gfx950.

Used by __graft_entry__.build(); importable without torch or a GPU (hipcc cross-compiles).
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libfmhip.so")
SYNTH = os.path.join(LIBDIR, "libfmsynth.so")

HIP_SOURCES = ["fm_forward.hip", "fm_backward.hip", "fm_apply.hip", "als_kernels.hip", "csc_build.hip", "fmhip_api.hip",
               "fmhip_dataset.hip", "fmhip_step.hip", "fmhip_comm.hip", "fmhip_host.cpp"]
HIP_DEPS = ["fm_kernels.h", "fm_constants.h", "fm_device.h", "als_kernels.h", "csc_build.h", "fmhip_internal.h", "fmhip_host.h",
            os.path.join("..", "..", "include", "fmhip.h"), os.path.join("..", "..", "include", "fmhip_experimental.h")]
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in HIP_SOURCES + HIP_DEPS]
    objs, jobs = [], []
    for src in HIP_SOURCES:     # the translation units compile side by side (the two big kernel files take ~25 s each)
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, deps):
            cmd = [_hipcc()] + HIPCC_FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, job in jobs:
        if job.wait() != 0:
            raise subprocess.CalledProcessError(job.returncode, cmd)
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_synth(force=False, verbose=False):
    """Host-only synthetic data generator used by bench.py and the tests (plain C, gcc)."""
    os.makedirs(LIBDIR, exist_ok=True)
    src = os.path.join(CSRC, "synth.c")
    if force or _stale(SYNTH, [src]):
        cmd = ["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-fPIC", "-shared", "-o", SYNTH, src, "-lm"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return SYNTH


def build_all(force=False, verbose=False):
    return build_lib(force, verbose), build_synth(force, verbose)


if __name__ == "__main__":
    import sys
    print(build_all(force="--force" in sys.argv, verbose=True))

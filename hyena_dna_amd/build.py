"""Build libhyena_fftconv.so for gfx950 with hipcc, in-tree (the .so travels with the repo snapshot).

    python -m hyena_dna_amd.build [--force]
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libhyena_fftconv.so")
SOURCES = [os.path.join(CSRC, "fftconv.hip"), os.path.join(CSRC, "onchip.hip"), os.path.join(CSRC, "onchip_dk.hip"), os.path.join(CSRC, "cm.hip"), os.path.join(CSRC, "proj.hip"), os.path.join(CSRC, "filter16.hip")]
HEADERS = [os.path.join(CSRC, "fftconv_kernels.h"), os.path.join(CSRC, "onchip_kernels.h"), os.path.join(CSRC, "onchip_host.h"), os.path.join(CSRC, "launch.h"), os.path.join(CSRC, "cm_kernels.h"), os.path.join(CSRC, "mixer_kernels.h"), os.path.join(CSRC, "filter_kernels.h"), os.path.join(CSRC, "filter16_kernels.h"), os.path.join(CSRC, "block_kernels.h"), os.path.join(CSRC, "proj_kernels.h"), os.path.join(CSRC, "proj2_kernels.h"),
           os.path.join(HERE, "..", "include", "hyena_fftconv.h"), os.path.join(HERE, "..", "include", "hyena_mixer.h"),
           os.path.join(HERE, "..", "include", "hyena_filter.h"), os.path.join(HERE, "..", "include", "hyena_block.h"), os.path.join(HERE, "..", "include", "hyena_proj.h")]
# -fno-slp-vectorize: hipcc otherwise packs the butterflies into v_pk_*_f32 (no faster on CDNA4) at the
# price of ~1000 v_mov per kernel and VGPR spills.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"]


def find_hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(f) <= t for f in SOURCES + HEADERS)


def build(force=False, verbose=True):
    """Safe under concurrent callers (several test processes in a fresh checkout): one builds, under a file lock, and the library is
    linked under a temporary name and renamed into place -- nobody ever loads a half-written one."""
    if not force and up_to_date():
        return LIB
    import fcntl
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and up_to_date():                # somebody else built it while this process waited
            return LIB
        hipcc = find_hipcc()
        objdir = os.path.join(CSRC, "_obj")
        os.makedirs(objdir, exist_ok=True)
        jobs = []
        for src in SOURCES:          # one hipcc process per translation unit, side by side
            obj = os.path.join(objdir, os.path.basename(src) + ".o")
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print("[hyena_dna_amd.build]", " ".join(cmd), flush=True)
            jobs.append((subprocess.Popen(cmd), cmd, obj))
        objs = []
        for proc, cmd, obj in jobs:
            if proc.wait() != 0:
                raise subprocess.CalledProcessError(proc.returncode, cmd)
            objs.append(obj)
        tmp = f"{LIB}.{os.getpid()}.tmp"
        cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", tmp]
        if verbose:
            print("[hyena_dna_amd.build]", " ".join(cmd), flush=True)
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

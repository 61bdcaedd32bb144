"""Build the CPU emulation of the HIP kernels (TEST INFRASTRUCTURE ONLY; see hipemu.h)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_emu_fftconv_test_only.so")
SRC = [os.path.join(ROOT, "hyena_dna_amd", "csrc", "fftconv.hip"), os.path.join(ROOT, "hyena_dna_amd", "csrc", "onchip.hip"),
       os.path.join(ROOT, "hyena_dna_amd", "csrc", "onchip_dk.hip"),
       os.path.join(ROOT, "hyena_dna_amd", "csrc", "cm.hip"), os.path.join(ROOT, "hyena_dna_amd", "csrc", "proj.hip"),
       os.path.join(ROOT, "hyena_dna_amd", "csrc", "filter16.hip"), os.path.join(HERE, "hipemu.cpp")]
DEPS = SRC + [os.path.join(ROOT, "hyena_dna_amd", "csrc", "fftconv_kernels.h"), os.path.join(ROOT, "hyena_dna_amd", "csrc", "onchip_kernels.h"),
              os.path.join(ROOT, "hyena_dna_amd", "csrc", "onchip_host.h"), os.path.join(ROOT, "hyena_dna_amd", "csrc", "launch.h"), os.path.join(ROOT, "hyena_dna_amd", "csrc", "cm_kernels.h"),
              os.path.join(ROOT, "hyena_dna_amd", "csrc", "mixer_kernels.h"),
              os.path.join(ROOT, "hyena_dna_amd", "csrc", "filter_kernels.h"), os.path.join(ROOT, "hyena_dna_amd", "csrc", "filter16_kernels.h"), os.path.join(ROOT, "hyena_dna_amd", "csrc", "block_kernels.h"),
              os.path.join(ROOT, "include", "hyena_block.h"), os.path.join(ROOT, "include", "hyena_filter.h"), os.path.join(HERE, "hipemu.h"),
              os.path.join(ROOT, "include", "hyena_fftconv.h"), os.path.join(ROOT, "include", "hyena_mixer.h"),
              os.path.join(ROOT, "hyena_dna_amd", "csrc", "proj_kernels.h"), os.path.join(ROOT, "hyena_dna_amd", "csrc", "proj2_kernels.h"), os.path.join(ROOT, "include", "hyena_proj.h")]


def _fresh():
    return os.path.exists(OUT) and all(os.path.getmtime(f) <= os.path.getmtime(OUT) for f in DEPS)


def build(force=False):
    """Several processes may ask at once (pytest -n in a fresh checkout): one builds, under a file lock, into a temporary name that is
    renamed into place when complete -- nobody ever loads a half-written library."""
    if not force and _fresh():
        return OUT
    import fcntl
    with open(OUT + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and _fresh():                    # somebody else built it while this process waited
            return OUT
        tmp = f"{OUT}.{os.getpid()}.tmp"
        cmd = ["g++", "-x", "c++", "-DHIPEMU", "-std=c++17", "-O2", "-fopenmp", "-fPIC", "-shared",
               "-Wno-unknown-pragmas", "-Wno-psabi", "-I", HERE] + SRC + ["-o", tmp]
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, OUT)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return OUT


if __name__ == "__main__":
    print(build(force=True))

"""The asynchronous register loads of csrc/proj_kernels.h (gld16_async ... PJ_VMWAIT_FOR) in the generated code: between the load's
inline assembly and the wait that releases its registers, no instruction may touch those registers (a compiler-inserted copy there would
read them before the data has landed).   usage: python scripts/check_async_loads.py [file.s]   (without a file: compiles csrc/proj.hip)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs_of(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def line_regs(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", line.split(";")[0]):
        out |= regs_of(tok)
    return out


def check(path):
    lines = open(path).read().splitlines()
    kernel, bad, n_loads = None, [], 0
    in_asm = False
    pending = []                                   # (line number, register set) of loads in flight
    for i, raw in enumerate(lines):
        line = raw.strip()
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            kernel, pending = m.group(1), []
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line[0] in ".;" or line.endswith(":"):
            continue
        if in_asm and line.startswith("global_load_dwordx4") and "s_waitcnt" not in lines[i + 1]:
            dst = regs_of(line.split()[1].rstrip(","))
            pending = [p for p in pending if not (p[1] & dst)]
            pending.append((i + 1, dst))
            n_loads += 1
            continue
        if in_asm and line.startswith("s_waitcnt") and "releases" in raw:
            rel = set()
            for tok in re.findall(r"v\[\d+:\d+\]", raw.split("releases")[1]):
                rel |= regs_of(tok)
            pending = [p for p in pending if not (p[1] & rel)]
            continue
        if line.startswith("s_endpgm"):
            pending = []
            continue
        used = line_regs(line)
        for ln, dst in pending:
            if used & dst:
                bad.append((kernel, ln, i + 1, line))
    return n_loads, bad


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = "/tmp/proj_check.s"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-S",
                               "--cuda-device-only", os.path.join(ROOT, "hyena_dna_amd/csrc/proj.hip"), "-o", path], stderr=subprocess.DEVNULL)
    n, bad = check(path)
    print(f"{n} asynchronous loads; {len(bad)} touched before their wait")
    for k, ln, at, line in bad[:20]:
        print(f"  {k}: load at line {ln}, registers touched at line {at}: {line}")
    return 1 if bad or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())

"""Instruction histogram of one kernel: python scripts/isa_hist.py <file.hip> <mangled-substring> [-D...] (hipcc -S, device only)."""
import collections
import re
import subprocess
import sys

src, pat = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                src, "-o", "/tmp/_isa.s"] + extra, check=True, capture_output=True)
lines = open("/tmp/_isa.s").read().splitlines()
for i, l in enumerate(lines):
    if l.startswith("_Z") and pat in l.split(":")[0] and l.split(":")[0].endswith("E") and ":" in l:
        c = collections.Counter()
        for m in lines[i + 1:]:
            if "s_endpgm" in m:
                break
            g = re.match(r"\s+([a-z_0-9]+)\s", m + " ")
            if g and not m.strip().startswith((".", ";")):
                c[g.group(1)] += 1
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        print(l.split(":")[0], "total", sum(c.values()), "valu", valu, "ds", sum(v for k, v in c.items() if k.startswith("ds_")),
              "vmem", sum(v for k, v in c.items() if k.startswith(("buffer", "global", "scratch"))), "salu", sum(v for k, v in c.items() if k.startswith("s_")))
        print("   ", ", ".join(f"{k} {v}" for k, v in c.most_common(18)))

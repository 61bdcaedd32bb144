"""Instruction mix of one kernel in gfx950 assembly: python scripts/asm_mix.py x.s <mangled-name-substring>"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and key in l)
mix = collections.Counter()
n = 0
for l in txt[start + 1:]:
    t = l.strip()
    if t.startswith("s_endpgm"):
        break
    if not l.startswith("\t") or t.startswith((";", ".")):
        continue
    op = t.split()[0]
    n += 1
    cls = ("VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_barrier", "s_load", "s_buffer", "s_nop"))
           else "LDS" if op.startswith("ds_") else "VMEM" if op.startswith(("buffer_", "global_", "scratch_", "flat_")) else op)
    mix[cls] += 1
    if cls in ("VALU",):
        mix["  " + re.sub(r"_e(32|64)$", "", op)] += 1
print("total", n)
for k, v in sorted(mix.items(), key=lambda kv: (-kv[1] if not kv[0].startswith("  ") else 0, kv[0])):
    if not k.startswith("  "):
        print(f"{k:12s} {v}")
print("top VALU ops:", ", ".join(f"{k.strip()}={v}" for k, v in sorted(((k, v) for k, v in mix.items() if k.startswith("  ")), key=lambda kv: -kv[1])[:14]))

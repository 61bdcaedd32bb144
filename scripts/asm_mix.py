"""Instruction mix per kernel from hipcc -S output: python scripts/asm_mix.py file.s substring [substring ...]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().splitlines()
starts = [(i, l.split(":")[0]) for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l)]
for pat in sys.argv[2:]:
    for n, (i, name) in enumerate(starts):
        if pat not in name:
            continue
        end = starts[n + 1][0] if n + 1 < len(starts) else len(txt)
        c = collections.Counter()
        for line in txt[i + 1:end]:
            line = line.strip()
            if not line or line[0] in ".;/" or line.endswith(":"):
                continue
            op = line.split()[0]
            if op == "s_endpgm":
                break
            if op.startswith("v_mfma"):
                c["mfma"] += 1
            elif op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")):
                c["trans"] += 1
            elif op.startswith("v_cvt"):
                c["cvt:" + op] += 1
            elif op.startswith("v_"):
                c["valu"] += 1
            elif op.startswith("ds_"):
                c["lds:" + op] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                c["mem:" + op] += 1
            elif op.startswith("s_waitcnt"):
                c["waitcnt"] += 1
            elif op.startswith("s_barrier"):
                c["barrier"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
        print(name)
        for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
            print(f"    {k:40s} {v}")

"""Where a kernel spills: sequence of barriers / spills / loads / stores in the gfx950 assembly of one kernel.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -o x.s file.hip
    python scripts/asm_events.py x.s <mangled-name-substring>
"""
import re
import sys

txt = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]))
body = []
for l in txt[start + 1:]:
    body.append(l)
    if "s_endpgm" in l:
        break
print("instructions:", sum(1 for l in body if l.startswith("\t") and not l.strip().startswith((";", "."))))
events, bar = [], 0
for i, l in enumerate(body):
    t = l.strip()
    if t.startswith("s_barrier"):
        bar += 1
        events.append((i, "BARRIER"))
    elif t.startswith("scratch_store"): events.append((i, "spill-st"))
    elif t.startswith("scratch_load"): events.append((i, "spill-ld"))
    elif re.match(r"buffer_load_(u?short|dword )", t): events.append((i, "in-load"))
    elif re.match(r"buffer_store_(short|dword )", t): events.append((i, "out-store"))
    elif t.startswith("buffer_load_dwordx2"): events.append((i, "ld64"))
    elif t.startswith("buffer_store_dwordx2"): events.append((i, "st64"))
    elif t.startswith("ds_write") or t.startswith("ds_read"): events.append((i, t.split()[0][:8]))
last, cnt, st = None, 0, 0
for i, e in events + [(len(body), "END")]:
    if e == last:
        cnt += 1
    else:
        if last:
            print(f"{st:6d} {last} x{cnt}")
        last, cnt, st = e, 1, i

"""Print VGPR / spill / scratch / occupancy per kernel of one HIP translation unit (hipcc -Rpass-analysis=kernel-resource-usage).

    python scripts/kernel_resources.py hyena_dna_amd/csrc/onchip.hip [filter-substring] [-D...]
"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = [a for a in sys.argv[2:] if not a.startswith("-")]
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kr.o"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*\)$", "", name).replace("void hyena::", "")}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print(f"{'kernel':58s} {'VGPR':>5s} {'spill':>5s} {'scratch':>7s} {'SGPR':>5s} {'sspill':>6s} {'occ':>3s}")
for r in rows:
    if flt and not any(f in r["name"] for f in flt):
        continue
    print(f"{r['name'][:58]:58s} {r.get('VGPRs','?'):>5s} {r.get('VGPRs Spill','?'):>5s} {r.get('ScratchSize [bytes/lane]','?'):>7s} "
          f"{r.get('SGPRs','?'):>5s} {r.get('SGPRs Spill','?'):>6s} {r.get('Occupancy [waves/SIMD]','?'):>3s}")

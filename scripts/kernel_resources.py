"""Per-kernel register / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    python scripts/kernel_resources.py hyena_dna_amd/csrc/filter16.hip [extra hipcc flags]
"""
import re
import subprocess
import sys

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage"]


def main():
    src, extra = sys.argv[1], sys.argv[2:]
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-c", src, "-o", "/dev/null"], capture_output=True, text=True)
    txt = out.stderr
    rows, cur = [], None
    for line in txt.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            if "error" in line:
                print(line)
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:") or body.startswith("Name:"):
            cur = {"name": body.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            cur[k.strip()] = v.strip()
    demangled = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print(f"{'kernel':90s} {'VGPR':>5s} {'AGPR':>5s} {'spill':>6s} {'occ':>4s} {'LDS':>7s}")
    for r, n in zip(rows, demangled):
        n = re.sub(r"\(.*\)$", "", n)[:90]
        print(f"{n:90s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('VGPRs Spill', '?'):>6s} {r.get('Occupancy [waves/SIMD]', '?'):>4s} {r.get('LDS Size [bytes/block]', '?'):>7s} sgpr {r.get('SGPRs', '?')} sspill {r.get('SGPRs Spill', '?')}")


if __name__ == "__main__":
    main()

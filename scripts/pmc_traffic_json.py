"""Build profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE csvs of scripts/gpu_pmc_cfg.sh.
usage: python scripts/pmc_traffic_json.py <tag>:<L>:<B>:<D> ...      (csvs under gpurun_out/<tag>/; 4 profiled steps = 1 warm-up + 3)
Counter unit: KB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-byte requests at 64 bytes -- doubled
for every kernel whose reads are full 128-byte wavefront requests; kept x1 for col_fwd<1024, bf16> (4-byte-per-lane pair loads, 256 B
per wavefront request, calibrated against the tensor size in round 1)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import KERNEL_SET  # noqa: E402  (the generation of each plan's kernels in this tree: the record is valid for it only)
STEPS = 4
out = []
for spec in sys.argv[1:]:
    tag, L, B, D = spec.split(":")
    L, B, D = int(L), int(B), int(D)
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        for row in csv.DictReader(open(os.path.join(ROOT, "gpurun_out", tag, counter + ".csv"))):
            if "hyena" not in row["kernel"]:
                continue
            k = per.setdefault(row["kernel"], {"fetch_bytes_per_step": 0.0, "write_bytes_per_step": 0.0, "fetch_correction": 2.0})
            val = float(row["sum"]) * 1024.0 / STEPS
            if counter == "FETCH_SIZE":
                if "col_fwd_kernel<1024, 1>" in row["kernel"]:
                    k["fetch_correction"] = 1.0
                k["fetch_bytes_per_step"] = val * k["fetch_correction"]
            else:
                k["write_bytes_per_step"] = val
    total = sum(v["fetch_bytes_per_step"] + v["write_bytes_per_step"] for v in per.values())
    alg = 5 * B * D * L * 2 + 12 * D * L + 8 * D
    out.append({"config": {"seq_len": L, "channels": D, "batch_per_gpu": B, "io_dtype": "bf16", "save_spectra": True},
                "traffic_bytes_per_step": total, "algorithmic_bytes_per_step": alg, "ratio": total / alg,
                "source": f"gpurun_out/{tag}/FETCH_SIZE.csv, WRITE_SIZE.csv (scripts/gpu_pmc_cfg.sh)",
                "kernel_set": ("onchip-" + KERNEL_SET["onchip"]) if L <= 32768 else ("twolevel-" + KERNEL_SET["twolevel"]), "per_kernel": per})
doc = {"method": __doc__.split("usage")[0].strip() + "  " + __doc__.split("Counter unit")[1].strip().join(["Counter unit", ""]) if False else
       "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) of bench.py, summed over all hyena kernels of one "
       "fwd+bwd step; counter unit KB; FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md HBM section) "
       "except for col_fwd<1024, bf16> (4-byte pair loads, calibrated x1 against the tensor size in round 1); scripts/gpu_pmc_cfg.sh, "
       "scripts/pmc_traffic_json.py",
       "configs": out}
# configurations not re-measured in this call keep their records (each carries the kernel generation it was taken on)
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
if os.path.exists(path):
    fresh = [json.dumps(c["config"], sort_keys=True) for c in out]
    for c in json.load(open(path)).get("configs", []):
        if json.dumps(c["config"], sort_keys=True) not in fresh:
            doc["configs"].append(c)
json.dump(doc, open(path, "w"), indent=1)
for c in out:
    print(c["config"], "traffic %.3f GB" % (c["traffic_bytes_per_step"] / 1e9), "algorithmic %.3f GB" % (c["algorithmic_bytes_per_step"] / 1e9), "ratio %.2f" % c["ratio"])

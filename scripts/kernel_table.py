"""Per-kernel bytes and rates of one configuration: rocprofv3 kernel stats + the two PMC passes -> a markdown table.
usage: python scripts/kernel_table.py <stats.csv> <FETCH_SIZE.csv> <WRITE_SIZE.csv>
(FETCH_SIZE doubled per the gfx950 correction except col_fwd<1024, bf16>: see scripts/pmc_traffic_json.py)"""
import csv
import sys

stats, fetch, write = sys.argv[1:4]
t = {r["Name"]: (int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(open(stats)) if "hyena" in r["Name"]}
by = {}
for path, key in ((fetch, "r"), (write, "w")):
    for r in csv.DictReader(open(path)):
        if "hyena" not in r["kernel"]:
            continue
        v = float(r["sum"]) * 1024.0 / float(r["dispatches"])
        if key == "r" and "col_fwd_kernel<1024, 1>" not in r["kernel"]:
            v *= 2.0
        by.setdefault(r["kernel"], {})[key] = v
print("| kernel | launches/step | read MB | written MB | time | TB/s |")
print("|---|---|---|---|---|---|")
tot_b = tot_t = 0.0
for k, (calls, avg) in sorted(t.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    b = by.get(k, {})
    rd, wr = b.get("r", 0.0), b.get("w", 0.0)
    per_step = calls / 12.0
    tot_b += (rd + wr) * per_step
    tot_t += avg * per_step
    name = k.replace("void hyena::", "").replace("hyena::", "").split("(")[0]
    print(f"| `{name}` | {per_step:g} | {rd / 1e6:.0f} | {wr / 1e6:.0f} | {avg / 1e3:.0f} us | {(rd + wr) / avg / 1e3:.2f} |")
print(f"| total per step | | {tot_b / 1e9:.2f} GB | | {tot_t / 1e6:.3f} ms | {tot_b / tot_t / 1e3:.2f} |")

#!/bin/bash
# round 6: per-kernel difference of one operator layer at the trainer's odd lengths against the aligned neighbour (B > 1)
for cfg in "32767 8" "32768 8" "1023 256" "1024 256"; do
  set -- $cfg
  d=256; [ $2 = 256 ] && d=128
  bash scripts/gpu_prof_operator.sh r6u_$1_$2 $1 $2 fused $d > /dev/null 2>&1
done
python - <<'PY'
import csv, os
def load(p):
    return {r['Name']: (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3) for r in csv.DictReader(open(p))}
for a, b in (("32767_8", "32768_8"), ("1023_256", "1024_256")):
    A, Bq = load(f"gpurun_out/r6u_{a}/op_stats.csv"), load(f"gpurun_out/r6u_{b}/op_stats.csv")
    print("====", a, "vs", b, " total us per step:", sum(v[2] for v in A.values()) / 13, sum(v[2] for v in Bq.values()) / 13)
    names = sorted(set(A) | set(Bq), key=lambda n: -(A.get(n, (0, 0, 0))[2] - Bq.get(n, (0, 0, 0))[2]))
    for n in names[:14] + names[-4:]:
        x, y = A.get(n, (0, 0, 0)), Bq.get(n, (0, 0, 0))
        print(f"{n[:100]:100s} calls {x[0]:4d}/{y[0]:4d} avg {x[1]:8.1f}/{y[1]:8.1f}  d_total/step {(x[2] - y[2]) / 13:8.1f}")
PY

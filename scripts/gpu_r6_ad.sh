#!/bin/bash
# round 6: per-kernel difference of one operator layer at 2^20 - 1 against 2^20 (B = 1)
for cfg in "1048575 1" "1048576 1"; do
  set -- $cfg
  bash scripts/gpu_prof_operator.sh r6ad_$1_$2 $1 $2 fused 256 > /dev/null 2>&1
done
python - <<'PY'
import csv
def load(p):
    return {r['Name']: (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3) for r in csv.DictReader(open(p))}
A, Bq = load("gpurun_out/r6ad_1048575_1/op_stats.csv"), load("gpurun_out/r6ad_1048576_1/op_stats.csv")
print("total us per step:", sum(v[2] for v in A.values()) / 13, sum(v[2] for v in Bq.values()) / 13)
names = sorted(set(A) | set(Bq), key=lambda n: -(A.get(n, (0, 0, 0))[2] - Bq.get(n, (0, 0, 0))[2]))
for n in names[:14] + names[-5:]:
    x, y = A.get(n, (0, 0, 0)), Bq.get(n, (0, 0, 0))
    print(f"{n[:95]:95s} calls {x[0]:4d}/{y[0]:4d} avg {x[1]:8.1f}/{y[1]:8.1f}  d/step {(x[2] - y[2]) / 13:8.1f}")
PY
rm -rf gpurun_out/r6ad_*/prof

#!/bin/bash
# round 3, GPU call 2: the new GPU tests, the one-launch kernels at L = 1024, the MFMA projection / MLP kernels against the
# library path (micro + whole layer + whole model step), the corrected L2-exchange follow-up, the graphed runner
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r3b; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_small.py tests/test_gpu_proj.py tests/test_gpu_contract.py tests/test_gpu_seqlen.py -q -m gpu --durations=25 > $OUT/pytest_new.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_new.txt
tail -45 $OUT/pytest_new.txt | cut -c1-220
echo "== oc_times (small kernels on)"; timeout 200 python scripts/oc_times.py "1024 8 128" "1024 8 256" "2048 8 256" "32768 8 256" 2>&1 | grep "L=" | tee $OUT/oc_times.txt
echo "== oc_times (HYENA_FFTCONV_SMALL=0)"; HYENA_FFTCONV_SMALL=0 timeout 200 python scripts/oc_times.py "1024 8 128" "1024 8 256" "2048 8 256" 2>&1 | grep "L=" | tee -a $OUT/oc_times.txt
echo "== bench_proj"; timeout 300 python scripts/bench_proj.py "1048576 1 256" "32768 8 256" "160000 2 256" "1024 8 128" 2>&1 | grep -v Warning | tee $OUT/bench_proj.txt
echo "== bench 1M (new kernels)"; timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_1m.json 2> $OUT/bench_1m.err; python - <<PY
import json
r = json.loads([l for l in open("$OUT/bench_1m.json") if l.startswith("{")][-1])
print("conv ms", r["ms_per_step"], "frac", r["roofline"]["frac"], "| layer", r.get("operator_layer", {}).get("ms_per_step"), "| model", (r.get("model_step") or {}).get("ms_per_step"), (r.get("model_step") or {}).get("error"))
PY
echo "== bench 1M (HYENA_INPROJ_MFMA=0 HYENA_FUSED_MLP=0)"; HYENA_INPROJ_MFMA=0 HYENA_FUSED_MLP=0 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_1m_lib.json 2> $OUT/bench_1m_lib.err; python - <<PY
import json
r = json.loads([l for l in open("$OUT/bench_1m_lib.json") if l.startswith("{")][-1])
print("conv ms", r["ms_per_step"], "| layer", r.get("operator_layer", {}).get("ms_per_step"), "| model", (r.get("model_step") or {}).get("ms_per_step"), (r.get("model_step") or {}).get("error"))
PY
echo "== bench 1k"; timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --seq-len 1024 --batch 8 --d-model 128 > $OUT/bench_1k.json 2> $OUT/bench_1k.err; python - <<PY
import json
r = json.loads([l for l in open("$OUT/bench_1k.json") if l.startswith("{")][-1])
print("conv ms", r["ms_per_step"], "frac", r["roofline"]["frac"], "| layer", r.get("operator_layer", {}).get("ms_per_step"), "| model", (r.get("model_step") or {}).get("ms_per_step"), (r.get("model_step") or {}).get("graphed"))
PY
timeout 200 ./build/xcd_flags > $OUT/xcd_flags.txt 2>&1; grep -A3 "consumer" $OUT/xcd_flags.txt | head -60
timeout 300 python scripts/train_hg38.py --steps 30 --graphed --synthetic-genome /tmp/genome2 dataset.max_length=32768 dataset.batch_size=4 \
  trainer.accumulate_grad_batches=1 model.d_model=256 model.n_layer=8 model.fused_dropout_add_ln=true scheduler.warmup_t=5 scheduler.t_initial=200 \
  > $OUT/runner_graphed.txt 2>&1; echo "runner exit $?" >> $OUT/runner_graphed.txt; tail -4 $OUT/runner_graphed.txt | cut -c1-300

#!/bin/bash
# round 3, GPU call 1: whole -m gpu suite (with the new contract-shape / seqlen tests), smoke, the runner on a synthetic genome,
# the one L2-exchange follow-up, kernel stats + PMC traffic at BASELINE configs 3 and 4
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r3a; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -q -m gpu --durations=15 > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt
tail -30 $OUT/pytest_gpu.txt
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.txt 2>&1; echo "smoke exit $?" >> $OUT/smoke.txt; tail -3 $OUT/smoke.txt
timeout 600 python scripts/train_hg38.py --steps 50 --synthetic-genome /tmp/genome dataset.max_length=32768 dataset.batch_size=4 \
  trainer.accumulate_grad_batches=1 model.d_model=256 model.n_layer=8 model.fused_dropout_add_ln=true scheduler.warmup_t=5 scheduler.t_initial=200 \
  > $OUT/runner.txt 2>&1; echo "runner exit $?" >> $OUT/runner.txt; tail -8 $OUT/runner.txt | cut -c1-400
timeout 300 python scripts/train_hg38.py --steps 30 --graphed --synthetic-genome /tmp/genome2 dataset.max_length=32768 dataset.batch_size=4 \
  trainer.accumulate_grad_batches=1 model.d_model=256 model.n_layer=8 model.fused_dropout_add_ln=true scheduler.warmup_t=5 scheduler.t_initial=200 \
  > $OUT/runner_graphed.txt 2>&1; echo "runner exit $?" >> $OUT/runner_graphed.txt; tail -4 $OUT/runner_graphed.txt | cut -c1-400
timeout 300 ./build/xcd_flags > $OUT/xcd_flags.txt 2>&1; tail -70 $OUT/xcd_flags.txt
bash scripts/gpu_prof_bench.sh r3a_160k --seq-len 160000 --batch 2 --no-operator --no-model
bash scripts/gpu_prof_bench.sh r3a_450k --seq-len 450560 --batch 1 --no-operator --no-model
bash scripts/gpu_pmc_cfg.sh r3a_pmc160k 160000 2 256
bash scripts/gpu_pmc_cfg.sh r3a_pmc450k 450560 1 256

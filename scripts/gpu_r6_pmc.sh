#!/bin/bash
# round 6: HBM traffic counters (FETCH_SIZE / WRITE_SIZE, separate passes) at the five contract configurations on the final kernels, kernel stats
# (rocprofv3 --kernel-trace --stats) of the default bench command's timed region at 2^20 and 32768 x 8, and of the model step (8 layers, 2^20)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash scripts/gpu_pmc_cfg.sh r6p_1k 1024 8 128
bash scripts/gpu_pmc_cfg.sh r6p_32k 32768 8 256
bash scripts/gpu_pmc_cfg.sh r6p_160k 160000 2 256
bash scripts/gpu_pmc_cfg.sh r6p_450k 450560 1 256
bash scripts/gpu_pmc_cfg.sh r6p_1m 1048576 1 256
bash scripts/gpu_prof_bench.sh r6z_bench1m --no-operator --no-model --no-sweep | tail -14
bash scripts/gpu_prof_bench.sh r6z_bench32k --no-operator --no-model --no-sweep --no-graph --seq-len 32768 --batch 8 | tail -8
bash scripts/gpu_prof_bench.sh r6z_bench16k --no-operator --no-model --no-sweep --no-graph --seq-len 16384 --batch 8 | tail -8
bash scripts/gpu_prof_model.sh r6m_model 1048576 1 256 6 | tail -3

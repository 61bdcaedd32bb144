#!/bin/bash
# round 5: HBM traffic counters (FETCH_SIZE / WRITE_SIZE, separate passes) at the five contract configurations on the final kernels,
# then kernel stats of the model step (8 layers, L = 2^20 - 1: the length the reference's trainer feeds it) and of the default bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash scripts/gpu_pmc_cfg.sh r5p_1k 1024 8 128
bash scripts/gpu_pmc_cfg.sh r5p_32k 32768 8 256
bash scripts/gpu_pmc_cfg.sh r5p_160k 160000 2 256
bash scripts/gpu_pmc_cfg.sh r5p_450k 450560 1 256
bash scripts/gpu_pmc_cfg.sh r5p_1m 1048576 1 256
bash scripts/gpu_prof_model.sh r5m_model 1048576 1 256 4 | tail -3
bash scripts/gpu_prof_model.sh r5m_model_real 1048575 1 256 4 | tail -3

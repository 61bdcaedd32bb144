#!/bin/bash
# round 4: HBM traffic counters (FETCH_SIZE / WRITE_SIZE, separate passes) at the five contract configurations on the final kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash scripts/gpu_pmc_cfg.sh r4p_1k 1024 8 128
bash scripts/gpu_pmc_cfg.sh r4p_32k 32768 8 256
bash scripts/gpu_pmc_cfg.sh r4p_160k 160000 2 256
bash scripts/gpu_pmc_cfg.sh r4p_450k 450560 1 256
bash scripts/gpu_pmc_cfg.sh r4p_1m 1048576 1 256

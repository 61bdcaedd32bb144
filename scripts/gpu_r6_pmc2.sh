#!/bin/bash
# round 6: HBM traffic counters at the five contract configurations on the LAST sources (fftconv.hip changed with add_norm_bwd's column sums: the
# record's kernel-set hash must follow)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash scripts/gpu_pmc_cfg.sh r6p_1k 1024 8 128
bash scripts/gpu_pmc_cfg.sh r6p_32k 32768 8 256
bash scripts/gpu_pmc_cfg.sh r6p_160k 160000 2 256
bash scripts/gpu_pmc_cfg.sh r6p_450k 450560 1 256
bash scripts/gpu_pmc_cfg.sh r6p_1m 1048576 1 256

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash scripts/gpu_prof_model.sh r6m_model_1023 1023 256 128 10 2 | tail -60

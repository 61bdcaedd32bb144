#!/bin/bash
# rocprofv3 kernel stats of one order-3 operator layer at 2^20 x 256 (channel-major route)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6x; mkdir -p $OUT
cat > /tmp/o3.py <<PY
import os, sys
sys.path.insert(0, "$R")
import torch
import hyena_dna_amd.hyena as H
dev = torch.device("cuda", 0)
L, D = 1 << 20, 256
op = H.HyenaOperator(d_model=D, l_max=L + 2, order=3, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10).to(dev)
u = torch.randn(1, L, D, device=dev).to(torch.bfloat16)
dy = torch.randn(1, L, D, device=dev).to(torch.bfloat16)
for _ in range(10):
    op.zero_grad(set_to_none=True)
    ud = u.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = op(ud)
    y.backward(dy)
torch.cuda.synchronize()
PY
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o op -- python /tmp/o3.py > $OUT/log.txt 2>&1
python $R/scripts/rocpd_stats.py $(find $OUT/prof -name '*.db' | head -1) $OUT/order3_stats.csv | head -40 | cut -c1-170
find $OUT/prof -name '*.db' -delete

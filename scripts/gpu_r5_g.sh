#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for lib in "" "$R/build/libhyena_oldfma.so"; do
if [ -n "$lib" ]; then export HYENA_FFTCONV_LIB=$lib; fi; python - <<PY
import os, torch
from tests._tiny_lm import train
l = train("cuda", steps=40, d=128, L=2048, B=4, n_layer=2, autocast_dtype=torch.bfloat16)
print(os.environ.get("HYENA_FFTCONV_LIB") or "product", [round(x, 2) for x in l[:3]], [round(x, 2) for x in l[-5:]], l[-1] / l[0])
PY
done

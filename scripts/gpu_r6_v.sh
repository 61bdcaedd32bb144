#!/bin/bash
out=gpurun_out/${1:-r6v}; mkdir -p $out
python -m pytest tests/test_gpu_proj.py tests/test_gpu_contract.py -q -x 2>&1 | tail -5 > $out/pytest.txt; cat $out/pytest.txt
for al in 1 0; do
  echo "== HYENA_PROJ_ALIGNED_PIECES=$al" >> $out/bench_outproj.txt
  HYENA_PROJ_ALIGNED_PIECES=$al python scripts/bench_outproj.py "32767 8 256" "32768 8 256" "1023 256 128" "1024 256 128" "159999 2 256" "160000 2 256" "1048575 1 256" >> $out/bench_outproj.txt 2>&1
done
cat $out/bench_outproj.txt

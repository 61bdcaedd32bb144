#!/bin/bash
out=gpurun_out/${1:-r6s}; mkdir -p $out
timeout 1500 python scripts/tune_gemms.py 1048576 1 256 2 $out/tunableop_1m.csv > $out/tune_1m.txt 2>&1
tail -40 $out/tune_1m.txt

#!/bin/bash
out=gpurun_out/${1:-r6t}; mkdir -p $out
python -m pytest tests/test_gpu_contract.py -q -x -k "higher_order" 2>&1 | tail -15 > $out/pytest_orders.txt
cat $out/pytest_orders.txt
python -m pytest tests/test_gpu_cm.py -q -x 2>&1 | tail -3
timeout 900 python scripts/bench_order3.py > $out/bench_order3.txt 2>&1
cat $out/bench_order3.txt

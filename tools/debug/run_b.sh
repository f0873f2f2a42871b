#!/bin/bash
mkdir -p gpurun_out/run_b
timeout 125 python -m pytest tests -q -m gpu --tb=line 2>&1 | tail -8 > gpurun_out/run_b/tests.txt
timeout 70 python bench.py > gpurun_out/run_b/bench_c2.json 2> gpurun_out/run_b/bench_c2.err
cat gpurun_out/run_b/tests.txt
cat gpurun_out/run_b/bench_c2.json

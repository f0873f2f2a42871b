#!/bin/bash
# First GPU call of a session: validate and A/B every opt-in kernel switch in one go (summaries under gpurun_out/probe/).
#   gpurun --timeout 420 -- 'bash tools/probe_optin.sh'
set -u
out=gpurun_out/probe
mkdir -p $out
# 1. opt-in kernels against ATen (own processes: the library reads its switches once)
CAT_EXPERIMENTAL=1 timeout 200 python -m pytest tests/test_experimental_gpu.py -q --tb=short 2>&1 | tail -15 > $out/experimental_tests.txt
# 2. LDS-tile conv vs the im2col kernel on the layers it targets
for v in 0 1; do CAT_CONV_TILE=$v timeout 60 python tools/debug/check_conv_tile.py --bench 2>&1 | grep -v amdgpu.ids; done > $out/conv_tile_bench.txt
# 3. whole step + student forward with the switch on (no CPU leg, short)
for v in 0 1; do
  CAT_CONV_TILE=$v timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null |
    python -c "import json,sys; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('CAT_CONV_TILE=$v', d['value'], 'img/s', d['ms_per_step'], 'ms; student fwd', d['student_forward']['ms'], 'ms', d['student_forward']['tflops'], 'TF')"
done > $out/bench_ab.txt 2>&1
# 3a. 32-bit element walks in the norm / affine / depthwise-wgrad kernels: kernel + model parity with the switch on, then the step
CAT_IDX32=1 timeout 150 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q --tb=line 2>&1 | tail -5 > $out/idx32_tests.txt
CAT_IDX32=1 timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null |
  python -c "import json,sys; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('CAT_IDX32=1', d['value'], 'img/s', d['ms_per_step'], 'ms; student fwd', d['student_forward']['ms'], 'ms')" >> $out/bench_ab.txt 2>&1
# 3b. N-tile choice by padded-N cost (frozen teacher's 176-wide GEMM)
CAT_TILE_BY_PAD=1 timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null |
  python -c "import json,sys; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('CAT_TILE_BY_PAD=1', d['value'], 'img/s', d['ms_per_step'], 'ms')" >> $out/bench_ab.txt 2>&1
# 4. wgrad pixel-split plan on the student's layers
timeout 60 python tools/debug/wgrad_blocks.py 2>&1 | grep -v amdgpu.ids > $out/wgrad_blocks.txt
tail -n +1 $out/*.txt

#!/bin/bash
# one-shot validation of the opt-in BK=32 dgrad / wgrad kernels and the channel-split PatchGAN head (GPU box)
mkdir -p gpurun_out/run_a
export CAT_DGRAD_BK32=1 CAT_WGRAD_BK32=1 CAT_SMALLCO_SPLITK=1
timeout 170 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q --tb=line 2>&1 | tail -25 > gpurun_out/run_a/tests_on.txt
timeout 60 python tools/conv_bench.py --shapes big --iters 10 > gpurun_out/run_a/conv_on.txt 2>&1
timeout 60 python bench.py --steps 20 --warmup 5 > gpurun_out/run_a/bench_on.json 2> gpurun_out/run_a/bench_on.err
unset CAT_DGRAD_BK32 CAT_WGRAD_BK32 CAT_SMALLCO_SPLITK
timeout 60 python tools/conv_bench.py --shapes big --iters 10 > gpurun_out/run_a/conv_off.txt 2>&1
cat gpurun_out/run_a/tests_on.txt
paste -d'|' gpurun_out/run_a/conv_off.txt gpurun_out/run_a/conv_on.txt | cut -c1-160
cat gpurun_out/run_a/bench_on.json | cut -c1-400
tail -3 gpurun_out/run_a/bench_on.err

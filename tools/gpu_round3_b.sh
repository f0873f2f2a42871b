set -x
export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
timeout 1200 python -m pytest tests/test_graph_gpu.py tests/test_headline_gpu.py tests/test_fused_block_gpu.py "tests/test_dp_gpu.py::test_single_rank_rccl_schedule_is_bit_identical_to_plain_step" -m gpu -x -q -s -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-kernel-profile --steps 20 --warmup 5 --sustained-steps 0"
$B > $O/b_plain.json 2> $O/b_plain.err
$B --teacher-side-stream 1 > $O/b_tss.json 2> $O/b_tss.err
$B --dp-schedule 1 > $O/b_dp.json 2> $O/b_dp.err
for f in plain tss dp; do python - <<P
import json
try:
    d=json.load(open('$O/b_$f.json')); print('$f', d['value'], d['ms_per_step'], d['config']['launch'])
except Exception as e: print('$f', 'ERR', e)
P
done
grep -h "student gradients\|headline parity\|\[dp\]\|gpu64\|updated weights" $O/pytest.log | cut -c1-400

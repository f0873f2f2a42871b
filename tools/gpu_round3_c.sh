set -x
export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
timeout 1500 python -m pytest "tests/test_graph_gpu.py::test_segment_graphs_without_reducer_equal_eager" "tests/test_headline_gpu.py::test_c3_step_at_256_matches_oracle" tests/test_fused_block_gpu.py tests/test_dp_gpu.py "tests/test_headline_gpu.py::test_spade_step_at_512x256_matches_oracle" -m gpu -q -s -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-kernel-profile --steps 20 --warmup 5 --sustained-steps 0"
for i in 1 2; do
$B > $O/b_plain$i.json 2> $O/b_plain$i.err
$B --segments 1 > $O/b_seg$i.json 2> $O/b_seg$i.err
done
for f in plain1 seg1 plain2 seg2; do python - <<P
import json
try:
    d=json.load(open('$O/b_$f.json')); print('$f', d['value'], d['ms_per_step'], d['config']['launch'][:40])
except Exception as e: print('$f', 'ERR', e)
P
done
grep -h "gradients\|headline parity\|\[dp\]\|gpu64\|ref64\|weights\]\|updated weights\|FAILED\|Error" $O/pytest.log | cut -c1-420

set -x
export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
timeout 600 python -m pytest tests/test_spade_gpu.py tests/test_fused_block_gpu.py -m gpu -q -x -k "fused" -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
B="python bench.py --workload spade --no-cpu-baseline --steps 10 --warmup 3 --sustained-steps 0"
$B > $O/b_fused.json 2> $O/b_fused.err
python - <<P
import json
try:
    d=json.load(open('$O/b_fused.json')); fam=d['roofline']['families']
    print('fused', d['value'], d['ms_per_step'], d['student_forward'], 'launches', sum(v['launches_per_step'] for v in fam.values()), 'serial ms', round(sum(v['ms_per_step'] for v in fam.values()),2))
except Exception as e: print('ERR', e)
P
tail -3 $O/b_fused.err

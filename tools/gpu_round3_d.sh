set -x
export TMPDIR=/tmp
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fused_block_gpu.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
B="python bench.py --no-cpu-baseline --steps 20 --warmup 5 --sustained-steps 0"
$B > $O/b_new.json 2> $O/b_new.err
CAT_STEM_TCONV=0 $B > $O/b_nostem.json 2> $O/b_nostem.err
for f in new nostem; do python - <<P
import json
try:
    d=json.load(open('$O/b_$f.json')); print('$f', d['value'], d['ms_per_step'], d['student_forward'])
    fam=d['roofline']['families']
    for k in sorted(fam):
        if k.startswith('conv_tconv') or k.startswith('conv_tstage1') or k.startswith('conv_fwd_4x2') or k.startswith('conv_fwd_4x4x4') : print('   ', k, fam[k])
except Exception as e: print('$f', 'ERR', e)
P
done

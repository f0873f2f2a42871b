set -x
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
B="python bench.py --no-cpu-baseline --size 512 --batch 16 --steps 6 --warmup 2 --sustained-steps 0"
$B > $O/b512_new.json 2> $O/b512_new.err
CAT_WGRAD_RAGGED=0 $B > $O/b512_old.json 2> $O/b512_old.err
for f in new old; do python - <<P
import json
try:
    d=json.load(open('$O/b512_$f.json')); print('$f', d['value'], d['ms_per_step'])
    fam=d['roofline']['families']
    for k in sorted(fam):
        if 'wgrad' in k: print('   ', k, fam[k])
except Exception as e: print('$f', 'ERR', e)
P
done

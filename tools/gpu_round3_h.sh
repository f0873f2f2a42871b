export TMPDIR=/tmp
O=gpurun_out/r3h; mkdir -p $O
B="python bench.py --workload spade --no-cpu-baseline --no-kernel-profile --steps 10 --warmup 3 --sustained-steps 0"
for t in 96 64 16; do
CAT_TCONV_MIN_TILES=$t $B > $O/b_$t.json 2> $O/b_$t.err
python - <<P
import json
try:
    d=json.load(open('$O/b_$t.json')); print('min_tiles $t', d['value'], d['ms_per_step'])
except Exception as e: print('$t ERR', e)
P
done

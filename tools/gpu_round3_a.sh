set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r3a
( timeout 1500 python -m pytest tests -m gpu -x -q -n 4 --dist loadfile -p no:cacheprovider > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3a/pytest.log ) 
tail -5 gpurun_out/r3a/pytest.log
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --sustained-steps 0 > gpurun_out/r3a/bench_pw1.json 2> gpurun_out/r3a/bench_pw1.err
CAT_TCONV_PW=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --sustained-steps 0 > gpurun_out/r3a/bench_pw0.json 2> gpurun_out/r3a/bench_pw0.err
python bench.py --no-cpu-baseline --no-kernel-profile --steps 20 --warmup 5 --sustained-steps 0 --dp-schedule 1 > gpurun_out/r3a/bench_dp.json 2> gpurun_out/r3a/bench_dp.err
python bench.py --no-cpu-baseline --no-kernel-profile --steps 20 --warmup 5 --sustained-steps 0 --dp-schedule 1 --graph 0 > gpurun_out/r3a/bench_dp_eager.json 2> gpurun_out/r3a/bench_dp_eager.err
for f in pw1 pw0 dp dp_eager; do python - <<P
import json
try:
    d=json.load(open('gpurun_out/r3a/bench_$f.json'))
    print('$f', d['value'], d['ms_per_step'], d['config']['launch'], (d.get('student_forward') or {}).get('ms'), (d.get('roofline') or {}).get('frac'))
    fam=(d.get('roofline') or {}).get('families') or {}
    for k in ('conv_tconv','conv_tconv_multi','conv_tstage1'):
        if k in fam: print('   ', k, fam[k])
except Exception as e: print('$f', 'ERR', e)
P
done
tail -3 gpurun_out/r3a/*.err

set -x
export TMPDIR=/tmp
O=$PWD/gpurun_out/validate; mkdir -p $O
( time timeout 1500 python -m pytest ${VALIDATE_TESTS:-tests} -m gpu -x -q -p no:cacheprovider ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --workload spade > $O/bench_spade.json 2> $O/bench_spade.err
export CAT_BRANCH_STREAMS=0
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof_spade -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile --graph 0 --sustained-steps 0 --workload spade > $O/prof_spade.log 2>&1)
python tools/rocprof_summary.py $O/prof_spade/bench_results.db $O/kernel_stats_spade.txt 6 > /dev/null
rm -rf $O/prof_spade
python - <<P
import json
d=json.load(open('$O/bench_spade.json')); fam=d['roofline']['families']
print('spade', d['value'], d['ms_per_step'], d['student_forward'], 'launches', sum(v['launches_per_step'] for v in fam.values()), d.get('cpu_baseline'))
P
grep -h "gradients\|headline parity\|\[dp\]\|weights\]\|updated weights" $O/pytest.log | cut -c1-330

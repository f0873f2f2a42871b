#!/bin/bash
# Round-5 fallback of tools/profile_round.sh (whose PMC passes no longer fit the round's GPU budget): the default bench line, the rocprofv3
# --kernel-trace --stats summary of the same (serial) C2 command and of the student forward.  Every step carries its own time limit; raw traces
# are summarised on the box and deleted (gpurun returns <= 64 MiB).   COMMIT=<head> bash tools/profile_quick.sh
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/quick
mkdir -p $OUT
python - <<PYEOF > $OUT/meta.json
import json, sys, datetime, importlib.util
sys.argv = ['x']
spec = importlib.util.spec_from_file_location('bench', 'bench.py'); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
print(json.dumps({'commit': '${COMMIT:-unknown}', 'csrc_sha': m.csrc_fingerprint(), 'date': datetime.date.today().isoformat(),
                  'command': 'COMMIT=<head> bash tools/profile_quick.sh (one MI355X, through gpurun)'}, indent=1))
PYEOF
S=$(date +%s)
timeout 330 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
echo "default bench.py: rc $? in $(( $(date +%s) - S )) s" > $OUT/timing.txt
export CAT_BRANCH_STREAMS=0
B="python $PWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary --graph 0 --sustained-steps 0"
(cd /tmp && timeout 160 rocprofv3 --kernel-trace --stats -d $OUT/prof_c2 -o bench -- $B > $OUT/prof_c2.log 2>&1)
python tools/rocprof_summary.py $OUT/prof_c2/bench_results.db $OUT/kernel_stats_c2.txt 6 > /dev/null 2>&1
rm -rf $OUT/prof_c2
(cd /tmp && REPS=5 timeout 110 rocprofv3 --kernel-trace --stats -d $OUT/prof_sfwd -o sfwd -- python $PWD/tools/debug/student_fwd_trace.py > $OUT/prof_sfwd.log 2>&1)
python tools/rocprof_summary.py $OUT/prof_sfwd/sfwd_results.db $OUT/kernel_stats_student_fwd.txt 8 > /dev/null 2>&1
rm -rf $OUT/prof_sfwd
echo "total $(( $(date +%s) - S )) s" >> $OUT/timing.txt
du -sh $OUT; cat $OUT/timing.txt; tail -c 400 $OUT/bench_c2.json

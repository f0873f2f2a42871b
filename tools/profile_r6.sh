#!/bin/bash
# Round-6 evidence pass, run ON THE GPU BOX through gpurun from the repo root; every step carries its own time limit and only summaries
# travel back (gpurun returns <= 64 MiB).            COMMIT=<head> [PMC=0] [SPADE=1] bash tools/profile_r6.sh
#   bench_c2.json              the default `python bench.py` line (roofline, cpu_baseline, secondary)
#   kernel_stats_c2.txt        rocprofv3 --kernel-trace --stats of the serial C2 command, LIBRARY kernels only (model construction's
#                              ATen fills / copies are filtered out of the table), per step
#   kernel_stats_student_fwd.txt   the same for replays of the captured student forward
#   pmc_sq_c2.txt, pmc_hbm.json    SQ issue / wait / MFMA-busy counters and FETCH_SIZE / WRITE_SIZE of 2 serial steps, collected ONLY for
#                              the conv / norm / depthwise kernels (--kernel-include-regex): three separate --pmc passes with
#                              --kernel-trace only, ~1-2 minutes each instead of ~10 for every dispatch of the process
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r6prof
mkdir -p $OUT
python - <<PYEOF > $OUT/meta.json
import json, sys, datetime, importlib.util
sys.argv = ['x']
spec = importlib.util.spec_from_file_location('bench', 'bench.py'); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
print(json.dumps({'commit': '${COMMIT:-unknown}', 'csrc_sha': m.csrc_fingerprint(), 'date': datetime.date.today().isoformat(),
                  'command': 'COMMIT=<head> bash tools/profile_r6.sh (one MI355X, through gpurun)'}, indent=1))
PYEOF
S=$(date +%s)
T() { echo "$1: $(( $(date +%s) - S )) s" >> $OUT/timing.txt; }
: > $OUT/timing.txt
if [ -z "$SKIP_BENCH" ]; then
timeout 400 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; T "default bench.py rc $?"
fi
export CAT_BRANCH_STREAMS=0
B="python $PWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary --graph 0 --sustained-steps 0"
(cd /tmp && timeout 170 rocprofv3 --kernel-trace --stats -d $OUT/prof_c2 -o bench -- $B > $OUT/prof_c2.log 2>&1); T "kernel trace c2 rc $?"
python tools/rocprof_summary.py $OUT/prof_c2/bench_results.db $OUT/kernel_stats_c2.txt 6 --library > /dev/null 2>&1
rm -rf $OUT/prof_c2
SF="python $PWD/tools/debug/net_fwd_trace.py"
(cd /tmp && NET=student REPS=5 timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/prof_sfwd -o sfwd -- $SF > $OUT/prof_sfwd.log 2>&1); T "kernel trace student fwd rc $?"
python tools/rocprof_summary.py $OUT/prof_sfwd/sfwd_results.db $OUT/kernel_stats_student_fwd.txt 8 --library > /dev/null 2>&1
rm -rf $OUT/prof_sfwd
if [ -n "$SPADE" ]; then
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_spade -o bench -- $B --workload spade > $OUT/prof_spade.log 2>&1); T "kernel trace spade rc $?"
python tools/rocprof_summary.py $OUT/prof_spade/bench_results.db $OUT/kernel_stats_spade.txt 6 --library > /dev/null 2>&1
rm -rf $OUT/prof_spade
fi
if [ "${PMC:-1}" != "0" ]; then
P="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-secondary --graph 0 --sustained-steps 0"
RX='conv_|ksum|tconv|tstage1|twgrad|pwgrad|qconv|smallco|norm_|dwm_|dw_|gram'
(cd /tmp && timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-trace --kernel-include-regex "$RX" --output-format csv -d $OUT/pmc_sq -o q -- $P > $OUT/pmc_sq.log 2>&1); T "pmc sq rc $?"
python tools/pmc_sq_summary.py $OUT/pmc_sq $OUT/pmc_sq_c2.txt > /dev/null 2>&1
rm -rf $OUT/pmc_sq
(cd /tmp && timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "$RX" --output-format csv -d $OUT/pmc/fetch -o f -- $P > $OUT/pmc_fetch.log 2>&1); T "pmc fetch rc $?"
(cd /tmp && timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "$RX" --output-format csv -d $OUT/pmc/write -o w -- $P > $OUT/pmc_write.log 2>&1); T "pmc write rc $?"
python tools/pmc_summary.py $OUT/pmc $OUT/pmc_hbm.json > /dev/null 2>&1
rm -rf $OUT/pmc
fi
T total
du -sh $OUT; cat $OUT/timing.txt; tail -c 300 $OUT/bench_c2.json 2>/dev/null

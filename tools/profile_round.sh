#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: the round's measured artefacts -> gpurun_out/round/
#   bench lines (c2 default incl. roofline + cpu_baseline, spade), rocprofv3 kernel-trace stats of the same commands,
#   PMC passes (FETCH_SIZE / WRITE_SIZE separately, kernel-trace only -- never combined with sys/hip traces).
# ROUND 5 NOTE: the whole-step PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_*) take ~10 minutes EACH on this pool; the full script ran into a
# 50-minute limit with nothing copied back.  They are therefore opt-in (WITH_PMC=1, give gpurun --timeout 4500); without it this script takes
# ~8 minutes.  tools/profile_quick.sh + tools/debug/patchgan_fwd_trace.py are the 3-minute fallback the round's evidence was taken with.
[ -z "$WITH_PMC" ] && SKIP_PMC=1
set -x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/round
mkdir -p $OUT
# which tree the tables are taken from: the commit is passed in (the GPU box has no .git), the kernel-source fingerprint is computed here --
# bench.py compares it with the running sources and reports roofline.profiles_stale
python - <<PYEOF > $OUT/meta.json
import json, sys, datetime
sys.argv = ['x']
import importlib.util
spec = importlib.util.spec_from_file_location('bench', 'bench.py'); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
print(json.dumps({'commit': '${COMMIT:-unknown}', 'csrc_sha': m.csrc_fingerprint(), 'date': datetime.date.today().isoformat(),
                  'command': 'COMMIT=<head> bash tools/profile_round.sh (one MI355X, through gpurun)'}, indent=1))
PYEOF
if [ -z "$SKIP_BENCH" ]; then
python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python bench.py --workload spade > $OUT/bench_spade.json 2> $OUT/bench_spade.err
python bench.py --size 512 --batch 16 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_c2_512.json 2> $OUT/bench_c2_512.err
python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1
python tools/qconv_bench.py > $OUT/qconv_layers.txt 2>&1
fi
# per-kernel durations are compared with bench.py's SERIAL roofline pass: branch streams off, no in-process event profiling
export CAT_BRANCH_STREAMS=0
B="python $PWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile --graph 0 --sustained-steps 0"
SF="python $PWD/tools/debug/student_fwd_trace.py"
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_c2 -o bench -- $B > $OUT/prof_c2.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_spade -o bench -- $B --workload spade > $OUT/prof_spade.log 2>&1)
if [ -z "$SKIP_PMC" ]; then
P="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --graph 0 --sustained-steps 0"
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc/fetch -o f -- $P > $OUT/pmc_fetch.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc/write -o w -- $P > $OUT/pmc_write.log 2>&1)
fi
# summarise on the box; only the summaries travel back (gpurun returns <= 64 MiB)
python tools/rocprof_summary.py $OUT/prof_c2/bench_results.db $OUT/kernel_stats_c2.txt 6 > /dev/null
# the student forward alone (captured graph replays) and the SQ counters of the serial C2 command
(cd /tmp && REPS=5 rocprofv3 --kernel-trace --stats -d $OUT/prof_sfwd -o sfwd -- $SF > $OUT/prof_sfwd.log 2>&1)
python tools/rocprof_summary.py $OUT/prof_sfwd/sfwd_results.db $OUT/kernel_stats_student_fwd.txt 8 > /dev/null
if [ -z "$SKIP_PMC" ]; then
(cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq -o q -- $P > $OUT/pmc_sq.log 2>&1)
python tools/pmc_sq_summary.py $OUT/pmc_sq $OUT/pmc_sq_c2.txt > /dev/null 2>&1
rm -rf $OUT/pmc_sq
fi
rm -rf $OUT/prof_sfwd
python tools/rocprof_summary.py $OUT/prof_spade/bench_results.db $OUT/kernel_stats_spade.txt 6 > /dev/null
[ -d $OUT/pmc ] && python tools/pmc_summary.py $OUT/pmc $OUT/pmc_hbm.json
rm -rf $OUT/prof_c2 $OUT/prof_spade $OUT/pmc
du -sh $OUT
tail -c 600 $OUT/bench_c2.json; tail -c 300 $OUT/bench_spade.err

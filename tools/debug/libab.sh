# A/B two library builds on one box through the serial kernel trace: usage  bash tools/debug/libab.sh <regex of kernels to sum> libA.so libB.so ...
R=$PWD; RX=$1; shift; cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/pn
  CAT_LIB=$R/$lib timeout 170 rocprofv3 --kernel-trace --stats -d /tmp/pn -o b -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-secondary --graph 0 --sustained-steps 0 > /tmp/log 2>&1
  DB=$(find /tmp/pn -name "*.db" | head -1); python $R/tools/rocprof_summary.py $DB /tmp/k.txt 6 --library > /dev/null
  echo "== $lib: $(grep -E "$RX" /tmp/k.txt | awk '{s+=$NF} END {print s}') us/step"; grep -E "$RX" /tmp/k.txt | cut -c1-50,79-140
done

run() { python bench.py "$@" --no-cpu-baseline --no-kernel-profile --sustained-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for w in 4096 2048 1024 256; do
  echo "== MINWG=$w spade"; CAT_PK_TW32_MINWG=$w run --workload spade
  echo "== MINWG=$w c2"; CAT_PK_TW32_MINWG=$w run
done

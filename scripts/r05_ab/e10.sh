#!/bin/bash
# round-5 A/B batch 10: stride-2 gather with the weight fragments shared through LDS (default) vs per-wave streams (BTS_LP_GATHERW=0)
cd "$GRAFT_REPO_ROOT"
one() { python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms'%d['ms_per_step'])"; }
for r in 1 2 3; do
  echo "bf16  gatherw $(one --dtype bf16 --batch 8 --steps 10 --warmup 3)   streams $(BTS_LP_GATHERW=0 one --dtype bf16 --batch 8 --steps 10 --warmup 3)"
  echo "infer gatherw $(one --infer --dtype f16 --steps 30 --warmup 10)   streams $(BTS_LP_GATHERW=0 one --infer --dtype f16 --steps 30 --warmup 10)"
done

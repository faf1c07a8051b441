#!/bin/bash
# round-5 A/B batch 3: split-K threshold of lp_s1d (whole-layer split for the 360-item 40x48x40 level of the inference volume)
cd "$GRAFT_REPO_ROOT"
one() { python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms'%d['ms_per_step'])"; }
for r in 1 2 3; do
  echo "infer default $(one --infer --dtype f16 --steps 30 --warmup 10)   SPLIT_ITEMS=400 $(BTS_LP_S1D_SPLIT_ITEMS=400 one --infer --dtype f16 --steps 30 --warmup 10)   SPLIT_ITEMS=1300 $(BTS_LP_S1D_SPLIT_ITEMS=1300 one --infer --dtype f16 --steps 30 --warmup 10)"
done

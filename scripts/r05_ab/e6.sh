#!/bin/bash
# round-5 A/B batch 6: four position groups per wave in the stride-2 gather kernel
cd "$GRAFT_REPO_ROOT"
one() { python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms'%d['ms_per_step'])"; }
for r in 1 2 3; do
  echo "bf16  default $(one --dtype bf16 --batch 8 --steps 10 --warmup 3)   VB4=0 $(BTS_LP_GATHERQ_VB4=0 one --dtype bf16 --batch 8 --steps 10 --warmup 3)   VB4>=128 $(BTS_LP_GATHERQ_VB4=128 one --dtype bf16 --batch 8 --steps 10 --warmup 3)"
  echo "infer default $(one --infer --dtype f16 --steps 30 --warmup 10)   VB4=0 $(BTS_LP_GATHERQ_VB4=0 one --infer --dtype f16 --steps 30 --warmup 10)   VB4>=64 $(BTS_LP_GATHERQ_VB4=64 one --infer --dtype f16 --steps 30 --warmup 10)"
done

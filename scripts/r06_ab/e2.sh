#!/bin/bash
# round-6 A/B batch 2: level-0 concat gradient as two dense 32-channel tensors (default) vs one 64-wide slab (BTS_LP_SC_SPLIT=0)
cd "$GRAFT_REPO_ROOT"
one() { python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms'%d['ms_per_step'])"; }
for r in 1 2 3; do
  echo "bf16  split $(one --dtype bf16 --batch 8 --steps 10 --warmup 3)   slab $(BTS_LP_SC_SPLIT=0 one --dtype bf16 --batch 8 --steps 10 --warmup 3)"
done

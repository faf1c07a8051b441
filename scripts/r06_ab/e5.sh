#!/bin/bash
# round-6 A/B batch 5: conv1's + the shortcut's weight gradients from one pass over the block input (default) vs two launches (BTS_LP_WPAIR=0)
cd "$GRAFT_REPO_ROOT"
one() { python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms'%d['ms_per_step'])"; }
for r in 1 2 3; do
  echo "bf16  fused $(one --dtype bf16 --batch 8 --steps 10 --warmup 3)   two-launch $(BTS_LP_WPAIR=0 one --dtype bf16 --batch 8 --steps 10 --warmup 3)"
done

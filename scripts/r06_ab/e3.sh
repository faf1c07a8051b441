#!/bin/bash
# round-6 A/B batch 3: conv1 + shortcut + squeeze from one pass over the block input (default) vs the 1x1x1 launch of its own (BTS_LP_FS=0)
cd "$GRAFT_REPO_ROOT"
one() { python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms'%d['ms_per_step'])"; }
for r in 1 2 3; do
  echo "bf16  fused $(one --dtype bf16 --batch 8 --steps 10 --warmup 3)   two-launch $(BTS_LP_FS=0 one --dtype bf16 --batch 8 --steps 10 --warmup 3)"
done

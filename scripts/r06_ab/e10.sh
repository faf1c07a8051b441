#!/bin/bash
# bf16 batch-8 step: the fused block backward's small launches merged (middle: GroupNorm-2 finalize + gate partial sums; tail: SE-MLP parameter
# gradients + the two bias-gradient finalizes; BTS_LP_BLK_BWD_MERGE=1, default) vs the six separate launches (=0); three alternating rounds in one box
cd "$GRAFT_REPO_ROOT"
one() { python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms'%d['ms_per_step'])"; }
for r in 1 2 3; do
  echo "bf16 b8: merged $(BTS_LP_BLK_BWD_MERGE=1 one --dtype bf16 --batch 8 --steps 10 --warmup 3)   separate $(BTS_LP_BLK_BWD_MERGE=0 one --dtype bf16 --batch 8 --steps 10 --warmup 3)"
done

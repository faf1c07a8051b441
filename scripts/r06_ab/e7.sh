#!/bin/bash
# round-6 A/B batch 7: the decoder top block's shortcut (64 -> 32 at 8 x 128^3) riding on conv1's two z-marching passes (default) vs the forward
# shortcut only where conv1 is ONE launch.  There is no switch for just that: BTS_LP_FS=0 turns every forward fusion off, so the comparison is
# against the e3 result (fused at the 16 / 32-channel layers only) measured in the same call by BTS_LP_FS_PAIR=0
cd "$GRAFT_REPO_ROOT"
one() { python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms'%d['ms_per_step'])"; }
for r in 1 2 3; do
  echo "bf16  two-pass-FS $(one --dtype bf16 --batch 8 --steps 10 --warmup 3)   one-launch layers only $(BTS_LP_FS_PAIR=0 one --dtype bf16 --batch 8 --steps 10 --warmup 3)"
done

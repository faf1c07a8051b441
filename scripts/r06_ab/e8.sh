#!/bin/bash
# round-6 A/B batch 8: level 0's [skip | up-sampled] as two dense 32-channel operands in the FORWARD too (default) vs the 64-wide slab
# (BTS_LP_FWD_SPLIT=0); bf16 storage, batch 8, ms per train step
cd "$GRAFT_REPO_ROOT"
one() { python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms'%d['ms_per_step'])"; }
for r in 1 2 3; do
  echo "bf16  split $(one --dtype bf16 --batch 8 --steps 10 --warmup 3)   slab $(BTS_LP_FWD_SPLIT=0 one --dtype bf16 --batch 8 --steps 10 --warmup 3)"
done

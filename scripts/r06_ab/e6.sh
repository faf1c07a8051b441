#!/bin/bash
# round-6 A/B batch 6: decoder blocks' skip part of conv1 + shortcut on a side stream as soon as the encoder level is done (default) vs one conv
# over the whole concat (BTS_LP_EARLY_SKIP=0); fp16 storage, 160x192x160 forward, ms per volume
cd "$GRAFT_REPO_ROOT"
one() { python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms'%d['ms_per_step'])"; }
for r in 1 2 3; do
  echo "infer early-skip $(BTS_LP_EARLY_SKIP=1 one --infer --dtype f16 --steps 30 --warmup 10)   whole-concat $(BTS_LP_EARLY_SKIP=0 one --infer --dtype f16 --steps 30 --warmup 10)"
done

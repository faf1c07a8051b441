#!/bin/bash
# round-5 A/B batch 4: output head folded into the top decoder block's epilogue (fp16 inference)
cd "$GRAFT_REPO_ROOT"
one() { python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms'%d['ms_per_step'])"; }
for r in 1 2 3; do
  echo "infer default $(one --infer --dtype f16 --steps 30 --warmup 10)   FUSE_HEAD=0 $(BTS_LP_FUSE_HEAD=0 one --infer --dtype f16 --steps 30 --warmup 10)"
done

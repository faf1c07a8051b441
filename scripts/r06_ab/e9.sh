#!/bin/bash
# inference: level 0 as two dense operands (BTS_LP_INF_SPLIT=1, default) vs the 64-channel slab (=0); three alternating rounds in one box
mkdir -p gpurun_out/r06
for r in 1 2 3; do
  for v in 1 0; do
    echo "== round $r BTS_LP_INF_SPLIT=$v"
    BTS_LP_INF_SPLIT=$v python3 bench.py --infer --dtype f16 --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done

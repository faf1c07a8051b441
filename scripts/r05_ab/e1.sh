#!/bin/bash
# round-5 A/B batch 1: used-form packing (fp32), K1-first order + two-pass s1z for 64 -> 32 (bf16 / fp16 inference)
cd "$GRAFT_REPO_ROOT"
one() { python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms'%d['ms_per_step'])"; }
for r in 1 2 3; do
  echo "f32   default $(one --steps 10 --warmup 3)   PACK_USED=0 $(BTS_PACK_USED=0 one --steps 10 --warmup 3)"
  echo "bf16  default $(one --dtype bf16 --batch 8 --steps 10 --warmup 3)   K1_FIRST=0 $(BTS_LP_K1_FIRST=0 one --dtype bf16 --batch 8 --steps 10 --warmup 3)   S1Z_PAIR=0 $(BTS_LP_S1Z_PAIR=0 one --dtype bf16 --batch 8 --steps 10 --warmup 3)   both=0 $(BTS_LP_S1Z_PAIR=0 BTS_LP_K1_FIRST=0 one --dtype bf16 --batch 8 --steps 10 --warmup 3)"
  echo "infer default $(one --infer --dtype f16 --steps 30 --warmup 10)   S1Z_PAIR=0 $(BTS_LP_S1Z_PAIR=0 one --infer --dtype f16 --steps 30 --warmup 10)"
done

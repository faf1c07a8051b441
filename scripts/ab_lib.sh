#!/bin/bash
# alternate two builds of libbts_hip.so inside one gpurun call: bash scripts/ab_lib.sh <other.so> [rounds]
OTHER=${1:?path of the other build}; ROUNDS=${2:-3}
one() { python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms'%d['ms_per_step'])"; }
for r in $(seq $ROUNDS); do
  echo "bf16  default $(one --dtype bf16 --batch 8 --steps 10 --warmup 3)   other $(BTS_HIP_LIB=$OTHER one --dtype bf16 --batch 8 --steps 10 --warmup 3)"
  echo "infer default $(one --infer --dtype f16 --steps 30 --warmup 5)   other $(BTS_HIP_LIB=$OTHER one --infer --dtype f16 --steps 30 --warmup 5)"
done

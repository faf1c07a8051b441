#!/bin/bash
# round-5 A/B batch 7: cache policy of lp_s1d's output stores (nt / sc0+nt / nt+sc1 builds against the default)
cd "$GRAFT_REPO_ROOT"
B=$GRAFT_REPO_ROOT/3d-brain-tumor-segmentation_amd/csrc/build
one() { python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms'%d['ms_per_step'])"; }
for r in 1 2; do
  echo "bf16  default $(one --dtype bf16 --batch 8 --steps 10 --warmup 3)   nt $(BTS_HIP_LIB=$B/libbts_hip_st2.so one --dtype bf16 --batch 8 --steps 10 --warmup 3)   sc0+nt $(BTS_HIP_LIB=$B/libbts_hip_st3.so one --dtype bf16 --batch 8 --steps 10 --warmup 3)   nt+sc1 $(BTS_HIP_LIB=$B/libbts_hip_st18.so one --dtype bf16 --batch 8 --steps 10 --warmup 3)"
  echo "infer default $(one --infer --dtype f16 --steps 30 --warmup 10)   nt $(BTS_HIP_LIB=$B/libbts_hip_st2.so one --infer --dtype f16 --steps 30 --warmup 10)   sc0+nt $(BTS_HIP_LIB=$B/libbts_hip_st3.so one --infer --dtype f16 --steps 30 --warmup 10)   nt+sc1 $(BTS_HIP_LIB=$B/libbts_hip_st18.so one --infer --dtype f16 --steps 30 --warmup 10)"
done

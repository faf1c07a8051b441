#!/bin/bash
# round-5 A/B batch 9: DPP lane-group sums in the 16-bit element-wise kernels (default) vs __shfl_xor (build/libbts_hip_gshfl.so)
cd "$GRAFT_REPO_ROOT"
OTHER=$GRAFT_REPO_ROOT/3d-brain-tumor-segmentation_amd/csrc/build/libbts_hip_gshfl.so
one() { python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms'%d['ms_per_step'])"; }
for r in 1 2 3; do
  echo "bf16  dpp $(one --dtype bf16 --batch 8 --steps 10 --warmup 3)   shfl $(BTS_HIP_LIB=$OTHER one --dtype bf16 --batch 8 --steps 10 --warmup 3)"
  echo "infer dpp $(one --infer --dtype f16 --steps 30 --warmup 10)   shfl $(BTS_HIP_LIB=$OTHER one --infer --dtype f16 --steps 30 --warmup 10)"
done

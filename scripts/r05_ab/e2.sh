#!/bin/bash
# round-5 A/B batch 2: DPP wave sums (default build) vs the ds_bpermute form (build/libbts_hip_shfl.so)
cd "$GRAFT_REPO_ROOT"
./scripts/ab/dpp_test
OTHER=$GRAFT_REPO_ROOT/3d-brain-tumor-segmentation_amd/csrc/build/libbts_hip_shfl.so
one() { python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms'%d['ms_per_step'])"; }
for r in 1 2 3; do
  echo "f32   dpp $(one --steps 10 --warmup 3)   shfl $(BTS_HIP_LIB=$OTHER one --steps 10 --warmup 3)"
  echo "bf16  dpp $(one --dtype bf16 --batch 8 --steps 10 --warmup 3)   shfl $(BTS_HIP_LIB=$OTHER one --dtype bf16 --batch 8 --steps 10 --warmup 3)"
  echo "infer dpp $(one --infer --dtype f16 --steps 30 --warmup 10)   shfl $(BTS_HIP_LIB=$OTHER one --infer --dtype f16 --steps 30 --warmup 10)"
done

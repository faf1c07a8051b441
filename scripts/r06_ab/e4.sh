#!/bin/bash
# round-6 A/B batch 4: the shortcut's centre-tap operand pair requested in stage 1 (default) vs in stage 2 (build/libbts_hip_sclate.so)
cd "$GRAFT_REPO_ROOT"
OTHER=$GRAFT_REPO_ROOT/3d-brain-tumor-segmentation_amd/csrc/build/libbts_hip_sclate.so
one() { python bench.py "$@" --no-cpu-baseline --no-also --no-profile --allow-overrides 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms'%d['ms_per_step'])"; }
for r in 1 2 3; do
  echo "bf16  early $(one --dtype bf16 --batch 8 --steps 10 --warmup 3)   late $(BTS_HIP_LIB=$OTHER one --dtype bf16 --batch 8 --steps 10 --warmup 3)"
done

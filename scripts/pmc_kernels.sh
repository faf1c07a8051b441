#!/bin/bash
# SQ / GRBM counters of single conv kernels on one layer shape (run ON the GPU box):
#   bash scripts/pmc_kernels.sh <tag> "<one_conv.py args>" ["<more args>" ...]
#   e.g. bash scripts/pmc_kernels.sh r02 "fwd 1 128 32 32 5" "wgrad 1 128 32 32 5"
# -> gpurun_out/<tag>_pmc_kernels.txt: per kernel and counter the per-launch mean, plus derived ratios.
# Counters go in groups of <= 8 SQ + 2 GRBM per pass (the SQ block has 8 slots), each pass its own process and output
# directory, kernel-trace only (never combined with the tracing domains gpurun refuses), program directly after `--`.
set -euo pipefail
TAG=${1:?usage: pmc_kernels.sh <tag> "<one_conv args>" ...}
shift
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
GROUPS_=(
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
  "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR"
)
: > "$O/${TAG}_pmc_kernels.txt"
for spec in "$@"; do
  stag=$(echo "$spec" | tr ' ' '_')
  gi=0
  for grp in "${GROUPS_[@]}"; do
    d="$O/pmck_${TAG}_${stag}_$gi"
    rm -rf "$d"
    # shellcheck disable=SC2086
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$d" -o p -- python3 "$R/scripts/one_conv.py" $spec > "$d.log" 2>&1 \
      || { echo "pass $gi of '$spec' failed"; tail -20 "$d.log"; exit 1; }
    test -n "$(find "$d" -name p_counter_collection.csv -size +0)" || { echo "no counters from pass $gi of '$spec'"; tail -20 "$d.log"; exit 1; }
    gi=$((gi + 1))
  done
  python3 "$R/scripts/pmc_summary.py" "$O" "$TAG" "$stag" "$spec" >> "$O/${TAG}_pmc_kernels.txt"
done
cat "$O/${TAG}_pmc_kernels.txt"

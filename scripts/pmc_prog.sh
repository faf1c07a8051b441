#!/bin/bash
# SQ / GRBM counters of every kernel of an arbitrary python program (run ON the GPU box):
#   bash scripts/pmc_prog.sh <tag> <kernel-name substring> <script.py> [args...]   -> gpurun_out/<tag>_pmc_prog.txt
# Same rules as pmc_kernels.sh: counter groups in separate passes, kernel-trace only, python3 directly after `--`.
set -euo pipefail
TAG=${1:?tag}; FILTER=${2:?kernel filter}; shift 2
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
GROUPS_=(
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
  "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
  "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"
)
gi=0
for grp in "${GROUPS_[@]}"; do
  d="$O/pmcp_${TAG}_$gi"; rm -rf "$d"
  # shellcheck disable=SC2086
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$d" -o p -- python3 "$@" > "$d.log" 2>&1 || { echo "pass $gi failed"; tail -20 "$d.log"; exit 1; }
  test -n "$(find "$d" -name p_counter_collection.csv -size +0)" || { echo "no counters from pass $gi"; tail -20 "$d.log"; exit 1; }
  gi=$((gi + 1))
done
python3 - "$O" "$TAG" "$FILTER" > "$O/${TAG}_pmc_prog.txt" <<'PY'
import collections, csv, glob, sys
O, TAG, FILTER = sys.argv[1:4]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('%s/pmcp_%s_*/**/p_counter_collection.csv' % (O, TAG), recursive=True)):
    for r in csv.DictReader(open(f)):
        vals[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in sorted(vals.items()):
    if FILTER not in k:
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    print('-- %s  (%d launches)' % (k, max(len(v) for v in cs.values())))
    for c in sorted(m):
        print('   %-32s %16.1f' % (c, m[c]))
    g = lambda c: (m.get(c) or float('nan'))
    print('   > launch cycles (GUI_ACTIVE/8)        %.0f' % (g('GRBM_GUI_ACTIVE') / 8))
    print('   > MFMA busy / (launch cycles * 1024)  %.3f' % (g('SQ_VALU_MFMA_BUSY_CYCLES') / (g('GRBM_GUI_ACTIVE') / 8 * 1024)))
    print('   > wave cycles: wait %.3f issue-stall %.3f active %.3f' % (g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'),
          g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'), g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES')))
    print('   > LDS bank-conflict cycles / LDS active cycles  %.3f' % (g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE')))
    print('   > waves per launch %.0f ; wave cycles per wave %.0f' % (g('SQ_WAVES'), g('SQ_WAVE_CYCLES') / g('SQ_WAVES')))
PY
cat "$O/${TAG}_pmc_prog.txt"

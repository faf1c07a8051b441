#!/bin/bash
# Memory-side counters of one kernel of an arbitrary python program (run ON the GPU box), one rocprofv3 pass per group:
#   bash scripts/pmc_mem.sh <tag> <kernel-name substring> <script.py> [args...]   -> gpurun_out/<tag>_pmc_mem.txt
set -uo pipefail
TAG=${1:?tag}; FILTER=${2:?kernel filter}; shift 2
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
GROUPS_=(
  "FETCH_SIZE GRBM_GUI_ACTIVE"
  "WRITE_SIZE GRBM_GUI_ACTIVE"
  "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum"
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
)
gi=0
for grp in "${GROUPS_[@]}"; do
  d="$O/pmcm_${TAG}_$gi"; rm -rf "$d"
  # shellcheck disable=SC2086
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$d" -o p -- python3 "$@" > "$d.log" 2>&1 || { echo "pass $gi ($grp) failed"; tail -5 "$d.log"; }
  gi=$((gi + 1))
done
python3 - "$O" "$TAG" "$FILTER" > "$O/${TAG}_pmc_mem.txt" <<'PY'
import collections, csv, glob, sys
O, TAG, FILTER = sys.argv[1:4]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('%s/pmcm_%s_*/**/p_counter_collection.csv' % (O, TAG), recursive=True)):
    for r in csv.DictReader(open(f)):
        vals[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in sorted(vals.items()):
    if FILTER not in k:
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    print('-- %s  (%d launches)' % (k, max(len(v) for v in cs.values())))
    for c in sorted(m):
        print('   %-32s %16.1f' % (c, m[c]))
PY
cat "$O/${TAG}_pmc_mem.txt"

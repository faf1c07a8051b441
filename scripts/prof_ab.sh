#!/bin/bash
# rocprofv3 kernel stats of the bf16 batch-8 step with and without one switch (run ON the GPU box): bash scripts/prof_ab.sh VAR=VALUE
set -uo pipefail
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
O=$R/gpurun_out
SW=${1:?VAR=VALUE}
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/pab_a" "$O/pab_b"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/pab_a" -o s -- python3 "$R/bench.py" --dtype bf16 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-also --serial-streams > "$O/pab_a.log" 2>&1
export "$SW"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/pab_b" -o s -- python3 "$R/bench.py" --dtype bf16 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-also --serial-streams --allow-overrides > "$O/pab_b.log" 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
tabs = []
for d in ('pab_a', 'pab_b'):
    f = glob.glob('%s/%s/**/s_kernel_stats.csv' % (O, d), recursive=True)[0]
    tabs.append({r['Name'].split('(')[0]: (int(r['Calls']), float(r['TotalDurationNs']) / 4e6) for r in csv.DictReader(open(f))})
a, b = tabs
print('total %.2f -> %.2f ms/step' % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
for k in sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, (0, 0))[1] - b.get(k, (0, 0))[1]))[:14]:
    x, y = a.get(k, (0, 0)), b.get(k, (0, 0))
    print('%-70s %4d x -> %4d x   %7.3f -> %7.3f ms  (%+.3f)' % (k[:70], x[0] // 4, y[0] // 4, x[1], y[1], y[1] - x[1]))
PY

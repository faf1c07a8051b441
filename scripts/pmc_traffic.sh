# HBM traffic of every kernel of the bench step from the L2's memory-side counters (MI355X guide, HBM section):
# FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots) -> two separate --pmc runs, kernel-trace only.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_traffic
mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O -o fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O -o write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
ls -la $O
python3 - <<PY
import csv, collections, json
out = {}
for tag, cname in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open('$O/%s_counter_collection.csv' % tag)):
        if r['Counter_Name'] == cname:
            agg[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    for k, v in agg.items():
        out.setdefault(k, {})[cname + '_KiB_mean'] = sum(v) / len(v)
        out[k]['launches'] = len(v)
json.dump(out, open('$O/traffic.json', 'w'), indent=1, sort_keys=True)
for k in sorted(out, key=lambda k: -out[k].get('FETCH_SIZE_KiB_mean', 0) * out[k]['launches'])[:12]:
    print(k[:60], out[k])
PY

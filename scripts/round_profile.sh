# Round-end evidence: kernel stats of the default bench, the bench JSON line, and the two HBM-traffic PMC passes.
# Usage on the GPU box: bash scripts/round_profile.sh r01c   -> gpurun_out/<tag>_*
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 10 --warmup 3 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
tail -c 600 $O/${TAG}_bench.json
rm -rf $O/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile > $O/${TAG}_bench_prof.log 2>&1
cp $(find $O/prof_$TAG -name bench_kernel_stats.csv | head -1) $O/${TAG}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_$TAG -o fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_$TAG -o write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
python3 - <<PY
import csv, collections, json, glob
out = {}
for tag, cname in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    agg = collections.defaultdict(list)
    f = glob.glob('$O/pmc_$TAG/**/%s_counter_collection.csv' % tag, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == cname:
            agg[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    for k, v in agg.items():
        out.setdefault(k, {})[cname + '_KiB_mean'] = sum(v) / len(v)
        out[k]['launches'] = len(v)
json.dump(out, open('$O/${TAG}_pmc_traffic.json', 'w'), indent=1, sort_keys=True)
for k in sorted(out, key=lambda k: -out[k].get('FETCH_SIZE_KiB_mean', 0) * out[k]['launches'])[:8]:
    print(k[:70], out[k])
PY
head -12 $O/${TAG}_bench_kernel_stats.csv | cut -c1-150

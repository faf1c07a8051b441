# rocprofv3 kernel stats of the default bench command -> gpurun_out/prof_bench/bench_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log | cut -c1-200
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_bench/**/bench_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:40]:
    print('%-72s %6s calls %9.1f us avg %7.2f ms/step %5.1f%%' % (r['Name'][:72], r['Calls'], float(r['AverageNs']) / 1e3,
          float(r['TotalDurationNs']) / 1e6 / 7, 100 * float(r['TotalDurationNs']) / tot))
PY

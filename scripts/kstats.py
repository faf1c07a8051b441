"""top kernels of a rocprofv3 --kernel-trace --stats run: python scripts/kstats.py <dir> [per_units] [top]"""
import csv, glob, sys
d = sys.argv[1]
per = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
f = sorted(glob.glob(d + '/**/*kernel_stats.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total %.3f ms per unit' % (tot / 1e6 / per))
for r in rows[:top]:
    print('%-90s calls %6s avg %9.1f us  total %8.3f ms/unit  %5.1f %%' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3,
                                                                         float(r['TotalDurationNs']) / 1e6 / per, float(r['Percentage'])))

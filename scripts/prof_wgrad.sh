# per-kernel durations of one weight-gradient call: scripts/prof_wgrad.sh D Cin Cout [env...]
cd /tmp && export TMPDIR=/tmp
D=$1; CI=$2; CO=$3
rm -rf /tmp/pw && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw -o pw -- python3 $GRAFT_REPO_ROOT/scripts/one_conv.py wgrad 1 $D $CI $CO 10 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/pw/**/pw_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'wgrad' in n or 'memset' in n.lower() or 'fill' in n.lower():
        print('%-70s calls %4s avg %9.1f us' % (n[:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY

"""last step of a rocprofv3 --kernel-trace capture as a per-launch table (launch order): python scripts/trace_list.py <dir>"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
n = len(names)
period = None
for tail in range(0, 65):           # launches after the last step (result read-outs) do not belong to it
    m = n - tail
    for p in range(8, m // 2):
        if names[m - p:m] == names[m - 2 * p:m - p]:
            period = p
            break
    if period:
        break
assert period, 'no repeating step found in %d dispatches' % n
last = rows[m - period:m]
short = lambda s: re.sub(r'\(.*', '', s.replace('void ', '')).replace('(anonymous namespace)::', '')[:70]
t0 = int(last[0]['Start_Timestamp'])
tot = 0.0
agg = {}
print('# %d launches per step; columns: index, start (us from the first launch), duration (us), gap to the previous end (us), grid, block, LDS, kernel' % period)
prev_end = t0
for i, r in enumerate(last):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    d = (e - s) / 1e3
    tot += d
    k = short(r['Kernel_Name'])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += d
    grid = 'x'.join(r.get(c, '?') for c in ('Grid_Size_X', 'Grid_Size_Y', 'Grid_Size_Z'))
    blk = r.get('Workgroup_Size_X', '?')
    print('%4d %9.1f %8.1f %6.1f  %-16s %4s %6s  %s' % (i, (s - t0) / 1e3, d, (s - prev_end) / 1e3, grid, blk, r.get('LDS_Block_Size', '?'), k))
    prev_end = e
print('# ---- by kernel ----')
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('# %-70s %4d x %8.1f us = %8.3f ms' % (k, c, d / c, d / 1e3))
print('# step: %.3f ms of kernels, %.3f ms from first start to last end' % (tot / 1e3, (int(last[-1]['End_Timestamp']) - t0) / 1e6))

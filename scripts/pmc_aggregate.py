#!/usr/bin/env python3
"""Aggregate the rocprofv3 --pmc passes of scripts/round_profile.sh into profiles-style tables (one JSON per configuration).

  python3 scripts/pmc_aggregate.py <dir with pmc_<TAG><sfx>_<pass>/> <TAG>

Per kernel name: the mean of every counter over all its dispatches in the capture + `launches` (as before), and -- where the capture
holds whole train steps -- a `steady` object: the same means and the launch count over the LAST FULL STEP only.  A step ends with its
Adam dispatch, so the last full step is what lies between the last two Adam dispatches.  The first step of a process also runs what
happens once (every weight image packed at first use, one launch each; later steps re-pack them in one batched launch), and dividing
the capture's totals by its step count charges that to every step: 118 + 76 pack launches = 0.6 GB of the bf16 step's 251.7 GB in
profiles/r06c.  bench.py's step totals use `steady` where present; per-kernel figures keep the all-dispatch means."""
import collections
import csv
import glob
import json
import sys

O, TAG = sys.argv[1], sys.argv[2]
CMD = {'': '', '_bf16_b8': ' --dtype bf16 --batch 8', '_infer_f16': ' --infer --dtype f16'}


def kname(r):
    return r['Kernel_Name'].split('(')[0]


def collect(rows, name):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        ctr = r['Counter_Name']
        if name == 'sq' and ctr == 'GRBM_GUI_ACTIVE':
            ctr = 'GRBM_GUI_ACTIVE_sqpass'                # (the launch cycles of the pass the SQ counters were taken in)
        agg[kname(r)][ctr].append(float(r['Counter_Value']))
        if ctr == 'GRBM_GUI_ACTIVE':                      # duration of the same dispatch, for the clock
            agg[kname(r)]['duration_ns'].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    return agg


def fold(agg, out):
    for k, cs in agg.items():
        for ctr, v in cs.items():
            key = ctr + ('_KiB' if ctr in ('FETCH_SIZE', 'WRITE_SIZE') else '')
            out.setdefault(k, {})[key + '_mean'] = sum(v) / len(v)
            out[k]['launches'] = len(v)


for sfx in ('', '_bf16_b8', '_infer_f16'):
    out, steady, steady_n = {}, {}, None
    for name in ('fetch', 'write', 'req', 'sq'):
        files = glob.glob('%s/pmc_%s%s_%s/**/%s_counter_collection.csv' % (O, TAG, sfx, name, name), recursive=True)
        if not files and name == 'sq':
            continue
        assert len(files) == 1, files
        rows = list(csv.DictReader(open(files[0])))
        assert rows, 'no rows in %s' % files[0]
        fold(collect(rows, name), out)
        # the last full step: dispatches after the second-to-last Adam dispatch up to and including the last one
        disp = sorted({int(r['Dispatch_Id']): kname(r) for r in rows}.items())
        marks = [d for d, k in disp if 'adam' in k.lower()]
        if len(marks) >= 2:
            lo, hi = marks[-2], marks[-1]
            fold(collect([r for r in rows if lo < int(r['Dispatch_Id']) <= hi], name), steady)
            n = sum(1 for d, _ in disp if lo < d <= hi)
            assert steady_n in (None, n), (sfx, name, steady_n, n)          # every pass replays the same step
            steady_n = n
    for tab in (out, steady):        # matrix-pipe busy fraction per kernel (0 for kernels without matrix instructions)
        for k, v in tab.items():
            if 'SQ_VALU_MFMA_BUSY_CYCLES_mean' in v and v.get('GRBM_GUI_ACTIVE_sqpass_mean'):
                v['mfma_busy'] = v['SQ_VALU_MFMA_BUSY_CYCLES_mean'] / (v['GRBM_GUI_ACTIVE_sqpass_mean'] / 8.0 * 1024.0)
    for k, v in steady.items():
        out[k]['steady'] = v
    # 3 steps per capture (1 warm-up + 2 timed; --no-profile: no further ones); one-time construction kernels are in there too
    out['_meta'] = {'steps_in_capture': 3, 'command': 'bench.py%s --steps 2 --warmup 1 --serial-streams under rocprofv3 --kernel-trace --pmc <group>' % CMD[sfx]}
    if steady_n:
        out['_meta']['steady_step_launches'] = steady_n
        out['_meta']['steady_step'] = 'the dispatches between the last two Adam dispatches of the capture (the second timed step)'
    json.dump(out, open('%s/%s_pmc_traffic%s.json' % (O, TAG, sfx), 'w'), indent=1, sort_keys=True)
    print('%s_pmc_traffic%s.json: %d kernels%s' % (TAG, sfx, len(out) - 1, ', steady step of %d launches' % steady_n if steady_n else ''))

#!/usr/bin/env python3
"""Summarise the counter CSVs of scripts/pmc_kernels.sh: pmc_summary.py <gpurun_out> <tag> <spec tag> "<one_conv args>"
Per kernel and counter the per-launch mean (rocprofv3 sums SQ counters over the chip's shader engines / CUs and GRBM_GUI_ACTIVE
over the 8 XCDs), then derived ratios that only divide like by like."""
import collections
import csv
import glob
import sys

O, TAG, stag, spec = sys.argv[1:5]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('%s/pmck_%s_%s_*/**/p_counter_collection.csv' % (O, TAG, stag), recursive=True)):
    for r in csv.DictReader(open(f)):
        vals[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
print('== one_conv.py %s' % spec)
for k, cs in sorted(vals.items()):
    if not any(s in k for s in ('w3_', 'wino', 'wgw', 'wgrad_kernel', 'igemm', 'k1', 'upm', 'dsc', 'c2_')):
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    print('-- %s  (%d launches)' % (k, max(len(v) for v in cs.values())))
    for c in sorted(m):
        print('   %-32s %16.1f' % (c, m[c]))
    g = lambda c: (m.get(c) or float('nan'))
    # SQ counters are summed over the chip's SEs/CUs as rocprofv3 reports them; ratios below only divide like by like
    # 1024 SIMDs (256 CUs x 4), each busy 64 cycles per v_mfma_f32_32x32x2_f32; GRBM_GUI_ACTIVE / 8 = cycles the launch took
    print('   > MFMA pipe utilisation = MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs)   %.3f' % (
        g('SQ_VALU_MFMA_BUSY_CYCLES') / (g('GRBM_GUI_ACTIVE') / 8.0 * 1024.0)))
    print('   > MFMA busy cycles per MFMA instruction                            %.1f' % (g('SQ_VALU_MFMA_BUSY_CYCLES') / g('SQ_INSTS_MFMA')))
    print('   > launch cycles (GUI_ACTIVE/8) per MFMA instruction per SIMD        %.1f' % (
        g('GRBM_GUI_ACTIVE') / 8.0 / (g('SQ_INSTS_MFMA') / 1024.0)))
    print('   > VALU insts per MFMA inst          %.3f' % ((g('SQ_INSTS_VALU') - g('SQ_INSTS_MFMA')) / g('SQ_INSTS_MFMA')))
    print('   > LDS insts per MFMA inst           %.3f' % (g('SQ_INSTS_LDS') / g('SQ_INSTS_MFMA')))
    print('   > SALU insts per MFMA inst          %.3f' % (g('SQ_INSTS_SALU') / g('SQ_INSTS_MFMA')))
    print('   > wave-cycle split wait/issue-stall/active  %.3f / %.3f / %.3f' % (
        g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'), g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'), g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES')))
    print('   > LDS issue stall / wave cycles     %.3f' % (g('SQ_WAIT_INST_LDS') / g('SQ_WAVE_CYCLES')))

#!/usr/bin/env python3
"""Summarise the rocprofv3 output of profiles/prof_recipe.sh (per-dispatch PMC table of one forward).
usage: python profiles/summarize_pmc.py gpurun_out/prof1 > profiles/<round>/pmc_table.txt"""
import collections
import csv
import glob
import os
import sys


def load(d):
    f = max(glob.glob(d + '/runc/*_counter_collection.csv'), key=os.path.getmtime)   # newest run
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = int(r['Dispatch_Id'])
        per.setdefault(k, {'name': r['Kernel_Name'], 'grid': r['Grid_Size'], 't0': int(r['Start_Timestamp']),
                           't1': int(r['End_Timestamp'])})
        per[k][r['Counter_Name']] = float(r['Counter_Value'])
    return per


def one_forward(per):
    """Dispatches of the last complete forward: everything after the previous decoder launch up to and
    including the last one (r3d_decode_f32 closes every forward)."""
    ids = sorted(per)
    ends = [i for i in ids if per[i]['name'].startswith('r3d_decode')]
    s, e = ends[-2], ends[-1]
    return [per[i] for i in ids if s < i <= e and per[i]['name'].startswith('r3d')]


root = sys.argv[1]
batch = '?'
try:
    import json
    batch = json.load(open(root + '/bench_line.json'))['config']['batch_per_gpu']
except Exception:
    pass
f1, f2, f3, f4 = (one_forward(load('%s/pmc%d' % (root, i))) for i in (1, 2, 3, 4))
print('# one forward (B=%s, RF 243, pos+trj); cycles in millions (SQ_* quad-cycle counters x4); FETCH_SIZE x2 per '
      'MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); clk = GRBM_GUI_ACTIVE / 8 XCDs / duration' % batch)
print('%-20s %9s %8s %8s %8s %8s %8s %8s | %7s %6s | %8s %8s %5s' % (
    'kernel', 'grid', 'dur_us', 'waveMcy', 'mfmaMcy', 'waitAny', 'waitInst', 'active', 'ldsIdxM', 'clkGHz', 'fetchMB',
    'writeMB', 'L2hit'))
for a, b, c, d in zip(f1, f2, f3, f4):
    dur = (a['t1'] - a['t0']) / 1e3
    print('%-20s %9s %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f | %7.2f %6.2f | %8.1f %8.1f %5.1f' % (
        a['name'][:20], a['grid'], dur, a['SQ_WAVE_CYCLES'] * 4 / 1e6, a['SQ_VALU_MFMA_BUSY_CYCLES'] / 1e6,
        a['SQ_WAIT_ANY'] * 4 / 1e6, a['SQ_WAIT_INST_ANY'] * 4 / 1e6, a['SQ_ACTIVE_INST_ANY'] * 4 / 1e6,
        b['SQ_LDS_IDX_ACTIVE'] / 1e6, b['GRBM_GUI_ACTIVE'] / 8 / (b['t1'] - b['t0']), c['FETCH_SIZE'] * 2 / 1e3,
        d['WRITE_SIZE'] / 1e3, 100 * d['TCC_HIT_sum'] / max(1, d['TCC_HIT_sum'] + d['TCC_MISS_sum'])))

MAIN = ('r3d_gemm', 'r3d_forward')      # the GEMM launches of the staged form / the single launch that holds all of them
tot_f = sum(c['FETCH_SIZE'] * 2 / 1e3 for c in f3 if c['name'].startswith(MAIN))
tot_w = sum(d['WRITE_SIZE'] / 1e3 for d in f4 if d['name'].startswith(MAIN))
n = sum(1 for c in f3 if c['name'].startswith(MAIN))
busy = sum(a['SQ_VALU_MFMA_BUSY_CYCLES'] for a in f1 if a['name'].startswith(MAIN))
dur = sum((a['t1'] - a['t0']) / 1e3 for a in f1 if a['name'].startswith(MAIN))
print('# %s: %d launch(es), fetch %.1f MB + write %.1f MB per forward = %.0f bytes per launch (profiles/traffic.json); '
      'MFMA-busy %.1f M SIMD-cycles over %.1f us' % ('/'.join(sorted(set(c['name'].split('(')[0][:18] for c in f3 if c['name'].startswith(MAIN)))),
                                                     n, tot_f, tot_w, (tot_f + tot_w) * 1e6 / max(n, 1), busy / 1e6, dur))

#!/usr/bin/env python3
"""Summarise the rocprofv3 output of profiles/prof_recipe.sh: the PMC table of the kernels the traced bench line times.

usage: python profiles/summarize_pmc.py gpurun_out/<name> > profiles/<name>/pmc_table.txt
       (also writes <dir>/pmc_summary.json: the figures bench.py merges into `roofline` through profiles/pmc.json)

The kernel is picked BY NAME - `roofline.kernel` of the run's bench_line.json (r3d_forward_uv_f32 for the pixel-keypoint
workloads, which also run the rays mode's r3d_forward_f32 for comparison) - and the last dispatch of that name in each PMC
pass is taken; one table row per r3d_* kernel name of the pass.  Units as MI355X_MICROARCH.md prescribes: SQ_* quad-cycle
counters x4, SQ_VALU_MFMA_BUSY_CYCLES in cycles, FETCH_SIZE x2 (gfx950 tallies 128-byte requests as 64), KiB -> bytes;
clock = GRBM_GUI_ACTIVE / 8 XCDs / duration."""
import collections
import csv
import glob
import json
import os
import sys

SIMDS = 256 * 4
FLOP_PER_BUSY_CYCLE = {"f32": 64.0, "bf16x3": 1024.0 / 6.0}    # per SIMD: v_mfma_f32_32x32x2_f32 4096 FLOP / 64 cycles; bf16 32x32x16 / 6 products


def base_name(n):
    return n.split('(')[0].split('.')[0].strip()


def load(d):
    files = glob.glob(d + '/**/*_counter_collection.csv', recursive=True)
    f = max(files, key=os.path.getmtime)
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = int(r['Dispatch_Id'])
        per.setdefault(k, {'name': base_name(r['Kernel_Name']), 'grid': r['Grid_Size'], 't0': int(r['Start_Timestamp']),
                           't1': int(r['End_Timestamp'])})
        per[k][r['Counter_Name']] = float(r['Counter_Value'])
    return per


def last_by_name(per):
    """{kernel name: its last dispatch of the pass} for the library's kernels, in order of first appearance."""
    out = collections.OrderedDict()
    for i in sorted(per):
        if per[i]['name'].startswith('r3d'):
            out[per[i]['name']] = per[i]
    return out


def main():
    root = sys.argv[1]
    line = {}
    try:
        line = json.load(open(root + '/bench_line.json'))
    except Exception:
        pass
    cfg = line.get('config', {})
    rl = line.get('roofline', {})
    main_kernel = rl.get('kernel', 'r3d_forward_f32')
    dtype = line.get('dtype', 'f32')
    p1, p2, p3, p4 = (last_by_name(load('%s/pmc%d' % (root, i))) for i in (1, 2, 3, 4))
    print('# one forward: %s windows, RF %s, %s joints (%s); timed kernel %s; cycles in millions (SQ_* quad-cycle counters x4); '
          'FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B); clk = GRBM_GUI_ACTIVE / 8 XCDs / duration; busy = MFMA-busy '
          'cycles / (1024 SIMDs x duration x clk)' % (cfg.get('batch_per_gpu', rl.get('windows', '?')), cfg.get('receptive_field', '?'), cfg.get('joints', '?'),
                                                       str(cfg.get('workload', ''))[:40], main_kernel))
    print('%-22s %9s %8s %8s %8s %8s %8s %8s %5s | %7s %6s | %8s %8s %5s' % (
        'kernel', 'grid', 'dur_us', 'waveMcy', 'mfmaMcy', 'waitAny', 'waitInst', 'active', 'busy', 'ldsIdxM', 'clkGHz', 'fetchMB',
        'writeMB', 'L2hit'))
    summary = {}
    for name, a in p1.items():
        b, c, d = p2.get(name), p3.get(name), p4.get(name)
        if not (b and c and d):
            continue
        dur = (a['t1'] - a['t0']) / 1e3                                   # us, pass 1 (the pass the SQ counters come from)
        clk = b['GRBM_GUI_ACTIVE'] / 8 / (b['t1'] - b['t0'])              # cycles per ns = GHz (pass 2's own duration)
        busy_frac = a['SQ_VALU_MFMA_BUSY_CYCLES'] / (SIMDS * dur * 1e3 * clk)
        fetch, write = c['FETCH_SIZE'] * 2 * 1e3, d['WRITE_SIZE'] * 1e3   # bytes
        l2 = 100 * d['TCC_HIT_sum'] / max(1, d['TCC_HIT_sum'] + d['TCC_MISS_sum'])
        print('%-22s %9s %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %5.2f | %7.2f %6.2f | %8.1f %8.1f %5.1f' % (
            name[:22], a['grid'], dur, a['SQ_WAVE_CYCLES'] * 4 / 1e6, a['SQ_VALU_MFMA_BUSY_CYCLES'] / 1e6,
            a['SQ_WAIT_ANY'] * 4 / 1e6, a['SQ_WAIT_INST_ANY'] * 4 / 1e6, a['SQ_ACTIVE_INST_ANY'] * 4 / 1e6, busy_frac,
            b['SQ_LDS_IDX_ACTIVE'] / 1e6, clk, fetch / 1e6, write / 1e6, l2))
        summary[name] = {"dur_us_pmc_pass": round(dur, 1), "clk_ghz_pmc_pass": round(clk, 3),
                         "mfma_busy_cycles": a['SQ_VALU_MFMA_BUSY_CYCLES'], "mfma_busy_frac": round(busy_frac, 4),
                         "executed_flops": a['SQ_VALU_MFMA_BUSY_CYCLES'] * FLOP_PER_BUSY_CYCLE.get(dtype, 64.0),
                         "traffic_bytes": int(fetch + write), "fetch_bytes": int(fetch), "write_bytes": int(write),
                         "l2_hit_pct": round(l2, 1)}
    m = summary.get(main_kernel)
    if m:
        print('# %s: fetch %.1f MB + write %.1f MB = %d bytes per launch; MFMA-busy %.1f M SIMD-cycles over %.1f us at %.2f GHz = %.2f '
              'of the SIMD-cycles; executed %.2f GFLOP' % (main_kernel, m['fetch_bytes'] / 1e6, m['write_bytes'] / 1e6, m['traffic_bytes'],
                                                           m['mfma_busy_cycles'] / 1e6, m['dur_us_pmc_pass'], m['clk_ghz_pmc_pass'],
                                                           m['mfma_busy_frac'], m['executed_flops'] / 1e9))
    json.dump({"timed_kernel": main_kernel, "dtype": dtype, "batch_per_gpu": cfg.get('batch_per_gpu', rl.get('windows')),
               "receptive_field": cfg.get('receptive_field'), "kernels": summary}, open(root + '/pmc_summary.json', 'w'), indent=1)


if __name__ == '__main__':
    main()

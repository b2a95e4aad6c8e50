#!/usr/bin/env python3
"""Summarise a scripts/profile_estep.sh output directory: per-kernel average duration from the
kernel trace stats and per-launch PMC averages for the gbrs kernels."""
import collections, csv, glob, sys
d = sys.argv[1]
for f in glob.glob(d + '/kt/*/*_kernel_stats.csv'):
    print('# kernel stats (rocprofv3 --kernel-trace --stats)')
    for r in csv.DictReader(open(f)):
        if 'gbrs::' in r['Name']:
            print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f} max_us={float(r['MaxNs'])/1e3:9.1f}")
for sub in ('pmc_sq', 'pmc_sq2', 'pmc_grbm', 'pmc_fetch', 'pmc_write'):
    for f in glob.glob(f'{d}/{sub}/*/*_counter_collection.csv'):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'gbrs::' in k and ('tile_estep' in k or 'gather' in k or 'mstep' in k):
                agg[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        print(f'# {sub} (per-launch mean)')
        for k, v in agg.items():
            print(' ', k, {c: round(sum(x) / len(x), 1) for c, x in v.items()})

# derived figures for the E-step kernel -> JSON fragment for profiles/pmc_traffic.json
import json, re
EST = re.compile(r'tile_estep_kernel<\d+, (true|false), false, ')      # the EM step's launches, not prepare()'s (ONES)
est = {}
for sub in ('pmc_sq', 'pmc_sq2', 'pmc_grbm', 'pmc_fetch', 'pmc_write'):
    for f in glob.glob(f'{d}/{sub}/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if EST.search(r['Kernel_Name']):
                est.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
avg = {k: sum(v) / len(v) for k, v in est.items()}
dur_us = None
for f in glob.glob(d + '/kt/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if EST.search(r['Name']):
            dur_us = float(r['AverageNs']) / 1e3
if avg and dur_us:
    out = dict(kernel_avg_us=dur_us)
    if 'FETCH_SIZE' in avg and 'WRITE_SIZE' in avg:
        out.update(fetch_size_kb=avg['FETCH_SIZE'], write_size_kb=avg['WRITE_SIZE'],
                   bytes_per_launch=int(avg['FETCH_SIZE'] * 1024 * 2 + avg['WRITE_SIZE'] * 1024))
    if 'GRBM_GUI_ACTIVE' in avg:
        # rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs (MI355X_MICROARCH.md, DVFS note); reads high on short dispatches
        out['shader_clock_mhz'] = avg['GRBM_GUI_ACTIVE'] / 8.0 / dur_us
    if 'SQ_INSTS_VALU' in avg:
        out['valu_insts_per_launch'] = avg['SQ_INSTS_VALU']
        clk = out.get('shader_clock_mhz', 2400.0)
        # every wave-level vector instruction takes >= 4 cycles of its SIMD's issue; 1,024 SIMDs
        out['valu_util'] = avg['SQ_INSTS_VALU'] * 4.0 / (1024.0 * dur_us * clk)
    if 'SQ_LDS_BANK_CONFLICT' in avg:
        out['lds_bank_conflict_cycles'] = avg['SQ_LDS_BANK_CONFLICT']
    if 'SQ_LDS_IDX_ACTIVE' in avg:
        out['lds_idx_active_cycles'] = avg['SQ_LDS_IDX_ACTIVE']
    for k in ('SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_WAVES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_LDS',
              'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
        if k in avg:
            out[k] = avg[k]
    print('# E-step JSON')
    print(json.dumps(out))

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
for sub in ('pmc_sq', 'pmc_sq2', 'pmc_fetch', 'pmc_write'):
    for f in glob.glob(f'{d}/{sub}/*/*_counter_collection.csv'):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'gbrs::' in k and ('tile_estep' in k or 'gather' in k or 'mstep' in k):
                agg[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        print(f'# {sub} (per-launch mean)')
        for k, v in agg.items():
            print(' ', k, {c: round(sum(x) / len(x), 1) for c, x in v.items()})

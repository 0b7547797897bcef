#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv per kernel: python tools/pmc_sum.py DIR [divide_by_launches]"""
import csv, glob, collections, sys
d = sys.argv[1]
for f in glob.glob(d + '/**/*_counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:48]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        calls[(k, r['Counter_Name'])] += 1
    for k, v in agg.items():
        if any(s in k for s in ('voigt', 'tud_', 'prep', 'ils', 'radiance')):
            n = max(calls[(k, c)] for c in v)
            print(k, 'launches', n, {a: '%.4g' % (b / n) for a, b in sorted(v.items())})

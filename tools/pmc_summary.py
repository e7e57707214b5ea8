#!/usr/bin/env python
"""Average PMC counters per kernel from rocprofv3 --pmc csv output dirs: python tools/pmc_summary.py gpurun_out/pmc_x_*"""
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void gpa::', '').replace('gpa::', '')
            acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name in sorted(acc):
    print(name)
    for c, v in sorted(acc[name].items()):
        print('    %-28s n=%4d mean=%.4g' % (c, len(v), sum(v) / len(v)))

#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel (short name), mean of each counter
over its dispatches.  usage: pmc_summary.py <counter_collection.csv> [...]"""
import csv, re, sys
from collections import defaultdict
for path in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            name = re.sub(r'\(anonymous namespace\)::', '', row['Kernel_Name'])
            name = re.sub(r'\(.*', '', name).replace('void ', '')
            acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
    print('==', path)
    for k, cs in acc.items():
        n = max(len(v) for v in cs.values())
        print(f'{k:40s} n={n:3d} ' + ' '.join(f'{c}={sum(v)/len(v):.4g}' for c, v in sorted(cs.items())))

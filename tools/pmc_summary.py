#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel name: tools/pmc_summary.py <dir> -> prints a table."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name'].split('(')[0][:48]
        a = acc[name][row['Counter_Name']]
        a[0] += float(row['Counter_Value'])
        a[1] += 1
for name in sorted(acc):
    print(name)
    for ctr in sorted(acc[name]):
        s, n = acc[name][ctr]
        print('    %-28s avg %16.1f  over %d dispatches' % (ctr, s / n, n))

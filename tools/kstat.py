#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --kernel-trace --stats output directory: python tools/kstat.py DIR [NAME-PART ...]
(prints the kernels whose name contains one of the parts, default: the round kernels), and ms_per_step of DIR.json if present."""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
parts = sys.argv[2:] or ['k_round', 'k_sample', 'k_apply']
f = glob.glob(os.path.join(d, '*', '*kernel_stats.csv')) + glob.glob(os.path.join(d, '*kernel_stats.csv'))
for r in csv.DictReader(open(f[0])):
    if any(p in r['Name'] for p in parts):
        print('%-48s calls %6s  avg %10.1f us' % (r['Name'][:48], r['Calls'], float(r['AverageNs']) / 1e3))
if os.path.exists(d + '.json'):
    j = json.load(open(d + '.json'))
    print('ms_per_step %.2f  value %.4g' % (j['ms_per_step'], j['value']))

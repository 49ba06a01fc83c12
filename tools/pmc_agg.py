#!/usr/bin/env python
"""Aggregate rocprofv3 PMC csv output(s): mean counter value per (kernel substring, counter).
    python tools/pmc_agg.py <dir> [kernel substring]"""
import collections, csv, glob, sys
pat = sys.argv[2] if len(sys.argv) > 2 else 'conv_igemm'
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    v = agg[k]
    print('%-28s n=%4d mean=%16.1f' % (k, len(v), sum(v) / len(v)))

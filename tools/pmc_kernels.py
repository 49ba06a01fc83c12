#!/usr/bin/env python
"""Per-kernel sums of rocprofv3 PMC counters (+ durations from the kernel trace of the same run) as JSON.
    python tools/pmc_kernels.py <rocprof output dir> [kernel substring ...]"""
import collections, csv, glob, json, sys
d = sys.argv[1]
pats = sys.argv[2:] or ['conv_wino', 'conv_igemm']
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
nl = collections.defaultdict(int)
for fn in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        for p in pats:
            if p in r['Kernel_Name']:
                cnt[p][r['Counter_Name']] += float(r['Counter_Value'])
dur = collections.defaultdict(list)
for fn in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        for p in pats:
            if p in r['Kernel_Name']:
                dur[p].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = {}
for p in pats:
    c = dict(cnt[p])
    o = {'launches': len(dur[p]), 'sum_us': round(sum(dur[p]), 1), 'counters': c}
    if 'GRBM_GUI_ACTIVE' in c and dur[p]:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        o['clock_GHz'] = round(c['GRBM_GUI_ACTIVE'] / 8 / (sum(dur[p]) * 1e3), 3)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
        o['mfma_util'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (c['GRBM_GUI_ACTIVE'] / 8), 4)
    if 'SQ_INSTS_VALU' in c and 'SQ_INSTS_MFMA' in c and c['SQ_INSTS_MFMA']:
        o['non_mfma_valu_per_mfma'] = round((c['SQ_INSTS_VALU'] - c['SQ_INSTS_MFMA']) / c['SQ_INSTS_MFMA'], 2)
    if c.get('SQ_LDS_IDX_ACTIVE'):
        # cycles the LDS is stalled by bank conflicts per cycle it is busy with indexed accesses
        o['lds_bank_conflict_frac'] = round(c.get('SQ_LDS_BANK_CONFLICT', 0.0) / c['SQ_LDS_IDX_ACTIVE'], 4)
        if 'GRBM_GUI_ACTIVE' in c:
            o['lds_busy_frac'] = round(c['SQ_LDS_IDX_ACTIVE'] / 256 / (c['GRBM_GUI_ACTIVE'] / 8), 4)      # per CU, of the kernel's cycles
    out[p] = o
print(json.dumps(out, indent=1))

#!/usr/bin/env python
"""MFMA-pipe utilisation of the conv engine from one rocprofv3 PMC pass.
    python tools/pmc_mfma_summary.py <dir with counter_collection csv> > profiles/rNN_pmc_mfma.json"""
import collections, csv, glob, json, sys
agg = collections.defaultdict(float)
n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value'])
            n[r['Counter_Name']] += 1
out = {'command': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES '
                  '--kernel-trace --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline',
       'kernel': 'conv_igemm_kernel (all launches)', 'launches': n.get('GRBM_GUI_ACTIVE', 0), 'sum': dict(agg)}
cyc = agg['GRBM_GUI_ACTIVE'] / 8.0
out['mfma_util'] = round(agg['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / cyc, 4)
out['mfma_util_formula'] = ('SQ_VALU_MFMA_BUSY_CYCLES / (256 CU x 4 SIMD) / (GRBM_GUI_ACTIVE / 8 XCD): fraction of kernel cycles each '
                            'MFMA pipe is busy (v_mfma_f32_32x32x2_f32 = 64 busy cycles)')
out['non_mfma_valu_per_mfma'] = round((agg['SQ_INSTS_VALU'] - agg['SQ_INSTS_MFMA']) / agg['SQ_INSTS_MFMA'], 2)
print(json.dumps(out, indent=1))

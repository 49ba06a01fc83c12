#!/usr/bin/env python
"""VALU issue picture of the fused render from one rocprofv3 PMC pass.
    python tools/pmc_render_summary.py <dir with counter_collection csv> > profiles/rNN_pmc_render.json"""
import collections, csv, glob, json, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'render_average' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
mean = {k: sum(v) / len(v) for k, v in agg.items()}
n = len(next(iter(agg.values())))
cyc = mean['GRBM_GUI_ACTIVE'] / 8.0
waves = mean['SQ_WAVES']
vpw = mean['SQ_INSTS_VALU'] / waves
out = {'command': 'rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU '
                  'SQ_INSTS_SMEM --kernel-trace --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs',
       'kernel': 'render_average_kernel<2> with footprints (740x1882 canvas, per-launch means over %d launches)' % n,
       'mean': mean, 'kernel_cycles': cyc, 'valu_instructions_per_wave': round(vpw, 1),
       'waves_per_simd': round(waves / 1024.0, 2)}
# every VALU instruction of a wave64 occupies its SIMD for 4 cycles, the quarter-rate v_log_f32 for 16: a two-view wave
# issues 126 of them, a single-view (two rows per lane) wave too; tiles no view reaches issue none
out['valu_issue_busy_estimate'] = round((waves / 1024.0) * (vpw * 4 + 126 * 12) / cyc, 3)
out['estimate_formula'] = ('(waves/SIMD x (VALU instr/wave x 4 cycles + 126 v_log_f32 x 12 extra cycles)) / kernel cycles; '
                           'upper estimate: waves of tiles that no view reaches issue no v_log_f32')
out['valu_active_frac'] = round(mean['SQ_ACTIVE_INST_VALU'] * 4 / 1024.0 / cyc, 3)      # SQ_ACTIVE_INST_* count quad-cycles, summed over the chip
print(json.dumps(out, indent=1))

#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), written as JSON.
    python tools/pmc_summary.py <dir with FETCH csv> <dir with WRITE csv> > profiles/rNN_pmc_hbm.json"""
import collections
import csv
import glob
import json
import sys

KEYS = (('conv_wino43', 'conv_wino43_kernel'), ('conv_wino_kernel', 'conv_wino_kernel'), ('conv_igemm', 'conv_igemm_kernel'), ('render_average', 'render_average_kernel'), ('cost_volume', 'cost_volume_kernel'),
        ('maxpool', 'maxpool_kernel'), ('linear_kernel', 'linear_kernel'), ('linear_grouped', 'linear_grouped_kernel'), ('homo_warp', 'homo_warp_kernel'), ('stem_pool_kernel', 'stem_pool_kernel'))


def load(path):
    agg = collections.defaultdict(list)
    for fn in glob.glob(path + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            for pat, key in KEYS:
                if pat in r['Kernel_Name']:
                    agg[key].append(float(r['Counter_Value']))
    return agg


def durations(path):
    agg = collections.defaultdict(list)
    for fn in glob.glob(path + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            for pat, key in KEYS:
                if pat in r['Kernel_Name']:
                    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return agg


f, w = load(sys.argv[1]), load(sys.argv[2])
dur = durations(sys.argv[1])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cmd = sys.argv[4] if len(sys.argv) > 4 else 'python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs'
out = {'steps_profiled': steps, 'command': 'rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- ' + cmd +
                  ' (one pass per counter; `launches` = all launches of the run, `steps_profiled` clips)',
       'units': 'counter value x 1000 = bytes.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies the 128-B '
                'requests of a 16-B/lane stream at 64 B, so the conv engine (buffer_load_dwordx4) and maxpool/linear (float4 '
                'loads, homography sampler: float4 taps) fetch bytes = 2 x FETCH_SIZE; dword gathers (render, cost volume) and WRITE_SIZE are used as reported.',
       'kernels': {}}
for key in f:
    corr = 2.0 if key in ('conv_wino43_kernel', 'conv_wino_kernel', 'conv_igemm_kernel', 'maxpool_kernel', 'linear_kernel', 'linear_grouped_kernel', 'homo_warp_kernel', 'stem_pool_kernel') else 1.0
    fa, wa = sum(f[key]) / len(f[key]), sum(w[key]) / len(w[key])
    out['kernels'][key] = {'launches': len(f[key]), 'FETCH_SIZE_avg': round(fa, 1), 'WRITE_SIZE_avg': round(wa, 1),
                           'fetch_correction': corr, 'hbm_bytes_per_launch': round((corr * fa + wa) * 1000.0)}
    if dur.get(key):
        us = sum(dur[key]) / len(dur[key])
        out['kernels'][key]['avg_us_under_pmc'] = round(us, 2)
        out['kernels'][key]['hbm_TB_per_s'] = round((corr * fa + wa) * 1000.0 / us / 1e6, 3)
# the conv engine as one family (what bench.py's roofline block quotes): all Winograd + implicit-GEMM launches together
fam = [k for k in ('conv_wino43_kernel', 'conv_wino_kernel', 'conv_igemm_kernel', 'stem_pool_kernel') if k in f]
if fam:
    nl = sum(len(f[k]) for k in fam)
    tot = sum(2.0 * sum(f[k]) + sum(w[k]) for k in fam) * 1000.0
    out['conv_family'] = {'kernels': fam, 'launches': nl, 'hbm_bytes_per_launch': round(tot / nl)}
    if all(dur.get(k) for k in fam):
        out['conv_family']['avg_us_under_pmc'] = round(sum(sum(dur[k]) for k in fam) / nl, 2)
print(json.dumps(out, indent=1))

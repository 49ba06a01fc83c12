#!/usr/bin/env python
"""Append to a kernel-stats summary how the rocprofv3 trace compares with bench.py's own HIP-event averages, for the profiled
run the trace belongs to and for the un-profiled run of the same command.
    python tools/prof_vs_events.py <kernel_stats.txt> <bench_profiled.json> <bench_unprofiled.json>"""
import json, re, sys


def last_json(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith('{'):
            return json.loads(line)
    raise SystemExit('no JSON line in ' + path)


stats = {}
for line in open(sys.argv[1]):
    m = re.match(r'^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)%$', line.rstrip())
    if m:
        stats[m.group(1)] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
prof, unprof = last_json(sys.argv[2]), last_json(sys.argv[3])
print('#')
print('# trace vs HIP events (bench.py per_kernel.avg_launch_us; same command):')
print('#   %-22s %12s %14s %16s %10s' % ('kernel family', 'trace avg us', 'events profiled', 'events unprofiled', 'trace/ev'))
for fam in ('conv_wino43_kernel', 'conv_wino_kernel', 'conv_igemm_kernel', 'stem_pool_kernel'):
    rows = [v for k, v in stats.items() if fam in k]
    calls = sum(r[0] for r in rows)
    tot = sum(r[1] for r in rows)
    if not calls:
        continue
    note = ''
    if fam == 'conv_igemm_kernel':
        # a split-K convolution is two kernels inside one event bracket: its partial sums are reduced by splitk_reduce_kernel
        red = [v for k, v in stats.items() if 'splitk_reduce_kernel' in k]
        if red:
            tot += sum(r[1] for r in red)
            note = '   (+ %d splitk_reduce_kernel launches, inside the same brackets)' % sum(r[0] for r in red)
    tr = tot / calls
    if fam not in prof['roofline']['per_kernel']:
        continue
    ep = prof['roofline']['per_kernel'][fam]['avg_launch_us']
    eu = unprof['roofline']['per_kernel'][fam]['avg_launch_us']
    print('#   %-22s %12.2f %14.2f %16.2f %10.3f%s' % (fam, tr, ep, eu, tr / ep, note))
print('#   frames/s: profiled run %.1f, un-profiled run %.1f; conv engine ms per clip (events): %.3f / %.3f'
      % (prof['value'], unprof['value'], prof['roofline']['kernel_ms_per_step'], unprof['roofline']['kernel_ms_per_step']))
print('#   (HIP events bracket a launch on the stream and include its ~2 us launch gap; the trace is the kernel alone)')

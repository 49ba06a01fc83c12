#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.
    python tools/prof_summary.py gpurun_out/prof/x_results.db > profiles/rNN_name.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
tot = sum(r[2] for r in rows)
print('# rocprofv3 --kernel-trace --stats summary of %s' % sys.argv[1])
if len(sys.argv) > 2:
    print('# command: %s' % sys.argv[2])
print('# durations in microseconds; total kernel time %.1f us' % tot)
print('%-96s %7s %12s %10s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
for name, calls, total, avg, pct in rows:
    print('%-96s %7d %12.1f %10.2f %6.2f%%' % (name[:96], calls, total, avg, pct))

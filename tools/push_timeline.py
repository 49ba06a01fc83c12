#!/usr/bin/env python
"""Timeline of ONE steady-state streaming push out of a rocprofv3 kernel trace (rocpd sqlite):
    python tools/push_timeline.py <results.db> [marker-substring] [which-from-the-end]
The push is delimited by two consecutive dispatches of the marker kernel (default: the stem, launched once per push); every
dispatch in between is printed in start order with its queue, start offset, duration and the gap to the end of the latest
kernel that finished before it started (a negative gap = it overlapped a kernel of another branch)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2] if len(sys.argv) > 2 else 'stem_pool'
    which = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rows = db.execute('select name, start, end, queue_id, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count '
                      'from kernels order by start').fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < which + 1:
        raise SystemExit('marker %r seen %d times' % (marker, len(marks)))
    a, b = marks[-which - 1], marks[-which]
    # a push starts with its input copies / layout kernels just before the marker: back up to the previous push's last kernel
    push = rows[a:b]
    t0 = push[0][1]
    print('# one push = dispatches [%d, %d) of %s, delimited by %r' % (a, b, sys.argv[1], marker))
    print('# %3s %-72s %5s %9s %8s %8s  %s' % ('#', 'kernel', 'queue', 'start_us', 'dur_us', 'gap_us', 'grid (workgroups) / lds / vgpr'))
    last_end = t0
    busy = 0.0
    for i, (name, s, e, q, gx, gy, gz, wx, lds, vg) in enumerate(push):
        wgs = (gx // max(wx, 1)) * gy * gz
        print('  %3d %-72s %5s %9.2f %8.2f %8.2f  %d / %d / %d' % (i, name[:72], q, (s - t0) / 1e3, (e - s) / 1e3, (s - last_end) / 1e3,
                                                               wgs, lds, vg))
        busy += (e - s) / 1e3
        last_end = max(last_end, e)
    print('# %d dispatches, span %.1f us, sum of kernel durations %.1f us' % (len(push), (last_end - t0) / 1e3, busy))
    per = {}
    for name, s, e, *_ in push:
        k = name.split('(')[0][:60]
        c = per.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += (e - s) / 1e3
    print('# per kernel:')
    for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print('#   %-62s %3d %9.2f' % (k, c, t))


if __name__ == '__main__':
    main()

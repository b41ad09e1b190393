"""Compressed per-queue timeline of the last window in a rocprofv3 rocpd database: runs of same-named kernels on one
queue are merged (count, first start, last end, busy us).   python tools/rocpd_phases.py results.db [window index, default -1 = last] [window_marker_kernel]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*', '', name)
    return name.replace('void ', '').replace('hvr::', '').replace('unsigned short', 'bf16')[:60]


def main(path, which='-1', marker='stem_fused_kernel'):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute('select name, start, end, queue_id from kernels order by start').fetchall()
    starts = [r[1] for r in rows if marker in r[0]]
    # window boundaries = marker launches that are far apart (one or two per window)
    bounds = [starts[0]]
    for s in starts[1:]:
        if s - bounds[-1] > 3e6:
            bounds.append(s)
    print('# %d windows; gaps (ms): %s' % (len(bounds) - 1, ' '.join('%.1f' % ((b - a) / 1e6) for a, b in zip(bounds, bounds[1:]))))
    k = int(which)
    lo, hi = bounds[k - 1], bounds[k]
    rows = [r for r in rows if lo <= r[1] < hi]
    print('# window of %.3f ms, %d kernels' % ((hi - lo) / 1e6, len(rows)))
    runs = []
    for name, s, e, q in rows:
        n = short(name)
        if runs and runs[-1][0] == q and runs[-1][1] == n:
            runs[-1][3] = e
            runs[-1][4] += 1
            runs[-1][5] += e - s
        else:
            runs.append([q, n, s, e, 1, e - s])
    for q, n, s, e, c, busy in runs:
        print('q%-2d %9.1f -> %9.1f  x%-3d busy %8.1f us  %s' % (q, (s - lo) / 1e3, (e - lo) / 1e3, c, busy / 1e3, n))


if __name__ == '__main__':
    main(*sys.argv[1:])

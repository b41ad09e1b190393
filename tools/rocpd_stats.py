"""Summarise a rocprofv3 rocpd database (`--kernel-trace --stats` output, *_results.db) into the
per-kernel table kept under profiles/:  name, calls, total ms, avg us, min us, max us, % of GPU time.

    python tools/rocpd_stats.py gpurun_out/prof1/bench_results.db > profiles/r01_bench_kernel_stats.txt

ROCPD_SPLIT="name_substring:microseconds[,...]": launches of a kernel whose name holds the substring are listed as two rows, shorter /
longer than the threshold -- e.g. relation_scores_bt_kernel:100 separates its one-window launches from the four-window (grouped) ones,
which share a name and a grid.
"""
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*', '', name)            # drop the argument list
    name = name.replace('void ', '').replace('hvr::', '')
    return name[:110]


def main(path):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = cur.execute('select %s, start, end from kernels' % name_col).fetchall()
    split = [kv.split(':') for kv in os.environ.get('ROCPD_SPLIT', '').split(',') if ':' in kv]
    agg = {}
    for name, s, e in rows:
        dur = (e - s) / 1e3
        key = short(name)
        for sub, thr in split:
            if sub in key:
                key += '  [launches %s %s us]' % ('<' if dur < float(thr) else '>=', thr)
        d = agg.setdefault(key, [0, 0.0, 1e30, 0.0])
        d[0] += 1
        d[1] += dur
        d[2] = min(d[2], dur)
        d[3] = max(d[3], dur)
    total = sum(d[1] for d in agg.values())
    span = (max(r[2] for r in rows) - min(r[1] for r in rows)) / 1e3
    print('# %d kernel dispatches, %.3f ms of kernel time, %.3f ms first-start to last-end' % (len(rows), total / 1e3, span / 1e3))
    print('%-112s %7s %11s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', '%'))
    for name, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-112s %7d %11.3f %10.2f %10.2f %10.2f %6.2f' % (name, d[0], d[1] / 1e3, d[1] / d[0], d[2], d[3], 100 * d[1] / total))


if __name__ == '__main__':
    main(sys.argv[1])

"""GPU busy / idle time over the steady-state windows of a rocprofv3 kernel trace (rocpd database).

    python tools/gpu_idle.py gpurun_out/prof/bench_results.db [marker-kernel-substring] [first] [windows]

The marker kernel (default: stem_fused_kernel, launched once per window) delimits windows; `windows` complete ones
starting at window index `first` (skip the warm-up) are analysed: wall span, union of kernel intervals (streams may
overlap), idle gaps by size.
"""
import sqlite3
import sys


def main(path, marker='stem_fused_kernel', first=3, windows=5):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = sorted(cur.execute('select %s, start, end from kernels' % name_col).fetchall(), key=lambda r: r[1])
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    first, windows = int(first), int(windows)
    assert len(marks) > first + windows, 'not enough windows in the trace'
    lo, hi = marks[first], marks[first + windows]
    seg = rows[lo:hi]
    span = (seg[-1][2] - seg[0][1]) / 1e3
    busy, gaps, big, cur_end, prev = 0.0, [], [], seg[0][1], seg[0][0]
    for name, s, e in seg:
        if s > cur_end:
            gaps.append((s - cur_end) / 1e3)
            if s - cur_end > 20e3:
                big.append(((s - cur_end) / 1e3, prev, name))
        if e > cur_end:
            busy += (e - max(s, cur_end)) / 1e3
            cur_end = e
            prev = name
    print('# %d windows, %d dispatches (%.1f / window)' % (windows, len(seg), len(seg) / windows))
    print('span %.3f ms/window   busy %.3f ms/window   idle %.3f ms/window (%.1f%%)' %
          (span / windows / 1e3, busy / windows / 1e3, (span - busy) / windows / 1e3, 100 * (span - busy) / span))
    for lo_us, hi_us in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 1e9)):
        g = [x for x in gaps if lo_us <= x < hi_us]
        print('gaps %4g-%-6g us: %5d / window, %.3f ms / window' % (lo_us, hi_us, len(g) / windows, sum(g) / windows / 1e3))
    print('# gaps > 20 us: idle us, kernel that ended before it -> kernel that started after it')
    for g, a, b in big[:len(big) // windows + 1]:
        print('%8.1f  %s -> %s' % (g, a[:60], b[:60]))


if __name__ == '__main__':
    main(*sys.argv[1:])

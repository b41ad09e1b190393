"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (`--pmc X` pass).

    python tools/rocpd_pmc.py gpurun_out/pmc_fetch/bench_results.db FETCH_SIZE

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts a wide coalesced 128-byte request as 64 bytes
(MI355X_MICROARCH.md, HBM section), so the `corrected_MB` column doubles FETCH_SIZE; WRITE_SIZE is left as is.
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*', '', name)
    return name.replace('void ', '').replace('hvr::', '')[:96]


def main(path, counter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute('select kernel_name, value, duration from counters_collection where counter_name = ?', (counter,)).fetchall()
    agg = {}
    for name, v, dur in rows:
        d = agg.setdefault(short(name), [0, 0.0, 0.0])
        d[0] += 1
        d[1] += v
        d[2] += dur
    factor = 2.0 if counter == 'FETCH_SIZE' else 1.0
    print('# %s per launch (KiB -> MB), %d dispatches' % (counter, len(rows)))
    print('%-98s %6s %12s %13s %10s' % ('kernel', 'calls', 'raw_MB', 'corrected_MB', 'avg_us'))
    for name, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        mb = d[1] / d[0] * 1024 / 1e6
        print('%-98s %6d %12.3f %13.3f %10.2f' % (name, d[0], mb, mb * factor, d[2] / d[0] / 1e3))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])

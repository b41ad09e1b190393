"""Per-kernel averages of every PMC counter found in a rocprofv3 rocpd database (one `--pmc A B C` pass).

    python tools/pmc_dump.py gpurun_out/pmc_x/out_results.db [kernel-substring]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*', '', name)
    return name.replace('void ', '').replace('hvr::', '')[:70]


def main(path, needle=''):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute('select kernel_name, counter_name, value, duration from counters_collection').fetchall()
    agg, counters = {}, []
    for name, cname, v, dur in rows:
        if needle and needle not in name:
            continue
        if cname not in counters:
            counters.append(cname)
        d = agg.setdefault(short(name), {})
        e = d.setdefault(cname, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += v
        e[2] += dur
    print('%-72s %6s %9s ' % ('kernel', 'calls', 'avg_us') + ' '.join('%22s' % c for c in counters))
    for name, d in sorted(agg.items()):
        first = d[counters[0]] if counters[0] in d else list(d.values())[0]
        print('%-72s %6d %9.2f ' % (name, first[0], first[2] / first[0] / 1e3) +
              ' '.join('%22.1f' % (d[c][1] / d[c][0]) if c in d else '%22s' % '-' for c in counters))


if __name__ == '__main__':
    main(*sys.argv[1:])

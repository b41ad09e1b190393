"""Timeline of the last window in a rocprofv3 rocpd database: per kernel start (us from the window's first launch),
duration, queue/stream and name.   python tools/rocpd_timeline.py results.db [n_last_kernels]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*', '', name)
    return name.replace('void ', '').replace('hvr::', '').replace('unsigned short', 'bf16')[:64]


def main(path, n=700):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
    print('# columns:', cols, file=sys.stderr)
    qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    sel = 'name, start, end' + (', ' + qcol if qcol else '') + (', stream_id' if 'stream_id' in cols and qcol != 'stream_id' else '')
    gcols = [c for c in cols if c.startswith('grid') or c.startswith('workgroup')]
    sel += ''.join(', ' + c for c in gcols)
    rows = cur.execute('select %s from kernels order by start' % sel).fetchall()
    rows = rows[-int(n):]
    t0 = rows[0][1]
    for r in rows:
        print('%9.1f %8.1f  q%-3s %-66s %s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, ' '.join(str(x) for x in r[3:5] if x is not None), short(r[0]), ' '.join(str(x) for x in r[5:])))


if __name__ == '__main__':
    main(*sys.argv[1:])

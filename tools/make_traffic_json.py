"""Writes profiles-style `relation_traffic.json` from the two PMC passes over tools/rel_bench.py (FETCH_SIZE, WRITE_SIZE;
separate `rocprofv3 --kernel-trace --pmc X` runs), keyed to the sha256 of the libhvr_hip.so that ran (bench.py refuses a file
whose key names another build).

    python tools/make_traffic_json.py <fetch.db> <write.db> <out.json> [groups] [bf16|f16x2]

groups = G: the passes ran `tools/rel_bench.py --groups G` (hvr_relation_fwd_grouped: G windows per call); the per-launch figures are per CALL.

FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 counts a 128-byte request as 64 bytes, MI355X_MICROARCH.md HBM
section).  Infinity-Cache hits are part of FETCH_SIZE: this is L2-miss traffic, an upper bound on HBM traffic.
"""
import hashlib
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MINE = ('relation_', 'tile_kernel', 'pc_tile')


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute('select kernel_name, value from counters_collection where counter_name = ?', (counter,)).fetchall()
    agg = {}
    for name, v in rows:
        if not any(m in name for m in MINE):
            continue
        key = re.sub(r'\(.*', '', name).replace('void ', '').replace('hvr::', '')
        d = agg.setdefault(key, [0, 0.0])
        d[0] += 1
        d[1] += v
    return agg


def main(fetch_db, write_db, out, groups=1, dtype='bf16'):
    groups = int(groups)
    es = 2 if dtype == 'f16x2' else 1   # bytes per element relative to the two-byte formats (split half: 4-byte elements)
    f, w = per_kernel(fetch_db, 'FETCH_SIZE'), per_kernel(write_db, 'WRITE_SIZE')
    calls = max(d[0] for k, d in f.items() if 'relation_scores' in k)
    kernels, total = {}, 0.0
    for k in sorted(set(f) | set(w)):
        fe = f.get(k, [0, 0.0])[1] * 1024 * 2.0 / calls
        wr = w.get(k, [0, 0.0])[1] * 1024 / calls
        kernels[k] = dict(launches_per_call=round(f.get(k, w.get(k))[0] / calls, 2), fetch_MB=round(fe / 1e6, 1), write_MB=round(wr / 1e6, 1))
        total += fe + wr
    lib = os.path.join(ROOT, 'hvrnet_amd', 'libhvr_hip.so')
    json.dump(dict(source='rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python tools/rel_bench.py --iters 5 --groups %d [--dtype]` '
                          '(%d window(s) per call, Mq = Mk = 4500, D = 1024, %s); tools/collect_profiles.sh' % (groups, groups, dtype), groups=groups, dtype=dtype,
                   note='FETCH_SIZE doubled (gfx950: a 128-byte request counts 64 bytes). Infinity-Cache hits are included: L2-miss traffic, '
                        'an upper bound on HBM traffic.',
                   lib_sha16=hashlib.sha256(open(lib, 'rb').read()).hexdigest()[:16], relation_calls=calls, per_full_relation_call=kernels,
                   traffic_bytes_per_launch=int(total), traffic_bytes_per_window=int(total / groups), algorithmic_bytes_per_launch=36900000 * groups * es, two_pass_floor_bytes=138700000 * groups * es,
                   two_pass_floor_note='per window, two-byte elements (split half: twice these): Q, K, V, O once (36.9 MB) + P~ written and read once (2 x 41.5 MB) + V^T written and read once (2 x 9.4 MB)'),
              open(out, 'w'), indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:6])

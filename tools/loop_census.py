"""Census of the MFMA loops in the compiler's assembly (hvrnet_amd/csrc/build/*.s, kept by build.sh's -save-temps): for every kernel,
the innermost loop holding the most MFMAs -- MFMAs per iteration, branches inside it (a taken branch costs a wave ~100 cycles:
profiles/r04_kloop_probe.txt), barriers, VALU / SALU instructions between the MFMAs.

    python tools/loop_census.py [file.s ...] [--min-mfma 16] [--filter substring]
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def loops_of(body):
    """(label, start, end) of every backward branch's span"""
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    spans = []
    for i, l in enumerate(body):
        m = re.match(r'\s*s_c?branch\S*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            spans.append((labels[m.group(1)], i))
    return spans


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    min_mfma = int(sys.argv[sys.argv.index('--min-mfma') + 1]) if '--min-mfma' in sys.argv else 16
    flt = sys.argv[sys.argv.index('--filter') + 1] if '--filter' in sys.argv else None
    if '--min-mfma' in sys.argv: args.remove(str(min_mfma))
    if flt: args.remove(flt)
    files = args or sorted(glob.glob(os.path.join(ROOT, 'hvrnet_amd/csrc/build/*gfx950.s')))
    rows = []
    for f in files:
        s = open(f).read()
        for m in re.finditer(r'^(_Z\w+):\s*; @', s, re.M):
            name = m.group(1)
            end = s.find('s_endpgm', m.end())
            body = s[m.end():end].split('\n')
            best = None
            for a, b in loops_of(body):
                seg = body[a:b + 1]
                n = sum(1 for l in seg if l.strip().startswith('v_mfma'))
                # innermost: no other backward span strictly inside with as many MFMAs is handled by picking the smallest span among equals
                if n >= min_mfma and (best is None or n > best[0] or (n == best[0] and b - a < best[2] - best[1])):
                    best = (n, a, b)
            if not best:
                continue
            n, a, b = best
            seg = [l.strip() for l in body[a:b + 1]]
            br = sum(1 for l in seg if re.match(r's_c?branch', l)) - 1
            ex = sum(1 for l in seg if re.match(r's_(and|or|andn2)_saveexec', l))
            bar = sum(1 for l in seg if l.startswith('s_barrier'))
            valu = sum(1 for l in seg if re.match(r'v_(?!mfma)', l))
            salu = sum(1 for l in seg if re.match(r's_(?!waitcnt|barrier|c?branch|nop|setprio|sleep)', l))
            vmem = sum(1 for l in seg if re.match(r'(buffer|global)_(load|store)', l))
            ds = sum(1 for l in seg if l.startswith('ds_'))
            rows.append((os.path.basename(f).split('-')[0], name, n, br, ex, bar, valu, salu, vmem, ds))
    dm = demangle([r[1] for r in rows])
    print('%-10s %5s %4s %4s %4s %5s %5s %5s %4s  kernel' % ('file', 'mfma', 'br', 'exec', 'bar', 'valu', 'salu', 'vmem', 'ds'))
    for r in rows:
        nm = dm.get(r[1], r[1])
        if flt and flt not in nm:
            continue
        print('%-10s %5d %4d %4d %4d %5d %5d %5d %4d  %s' % (r[0], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], nm[:150]))


if __name__ == '__main__':
    main()

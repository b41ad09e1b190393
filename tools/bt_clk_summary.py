"""Summarises tools/bt_clk.py's output (the BTCLK / ABCLK lines of the second, warm call) into the per-phase timeline kept under
profiles/:   python tools/bt_clk_summary.py bt_clk_g4.txt bt_clk_g1.txt ...
Ticks are 10 ns (100 MHz wall clock), counted from each workgroup's own start."""
import re
import statistics as st
import sys


def med(v):
    return st.median(v) if v else float('nan')


def summarise(path):
    txt = open(path).read()
    warm = txt.split('CALL 1')[-1]
    bt = re.findall(r'BTCLK wg (\d+) xcc (\d+) t0 (\d+) vt (\d+) \| tile0 loop (\d+)\.\.(\d+) epi (\d+) \| tile1 loop (\d+)\.\.(\d+) epi (\d+) \| '
                    r'tile2 loop (\d+)\.\.(\d+) epi (\d+) \| tile3 loop (\d+)\.\.(\d+) epi (\d+) \| drained (\d+)', warm)
    ab = re.findall(r'ABCLK wg (\d+) t0 (\d+) prologue (\d+) first-dma (\d+) loop (\d+) epilogue (\d+)', warm)
    print('%s: %d scores workgroups, %d apply workgroups parsed' % (path, len(bt), len(ab)))
    if bt:
        rows = [[int(x) for x in r] for r in bt]
        t0 = [r[2] for r in rows]
        print('  scores launch: start skew over the chip %d ticks; V^T / first-DMA prologue up to the first loop: median %d' % (max(t0) - min(t0), med([r[3] for r in rows])))
        for t in range(4):
            have = [r for r in rows if r[4 + 3 * t + 1] > 0]
            if not have:
                continue
            ls, le, ee = [r[4 + 3 * t] for r in have], [r[5 + 3 * t] for r in have], [r[6 + 3 * t] for r in have]
            gap = ''
            if t > 0:
                gap = ', %d ticks after the previous tile\'s epilogue ended' % med([r[4 + 3 * t] - r[6 + 3 * (t - 1)] for r in have])
            print('  tile %d (%3d workgroups): loop starts at %5d, K loop %4d ticks, epilogue (up to the issue of its last store) %4d ticks%s'
                  % (t, len(have), med(ls), med([b - a for a, b in zip(ls, le)]), med([c - b for b, c in zip(le, ee)]), gap))
        dr = [r[16] for r in rows]
        print('  everything drained (vmcnt 0): median %d, latest %d ticks after the workgroup\'s own start' % (med(dr), max(dr)))
    if ab:
        rows = [[int(x) for x in r] for r in ab]
        print('  apply launch: prologue (block statistics -> shift table) median %d, first DMA wait %d, K loop %d, epilogue %d ticks'
              % (med([r[2] for r in rows]), med([r[3] for r in rows]), med([r[4] for r in rows]), med([r[5] for r in rows])))


for p in sys.argv[1:]:
    summarise(p)

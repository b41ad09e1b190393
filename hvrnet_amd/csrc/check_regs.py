"""Build-time check on the tile kernels' register allocation (run by build.sh on the compiler's
-Rpass-analysis=kernel-resource-usage remarks).

Kernels that use loads the compiler does not track (the pipelined relation apply pass: its block weights; the
pipelined residual kernels that prefetch the residual tile, i.e. every RESPRE kernel with NS > 2 except the 144x256
shapes) keep the destination registers untouched until the hand-counted wait has passed.  A register spill in such a
kernel could store a destination before its load has landed, so the build fails if one of them spills.
"""
import re
import sys


def main(*paths):
    text = ''.join(open(p).read() for p in paths)
    blocks = re.split(r'remark: (?:\S+ )?Function Name: ', text)[1:]   # (with -save-temps the remarks carry file:line:col)
    bad, seen = [], 0
    for b in blocks:
        name = b.split()[0]
        if name.startswith(('_ZN3hvr17expand_res_kernel', '_ZN3hvr19expand_split_kernel')):   # expand.hip: sized for two workgroups per CU, a spill means it no longer fits
            seen += 1
            spill = int(re.search(r'VGPRs Spill: (\d+)', b).group(1))
            scratch = int(re.search(r'ScratchSize \[bytes/lane\]: (\d+)', b).group(1))
            if spill or scratch:
                bad.append('%s: %d VGPRs spilled, %d bytes of scratch' % (name, spill, scratch))
            continue
        if name.startswith(('_ZN3hvr19mc_nms_', '_ZN3hvr15nms_', '_ZN3hvr16nms_', '_ZN3hvr17rpn_', '_ZN3hvr25relation_scores_bt', '_ZN3hvr14pc_tile_kernel', '_ZN3hvr15big_tile_kernel')):
            # the serial read-out / proposal kernels and the one-round scores kernel: scratch traffic inside their dependency
            # chains (the greedy sweep's prefetched IoU rows, the 176 accumulators) is a silent slowdown -- fail the build
            seen += 1
            spill = int(re.search(r'VGPRs Spill: (\d+)', b).group(1))
            scratch = int(re.search(r'ScratchSize \[bytes/lane\]: (\d+)', b).group(1))
            # The producer / consumer kernels with the fattest compute waves park values in scratch OUTSIDE the K loop only
            # (checked in the .s, `-save-temps`: no scratch access between the loop's barriers): pc_tile_kernel<4, LINEAR, 3>
            # a few epilogue addresses before the loop (<= 32 B), pc_tile_kernel<2, APPLY, 4> the folded accumulators between
            # the last block and the epilogue's LDS staging (<= 256 B).  Anything beyond that fails the build.
            allow = 32 if re.match(r'_ZN3hvr14pc_tile_kernelI(?:t|NS_5f16_tE)Li4ELi0E', name) else (256 if re.match(r'_ZN3hvr14pc_tile_kernelI(?:t|NS_5f16_tE)Li2ELi2E', name) else 0)
            if scratch > allow or (spill and not allow):
                bad.append('%s: %d VGPRs spilled, %d bytes of scratch' % (name, spill, scratch))
            continue
        m = re.match(r'_ZN3hvr11tile_kernelI(?:t|f|NS_5f16_tE|NS_6f16s_tE)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELb([01])ELi(\d+)E', name)
        if not m:
            continue
        wm, wn, fm, fn, epi, glds, respre, ns = [int(x) for x in m.groups()]
        spill = int(re.search(r'VGPRs Spill: (\d+)', b).group(1))
        bn = wn * fn * 16
        # the split-half K-step keeps three fragment sets on the B side: its pipelined shapes are sized to fit without scratch traffic
        # in the loop (the 6-wave 144 x 256 ring does not and is never chosen for split operands, gemm.hip: choose_tile)
        split_pipe = 'f16s_t' in name and ns > 2 and not (wm == 3 and fn == 8)
        untracked = epi == 2 or (ns > 2 and respre == 1 and bn != 256) or split_pipe
        if untracked:
            seen += 1
            if spill:
                bad.append('%s: %d VGPRs spilled' % (name, spill))
    if not seen:
        sys.exit('check_regs: no kernel recognised in the remarks (format change?)')
    if bad:
        sys.exit('check_regs: these kernels must not spill:\n  ' + '\n  '.join(bad))
    print('check_regs: %d kernels checked (untracked loads / serial sweeps / big accumulator tiles), none spills' % seen)


if __name__ == '__main__':
    main(*sys.argv[1:])

"""Build-time lint on the device assembly (run by build.sh on the `-save-temps` .s of the kernels with hand-counted waits).

The pipelined kernels read their LDS fragments (and a few global values) through inline asm and wait for them with hand-placed
`s_waitcnt`s.  The compiler knows nothing of that latency: to it the asm's output register is valid the moment the asm statement
has executed, so anything IT adds on such a register -- a loop-header copy that resolves a renamed value, a spill, a register
shuffle -- may sit in front of the wait and read the register before the data has landed.  That is invisible in tests that run a
kernel alone (the data has long arrived after a dozen MFMAs) and corrupts tiles as soon as a co-resident workgroup of another
launch keeps the LDS busy (round 3: the split-half K loop's `B_hi` rename, two windows in flight).

Rule checked, per kernel: a VGPR written by an inline-asm `ds_read*` / `global_load*` must not be read or overwritten by any
instruction before an `s_waitcnt` that covers the load (lgkmcnt / vmcnt, counted in issue order: `cnt(N)` leaves the N youngest
outstanding).  Scanned: every loop (a backward branch and the text it spans), along the text order, the forward branches inside it
and its back edges -- a back edge carries the pending loads to the loop header, which is where the register allocator parks the
copies of a value renamed across iterations.
"""
import re
import sys

_REG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')
_LABEL = re.compile(r'^([.\w$]+):')


def _regs(text):
    out = set()
    for a, b, c in _REG.findall(text):
        if c:
            out.add(int(c))
        else:
            out.update(range(int(a), int(b) + 1))
    return out


def _parse(lines):
    """-> list of (mnemonic, operand string, in_asm), labels {name: index}"""
    ins, labels, in_asm = [], {}, False
    for ln in lines:
        s = ln.split(';')[0].strip() if not ln.lstrip().startswith(';;#') else ln.strip()
        if s.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if s.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not s:
            continue
        m = _LABEL.match(s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if s.startswith('.'):
            continue
        parts = s.split(None, 1)
        ins.append((parts[0], parts[1] if len(parts) > 1 else '', in_asm))
    return ins, labels


def _is_lgkm(mn):
    return mn.startswith('ds_') or mn.startswith('s_load') or mn.startswith('s_buffer_load')


def _is_vm(mn):
    return mn.startswith(('global_load', 'global_store', 'buffer_load', 'buffer_store', 'flat_load', 'flat_store', 'scratch_', 'global_atomic', 'buffer_atomic'))


def _scan(name, ins, labels, lo, hi, problems):
    """Paths inside ins[lo .. hi] (a loop: hi is a backward branch to lo), from a clean state at every instruction in text order,
    through forward branches and the back edges; a path that leaves the range ends (the exits' own waits are the caller's
    business: an exit taken with loads pending is how a K loop with zero steps would look, which never runs)."""
    seen, visits = set(), {}
    work = [(lo, (), ())]
    while work:
        pos, lg, vm = work.pop()
        while lo <= pos <= hi:
            mn, ops, in_asm = ins[pos]
            if mn == 's_waitcnt':
                m = re.search(r'lgkmcnt\((\d+)\)', ops)
                if m:
                    n = int(m.group(1))
                    lg = lg[len(lg) - n:] if n else ()
                m = re.search(r'vmcnt\((\d+)\)', ops)
                if m:
                    n = int(m.group(1))
                    vm = vm[len(vm) - n:] if n else ()
            else:
                # LDS fragments: any touch.  Untracked global loads (block weights, the residual prefetch): their waits sit behind
                # branches that are correlated with the branch that issued them (no load on the path that skips the wait), which
                # a scan of the text cannot know -- there only what the COMPILER adds is checked: plain copies and spills
                is_copy = mn.startswith(('v_mov_b', 'v_accvgpr_write', 'scratch_store', 'v_accvgpr_mov'))
                pending = set().union(*lg) if lg else set()
                if vm and is_copy:
                    pending = pending.union(*vm)
                if pending:
                    touched = _regs(ops) & pending
                    if touched:
                        problems.append('%s: `%s %s` touches v%s before the wait that covers its inline-asm load' % (name, mn, ops, sorted(touched)))
                        lg = tuple(q - touched for q in lg)   # reported once, then treated as landed
                        vm = tuple(q - touched for q in vm)
                if _is_lgkm(mn):
                    lg = lg + (frozenset(_regs(ops.split(',')[0])) if (in_asm and mn.startswith('ds_read')) else frozenset(),)
                elif _is_vm(mn):
                    vm = vm + (frozenset(_regs(ops.split(',')[0])) if (in_asm and 'load' in mn and ' lds' not in ops) else frozenset(),)
            while lg and not lg[0]:   # the oldest entries only matter while a pending load sits behind them
                lg = lg[1:]
            while vm and not vm[0]:
                vm = vm[1:]
            vm = vm[-48:]
            if mn.startswith('s_cbranch') or mn == 's_branch':
                tgt = labels.get(ops.strip())
                if tgt is not None and lo <= tgt <= hi and (lg or vm) and (tgt, lg, vm) not in seen and visits.get(tgt, 0) < 16:
                    seen.add((tgt, lg, vm))
                    visits[tgt] = visits.get(tgt, 0) + 1   # (bounded: a loop body full of conditional loads has many queue states)
                    work.append((tgt, lg, vm))
            if mn in ('s_branch', 's_endpgm'):
                lg, vm = (), ()   # no fall-through: the text below is reached by branches only, scanned on from a clean state
            pos += 1
            if not (lg or vm):
                if (pos, (), ()) in seen:
                    break
                seen.add((pos, (), ()))


def check_kernel(name, lines):
    ins, labels = _parse(lines)
    problems = []
    loops = {}
    for j, (mn, ops, _) in enumerate(ins):
        if mn.startswith('s_cbranch') or mn == 's_branch':
            i = labels.get(ops.strip())
            if i is not None and i <= j:
                loops[i] = max(loops.get(i, j), j)
    for lo, hi in sorted(loops.items()):
        if any(a and (m.startswith('ds_read') or 'load' in m) for m, _, a in ins[lo:hi + 1]):
            _scan(name, ins, labels, lo, hi, problems)
    return problems


def main(*paths):
    problems, kernels = [], 0
    for path in paths:
        lines = open(path).read().split('\n')
        starts = [i for i, ln in enumerate(lines) if re.match(r'^_Z\w+:', ln)]
        for k, s in enumerate(starts):
            end = next((i for i in range(s, len(lines)) if lines[i].strip().startswith('s_endpgm')), len(lines) - 1)
            nxt = starts[k + 1] if k + 1 < len(starts) else len(lines)
            body = lines[s + 1:min(nxt, len(lines))]
            if not any(';;#ASMSTART' in ln for ln in body):
                continue
            kernels += 1
            problems += check_kernel(lines[s].split(':')[0], body)
    if problems:
        uniq = list(dict.fromkeys(problems))
        sys.exit('check_asm_waits: registers of inline-asm loads used before their wait:\n  ' + '\n  '.join(uniq[:40])
                 + ('\n  ... %d more' % (len(uniq) - 40) if len(uniq) > 40 else ''))
    print('check_asm_waits: %d kernels with inline-asm loads scanned, no early use' % kernels)


if __name__ == '__main__':
    main(*sys.argv[1:])

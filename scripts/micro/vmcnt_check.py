"""Static check of an AMDGPU ISA listing (hipcc -S): does any instruction read a VGPR that a still-outstanding buffer/global load
writes?  Models vmcnt the way the hardware counts it for loads (in-order return): every buffer_load / global_load pushes its
destination registers, `s_waitcnt vmcnt(n)` retires all but the n youngest.  Linear scan (all uniformly-branched blocks taken: the
M = 3 path of the head weight gradient); back edges are handled by scanning the loop body twice.  Also lists v_pk_* instructions
whose 64-bit source pairs mix a freshly loaded register with an older one (op_sel forms).
usage: python vmcnt_check.py file.s kernel_symbol"""
import re
import sys


def regs(tok):
    m = re.fullmatch(r'v(\d+)', tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def main(path, sym):
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith(sym + ':'))
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    body = lines[start:end]
    pending = []
    hazards = []
    nload = nwait = nvalu = npk = 0
    for rep in range(2):        # second pass: state carried over the loop's back edge
        for ln, l in enumerate(body):
            l = l.split(';')[0].strip()
            if not l or l.endswith(':') or l.startswith('.'):
                continue
            op, _, rest = l.partition(' ')
            toks = [t.strip() for t in re.split(r',\s*(?![^\[]*\])', rest)] if rest else []
            toks = [t.split(' ')[0] for t in toks]
            if op.startswith('buffer_load') or op.startswith('global_load'):
                src = set().union(*[regs(t) for t in toks[1:]]) if len(toks) > 1 else set()
                for d in pending:
                    if d & src:
                        hazards.append((rep, ln, l, 'address register still loading'))
                pending.append(regs(toks[0]))
                nload += rep == 0
                continue
            if op == 's_waitcnt':
                m = re.search(r'vmcnt\((\d+)\)', rest)
                if m:
                    n = int(m.group(1))
                    pending = pending[len(pending) - n:] if n else []
                    nwait += rep == 0
                continue
            if op.startswith('buffer_store') or op.startswith('global_store') or op.startswith('ds_'):
                src = set().union(*[regs(t) for t in toks]) if toks else set()
            elif op.startswith('v_'):
                src = set().union(*[regs(t) for t in toks[1:]]) if len(toks) > 1 else set()
                if toks:
                    src |= regs(toks[0])        # the destination too: a VALU write over a register an outstanding load will write
                                                # (WAW) is as wrong as a read of it
                nvalu += rep == 0
                npk += (rep == 0 and op.startswith('v_pk_'))
            else:
                continue
            for d in pending:
                if d & src:
                    hazards.append((rep, ln, l, 'reads v%s while its load is outstanding (pending loads: %d)' % (sorted(d & src), len(pending))))
    print('%s: %d loads, %d vmcnt waits, %d VALU (%d packed)' % (sym, nload, nwait, nvalu, npk))
    if not hazards:
        print('  no instruction reads a register with an outstanding load')
    for h in hazards[:20]:
        print('  pass %d line %d: %s   <- %s' % h)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])

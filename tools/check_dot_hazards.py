"""Scan gfx950 assembly for readers / writers of a v_dot2c result that follow it too closely.

The dot2c instructions are emitted through inline assembly, which the compiler's hazard
recogniser cannot see into: a different VALU or memory instruction that reads the result within
3 wait states, or a VALU write of the same register within 4, would get no s_nop.  This walks the
control flow (fall-through and branch targets) from every dot2c and reports such pairs.
usage: check_dot_hazards.py file.s [mnemonic-prefix]"""
import re
import sys


def parse(path):
    instrs, labels = [], {}
    for line in open(path):
        s = line.split(';')[0].rstrip()
        if not s:
            continue
        m = re.match(r'^([.\w$]+):', s)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        t = s.strip()
        if not line.startswith('\t') or t.startswith('.'):
            continue
        instrs.append(t)
    return instrs, labels


def regs(tok):
    m = re.fullmatch(r'v(\d+)', tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def operands(ins):
    parts = ins.split(None, 1)
    if len(parts) < 2:
        return []
    return [p.strip() for p in re.split(r',\s*(?![^\[]*\])', parts[1])]


def main():
    path = sys.argv[1]
    prefix = sys.argv[2] if len(sys.argv) > 2 else 'v_dot2c'
    instrs, labels = parse(path)
    found = 0
    for i, ins in enumerate(instrs):
        if not ins.startswith(prefix):
            continue
        dst = regs(operands(ins)[0])
        frontier = [(i + 1, 0)]
        seen = set()
        while frontier:
            j, waited = frontier.pop()
            if j >= len(instrs) or (j, waited) in seen or waited >= 4:
                continue
            seen.add((j, waited))
            nxt = instrs[j]
            op = nxt.split()[0]
            ops = operands(nxt)
            if op == 's_nop':
                frontier.append((j + 1, waited + int(ops[0], 0) + 1))
                continue
            if op == 's_endpgm':
                continue
            is_valu = op.startswith('v_')
            is_mem = op.startswith(('ds_', 'global_', 'buffer_', 'flat_', 'scratch_'))
            if is_valu or is_mem:
                same = op.startswith(prefix)
                written = regs(ops[0]) if is_valu and ops else set()
                read_ops = ops[1:] if is_valu else ops
                read = set().union(*[regs(o) for o in read_ops]) if read_ops else set()
                if same:
                    # the accumulator (first operand, tied) may be read back to back
                    read = set().union(*[regs(o) for o in ops[1:]])
                    written = set()
                if waited < 3 and dst & read:
                    found += 1
                    print(f'RAW after {waited} wait states: [{i}] {ins}  ->  [{j}] {nxt}')
                if waited < 4 and is_valu and not same and dst & written:
                    found += 1
                    print(f'WAW after {waited} wait states: [{i}] {ins}  ->  [{j}] {nxt}')
            if op == 's_branch':
                frontier.append((labels[ops[0]], waited + 1))
                continue
            if op.startswith('s_cbranch'):
                frontier.append((labels[ops[0]], waited + 1))
            frontier.append((j + 1, waited + 1))
    print(f'{found} hazard(s) in {path}')
    return 1 if found else 0


if __name__ == '__main__':
    sys.exit(main())

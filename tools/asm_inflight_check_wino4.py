#!/usr/bin/env python3
"""Static check of conv_wino4_kernel's hand-waited chunk loop in a hipcc assembly listing (csrc/conv_wino4.hip; the sibling of
asm_inflight_check_wino.py for the eight-wave kernel).  The loop fetches its weight fragments (global_load_lds_dwordx4) and patch
pieces (buffer_load_dwordx4 ... lds) by LDS-DMA from inline asm and waits for them with counted `s_waitcnt vmcnt(N)`; the compiler
sees neither.  Checked: (1) the counted-wait protocol -- on every path the waits in front of the four fragment groups' first reads leave
8 / 4 / 4 / 10 of the 12 / 8 / 8 / 20 operations in flight; (2) should a load with a VGPR destination
ever come back into the loop (the kernel's first version had them, and the compiler's spills: scratch_load), nothing may touch its
destination while it is in flight.  Model: vector memory operations return in issue order; `s_waitcnt vmcnt(N)` leaves the N youngest
in flight.  The chunk loop (the function's depth-2 loop) is
walked twice around along EVERY combination of its forward conditional branches (the cursor bookkeeping at the end of a step: new
tile, new source, ragged chunk -- rare paths, some with compiler spill reloads, which drain the queue); no instruction on any path
may read or write a VGPR that is the destination of a load still in flight.  Also reported per path: the counted waits with the
number of operations in flight in front of each (the step's protocol is 8 / 4 / 4 / 10 of 12 / 8 / 8 / 20).
    python tools/asm_inflight_check_wino4.py kernel.s"""
import itertools
import re
import sys

lines = open(sys.argv[1]).read().split('\n')


def regs(tok):
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


def find_loops():
    out = []
    starts = [n for n, ln in enumerate(lines) if re.match(r'^_Z\w*conv_wino4_kernel\w*:', ln)]
    for st in starts:
        end = next(n for n in range(st, len(lines)) if lines[n].strip() == 's_endpgm')
        hdr = next((n for n in range(st, end) if 'Inner Loop Header: Depth=2' in lines[n]), None)
        if hdr is None:
            raise SystemExit(f'{lines[st]} no depth-2 loop')
        while not re.match(r'^\.LBB\d+_\d+:', lines[hdr]):
            hdr -= 1   # the label line precedes the comment lines
        name = re.match(r'^(\.LBB\d+_\d+):', lines[hdr]).group(1)
        member = [n for n in range(st, end) if re.match(r'^\.LBB\d+_\d+:', lines[n]) and f'Header={name[2:]} Depth=2' in lines[n]]
        labels = [n for n in range(st, end) if re.match(r'^\.LBB\d+_\d+:', lines[n])]
        last_blk = max([m for m in member if m > hdr] + [hdr])
        last = next(n for n in labels if n > last_blk) - 1
        pre = [m for m in member if m < hdr]   # blocks of the loop the compiler laid out in front of its header (they fall into it)
        out.append((lines[st].rstrip(':').split(':')[0], name, hdr, last, (min(pre), hdr - 1) if pre else None))
    return out


def parse(hdr, last, pre):
    """instructions of the loop in layout order (header .. last block, then the blocks in front of the header, which fall through into
    it): (op, vgprs touched, load destination | None, vmcnt | None, branch target | None, line)"""
    ins, labels = [], {}
    order = list(range(hdr, last + 1)) + (list(range(pre[0], pre[1] + 1)) if pre else [])
    for n in order:
        t = lines[n].strip()
        m = re.match(r'^(\.L\w+):', t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not t or t.startswith((';', '.', '//')):
            continue
        t = t.split(';')[0].strip()
        if not t:
            continue
        op, _, rest = t.partition(' ')
        toks = [x.strip() for x in re.split(r'[,\s]+', rest) if x.strip()]
        touched = set()
        for x in toks:
            touched |= regs(x)
        dest, cnt, tgt = None, None, None
        if re.match(r'(global_load_lds|buffer_load\w* .* lds$)', t) or (re.match(r'(global_load|buffer_load)', op) and t.endswith(' lds')):
            dest = set()             # LDS-DMA: counts in vmcnt, no destination register
        elif re.match(r'(global_load|buffer_load|scratch_load)', op):
            dest = regs(toks[0])
            touched -= dest          # (the address operands are read at issue; the destination is what is in flight)
            touched |= set().union(*[regs(x) for x in toks[1:]]) if len(toks) > 1 else set()
        elif re.match(r'(global_store|buffer_store|scratch_store|global_atomic|buffer_atomic)', op):
            dest = set()             # counts in vmcnt, no destination register
        if op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', t)
            cnt = int(m.group(1)) if m else None
        if op.startswith('s_cbranch') or op == 's_branch':
            tgt = toks[-1]
        ins.append((op, touched, dest, cnt, tgt, n + 1))
    return ins, labels


def walk(ins, labels, loop_label, choices):
    """twice around the loop; choices: iterator of booleans for the forward conditional branches met (True = taken)"""
    inflight, bad, waits, pc, rounds, steps = [], [], [], 0, 0, 0
    ch = iter(choices)
    while rounds < 2 and steps < 200000:
        steps += 1
        if pc >= len(ins):   # fell off the blocks in front of the header: into the header, one more round
            rounds += 1
            pc = 0
            continue
        op, touched, dest, cnt, tgt, lno = ins[pc]
        if cnt is not None:
            waits.append((cnt, len(inflight)))
            inflight = inflight[len(inflight) - cnt:] if cnt < len(inflight) else inflight
            if cnt == 0:
                inflight = []
        busy = set().union(*[d for d in inflight]) if inflight else set()
        hit = (touched | (dest or set())) & busy
        if hit:
            bad.append((lno, op, sorted(hit)[:4]))
        if dest is not None:
            inflight.append(set(dest))
        if tgt is not None:
            if tgt == loop_label:
                if op == 's_branch' or True:   # the back edge (conditional: taken while chunks remain)
                    rounds += 1
                    pc = labels[tgt]
                    continue
            if tgt in labels:   # (a block laid out behind the body may jump back into it: the step bound ends a walk that would cycle)
                taken = True if op == 's_branch' else next(ch, True)
                if taken:
                    pc = labels[tgt]
                    continue
        pc += 1
    return bad, waits


def whole_function(st, end):
    """every path of the whole kernel (the patch registers are in flight across the end of a tile: last step, output exchange,
    epilogue, first step of the next tile): worklist over (instruction, in-flight queue) states, both arms of every conditional branch;
    a state seen before is not walked again.  Returns (hazards, states walked)."""
    ins, labels = parse(st + 1, end, None)
    seen, work, bad = set(), [(0, ())], {}
    while work:
        pc, q = work.pop()
        while pc < len(ins):
            key = (pc, q)
            if key in seen:
                break
            seen.add(key)
            if len(seen) > 4000000:
                raise SystemExit('state space too large')
            op, touched, dest, cnt, tgt, lno = ins[pc]
            if cnt is not None and cnt < len(q):
                q = q[len(q) - cnt:] if cnt else ()
            busy = frozenset().union(*q) if q else frozenset()
            hit = (touched | (dest or set())) & busy
            if hit:
                bad.setdefault(lno, (op, sorted(hit)[:4]))
            if dest is not None:
                q = q + (frozenset(dest),)
                k = next((i for i, d in enumerate(q) if d), len(q))   # operations older than the oldest register in flight only count
                q = q[k:][-63:]                                      # themselves: dropped (and the counter holds 63)
            if op == 's_endpgm':
                break
            if tgt is not None and tgt in labels:
                if op != 's_branch':
                    work.append((pc + 1, q))
                pc = labels[tgt]
                continue
            pc += 1
    return bad, len(seen)


def main():
    total = 0
    if len(sys.argv) > 2 and sys.argv[2] == 'whole':
        starts = [n for n, ln in enumerate(lines) if re.match(r'^_Z\w*conv_wino4_kernel\w*:', ln)]
        for st in starts:
            end = next(n for n in range(st, len(lines)) if lines[n].strip() == 's_endpgm')
            bad, nstate = whole_function(st, end)
            print(f'{lines[st].split(":")[0]}: whole kernel, {nstate} states, in-flight register hazards: {len(bad)}'
                  + (f'   first: {sorted(bad.items())[:4]}' if bad else ''))
            total += len(bad)
        sys.exit(1 if total else 0)
    for fn, label, hdr, last, pre in find_loops():
        ins, labels = parse(hdr, last, pre)
        nbr = sum(1 for op, _, _, _, tgt, _ in ins if tgt and tgt != label and op != 's_branch' and tgt in labels)
        nbr = min(nbr, 10)
        worst, nb, seen = None, 0, set()
        for combo in itertools.product((True, False), repeat=nbr):
            bad, waits = walk(ins, labels, label, combo + (True,) * 64)
            nb = max(nb, len(bad))
            if bad and worst is None:
                worst = bad
            seen.add(tuple(waits))
        common = walk(ins, labels, label, (True,) * 64)[1]
        print(f'{fn}: loop {label} lines {hdr + 1}-{last + 1}, {len(ins)} instructions, {nbr} forward conditional branches, '
              f'{2 ** nbr} paths walked twice')
        print(f'   counted waits on the common path (vmcnt, in flight before): {common[:16]}')
        print(f'   in-flight register hazards: {nb}' + (f'   first: {worst[:4]}' if worst else ''))
        total += nb
    sys.exit(1 if total else 0)


if __name__ == '__main__':
    main()

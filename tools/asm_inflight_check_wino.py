#!/usr/bin/env python3
"""Static check of conv_wino_kernel's hand-waited chunk loop in a hipcc assembly listing (as asm_inflight_check.py, for a loop whose
patch wait / patch loads exist twice, each inside an asm-internal `s_cbranch_scc1 .Lwino_*` skip: one wave group runs the first
pair, the other the second).  For both groups the loop body is walked twice; no instruction may touch a VGPR that is the
destination of a global_load still in flight (loads return in issue order, `s_waitcnt vmcnt(N)` leaves the N youngest in flight).
One exception is reported separately and tolerated: the v_cndmask_b32 of the raw store that the compiler hoists out of the
`if (on)` of the group whose turn it is not -- they READ patch registers in flight, their results feed only the skipped LDS stores.
    python tools/asm_inflight_check_wino.py kernel.s      every conv_wino_kernel instantiation of the listing; the chunk loop is the
        function's depth-2 loop, walked from its header to the back branch and then through the blocks the compiler rotated in front
        of the header (the second store / request and the barrier)
    python tools/asm_inflight_check_wino.py kernel.s first_line last_line [first2 last2 ...]   the same with explicit line ranges"""
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
    """[(function name, [first, last, first2, last2])] of the chunk loops (1-based line numbers)"""
    out = []
    starts = [n for n, ln in enumerate(lines) if re.match(r'^_Z\w*conv_wino_kernel\w*:', ln)]
    for st in starts:
        end = next(n for n in range(st, len(lines)) if lines[n].strip() == 's_endpgm')
        hdr = next((n for n in range(st, end) if 'Inner Loop Header: Depth=2' in lines[n]), None)
        if hdr is None:
            raise SystemExit(f'{lines[st]} no depth-2 loop')
        hdr -= 1   # the label line
        name = re.match(r'^(\.LBB\d+_\d+):', lines[hdr]).group(1)[2:]
        member = [n for n in range(st, end) if re.match(r'^\.LBB\d+_\d+:', lines[n]) and f'Header={name} Depth=2' in lines[n]]
        labels = [n for n in range(st, end) if re.match(r'^\.LBB\d+_\d+:', lines[n])]
        last_blk = max(m for m in member if m > hdr)
        last = next(n for n in labels if n > last_blk) - 1
        pre = [m for m in member if m < hdr]
        rng = [hdr + 1, last + 1] + ([min(pre) + 1, hdr] if pre else [])
        out.append((lines[st].rstrip(':'), rng))
    return out


def check(rng):
    body, where_ = [], []
    for a_, b_ in zip(rng[0::2], rng[1::2]):
        body += lines[a_ - 1:b_]
        where_ += list(range(a_, b_ + 1))
    # the asm-internal skips, in order of appearance: (branch line index, label line index)
    skips = []
    for n, ln in enumerate(body):
        m = re.search(r's_cbranch_scc1\s+(\.Lwino_[wg]\d+)', ln)
        if m:
            lab = m.group(1) + ':'
            end = next(k for k in range(n, len(body)) if body[k].strip().startswith(lab))
            skips.append((n, end))
    print(f'{len(skips)} asm-internal skips in the loop body (expected 4 per sub-step: wait, loads, wait, loads)')
    total = 0
    for group, active in (('early (waves 4-7)', {i for i in range(len(skips)) if i % 4 < 2}), ('late (waves 0-3)', {i for i in range(len(skips)) if i % 4 >= 2})):
        dead = set()
        for i, (a, b) in enumerate(skips):
            if i not in active:
                dead |= set(range(a, b + 1))
        inflight, bad, loads, waits, spec = [], 0, 0, [], 0
        for rep in range(2):
            for n, ln in enumerate(body):
                if n in dead:
                    continue
                t = ln.strip()
                if not t or t.startswith((';', '.', '//')) or t.endswith(':'):
                    continue
                t = t.split(';')[0].strip()
                op, _, rest = t.partition(' ')
                toks = [x.strip() for x in re.split(r'[,\s]+', rest) if x.strip()]
                if op == 's_waitcnt':
                    m = re.search(r'vmcnt\((\d+)\)', t)
                    if m:
                        k = int(m.group(1))
                        if rep:
                            waits.append((k, len(inflight)))
                        inflight = inflight[len(inflight) - k:] if k < len(inflight) else inflight
                        if k == 0:
                            inflight = []
                    continue
                used = set()
                for x in toks:
                    used |= regs(x)
                wr = regs(toks[0]) if toks else set()
                for dest, where in inflight:
                    if used & dest:
                        if op.startswith('v_cndmask_b32') and not (wr & dest):
                            spec += 1
                            continue
                        bad += 1
                        if bad <= 10:
                            print(f'{group} pass {rep}: line {where_[n]}: `{t}` touches v{sorted(used & dest)} of the load issued at line {where}')
                if op.startswith(('global_load', 'scratch_load', 'buffer_load')):
                    inflight.append((regs(toks[0]), where_[n]))
                    loads += rep
        print(f'{group}: {loads} loads per iteration, waits (vmcnt, in flight before) {waits}, in-flight register hazards: {bad} (+ {spec} speculated selects reading patch registers in flight)')
        total += bad
    return total


if len(sys.argv) > 2:
    sys.exit(1 if check([int(v) for v in sys.argv[2:]]) else 0)
worst = 0
for fn, rng in find_loops():
    print(f'{fn}: loop body lines {rng}')
    worst += check(rng)
sys.exit(1 if worst else 0)

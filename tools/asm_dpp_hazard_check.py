#!/usr/bin/env python3
"""hipcc cannot see inside inline asm: check by hand that no inline-asm DPP instruction reads (as its DPP-shifted src0) a VGPR
that one of the two preceding instructions wrote with the VALU (gfx9: "VALU writes VGPR -> DPP reads that VGPR" needs 2 wait
states; s_nop N counts N+1).     python tools/asm_dpp_hazard_check.py kernel.s"""
import re
import sys

lines = [ln.split(';')[0].strip() for ln in open(sys.argv[1])]
ins = [ln for ln in lines if ln and not ln.startswith(('.', ';')) and not ln.endswith(':')]


def regs(tok):
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


bad = n = 0
for i, t in enumerate(ins):
    if not t.startswith('v_add_f32_dpp'):
        continue
    n += 1
    ops = [x.strip() for x in re.split(r'[,\s]+', t.partition(' ')[2]) if x.strip()]
    src0 = regs(ops[1])
    states = 0
    j = i - 1
    while j >= 0 and states < 2:
        p = ins[j]
        op = p.split()[0]
        if op == 's_nop':
            states += int(p.split()[1]) + 1
        else:
            if op.startswith('v_') and not op.startswith('v_cmp') and not op.startswith('v_mfma'):
                dst = regs(re.split(r'[,\s]+', p.partition(' ')[2])[0])
                if dst & src0:
                    bad += 1
                    print(f'hazard: `{p}` then (within 2 states) `{t}`')
            states += 1
        j -= 1
print(f'{n} DPP adds checked, {bad} hazards')

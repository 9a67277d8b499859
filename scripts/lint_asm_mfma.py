"""Static check of k_cnet1w's inline-asm MFMAs (cnet1w_sh.hip): hipcc pads nothing around an asm statement, so the two hazards an
MFMA with VGPR operands has towards compiler-generated code are checked on the ISA itself:
  (1) VALU write of a VGPR -> MFMA reads it as srcA / srcB / srcC: 2 wait states.  The operands come from LDS reads and from
      epilogue code several slots earlier; a register copy the compiler might insert right in front of the asm would be a violation.
  (2) MFMA result -> any other reader / writer: 12 wait states (8-pass MFMA).  Readers go through c1_settle (s_nop 12) in the source;
      here: no non-MFMA instruction touches an asm MFMA's destination within 12 states behind it (s_nop N counts N + 1).
usage: lint_asm_mfma.py file.s   (hipcc -save-temps output); exit status 1 on a violation."""
import re, sys

def regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    if m: return {int(m.group(1))}
    return set()

def parse(line):
    line = line.split(';')[0].strip()
    if not line or line.startswith('.') or line.endswith(':'): return None
    parts = line.split(None, 1)
    op = parts[0]
    ops = [t.strip() for t in parts[1].split(',')] if len(parts) > 1 else []
    return op, ops

def main(path):
    lines = open(path).read().splitlines()
    ins = []          # (op, operands, in_asm)
    in_asm = False
    for ln in lines:
        if '#ASMSTART' in ln: in_asm = True; continue
        if '#ASMEND' in ln: in_asm = False; continue
        p = parse(ln)
        if p: ins.append((p[0], p[1], in_asm))
    bad = 0; n_asm = 0
    for i, (op, ops, a) in enumerate(ins):
        if not (a and op.startswith('v_mfma') and ops and ops[0].startswith('v')): continue
        n_asm += 1
        dst = regs(ops[0]); src = set()
        for t in ops[1:4]: src |= regs(t)
        # (1) two states back
        states = 0; j = i - 1
        while j >= 0 and states < 2:
            o, oo, _ = ins[j]
            if o == 's_nop': states += int(oo[0]) + 1
            else:
                states += 1
                if o.startswith('v_') and not o.startswith('v_mfma') and oo and (regs(oo[0]) & src):
                    print(f"VALU write -> asm MFMA operand: {o} {', '.join(oo)}  ->  {op} {', '.join(ops)}"); bad += 1
            j -= 1
        # (2) twelve states ahead
        states = 0; j = i + 1
        while j < len(ins) and states < 12:
            o, oo, _ = ins[j]
            if o == 's_nop': states += int(oo[0]) + 1; j += 1; continue
            states += 1
            touched = set()
            for t in oo: touched |= regs(t)
            if (touched & dst) and not o.startswith('v_mfma'):
                print(f"asm MFMA result touched {states} states later: {op} {ops[0]}  ->  {o} {', '.join(oo)}"); bad += 1
            j += 1
    print(f"{n_asm} inline-asm MFMAs with VGPR destinations checked, {bad} violation(s)")
    return 1 if bad else 0

if __name__ == '__main__':
    sys.exit(main(sys.argv[1]))

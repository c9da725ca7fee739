"""Finds packed (VOP3P) instructions whose destination pair overlaps a source pair while op_sel / op_sel_hi CROSS the halves of that
source -- e.g.  v_pk_add_f32 v[0:1], v[18:19], v[0:1] op_sel:[0,1] op_sel_hi:[1,0]   (lo = a.lo + b.HI, hi = a.hi + b.LO, b == dst).

Why (profiles/r04_hazard.txt): on gfx950 that instruction is not safe.  With other waves active on the SIMD the low result of lanes
48-63 can be computed from the NEW high half (the instruction's own high result) instead of the old one.  It is what made some
instruction orders of the bf16x6 attention kernel produce wrong O for 16 of 32 queries (round 3): hipcc's SLP
vectoriser emits it for the `f0 * a0 + f1 * a1` pairs of the (O, m, l) LDS merge; the MFMA code was innocent.  Pinned by an
ISA-level bisect (tools/dev/isa_bisect.py: with every packed op of the merge replaced by scalar ops EXCEPT one of the two in-place
crossed v_pk_add_f32, the failure is back; with only those two replaced it is gone).

    python tools/dev/isa_pk_inplace_audit.py file.s [...]         exit status 1 if any such instruction exists
    python tools/dev/isa_pk_inplace_audit.py --strict file.s [...] ANY in-place packed fp32 instruction (destination pair == a source
                                                                  pair), crossed or not, is an error

--strict (round 5): what triggers the wrong low half is not established below the instruction form (profiles/r04_hazard.txt section 5,
profiles/r05_erratum_pk_inplace.md), so every source that does not write packed fp32 arithmetic BY HAND is built without the SLP
vectoriser and must contain no in-place packed fp32 instruction at all; attention.hip (explicit v_pk_fma / v_pk_mul / v_pk_add, never
with op_sel on the overwritten operand) is held to the crossed-halves rule.
"""
import re
import sys

PK = re.compile(r'^\s*(v_pk_\w+)\s+v\[(\d+):(\d+)\],\s*(.*)$')
SRC = re.compile(r'v\[(\d+):(\d+)\]')


def audit(path, strict=False):
    out, kern = [], None
    for ln in open(path, errors='replace'):
        s = ln.split(';')[0].rstrip()
        m = re.match(r'^([_A-Za-z][\w.$]*):\s*$', s)
        if m and not s.startswith('.L'):
            kern = m.group(1)
            continue
        m = PK.match(s)
        if not m:
            continue
        op, d0, d1, rest = m.group(1), int(m.group(2)), int(m.group(3)), m.group(4)
        sel = re.search(r'op_sel:\[([\d,]+)\]', rest)
        selh = re.search(r'op_sel_hi:\[([\d,]+)\]', rest)
        srcs = [(int(a), int(b)) for a, b in SRC.findall(rest.split('op_sel')[0])]
        n = len(rest.split('op_sel')[0].split(','))
        lo = [int(x) for x in sel.group(1).split(',')] if sel else [0] * 3
        hi = [int(x) for x in selh.group(1).split(',')] if selh else [1] * 3
        # operand positions of the VGPR-pair sources among the instruction's sources
        toks = [t.strip() for t in rest.split('op_sel')[0].rstrip(', ').split(',')]
        for pos, t in enumerate(toks):
            mm = SRC.fullmatch(t)
            if not mm:
                continue
            a0, a1 = int(mm.group(1)), int(mm.group(2))
            if (a0, a1) != (d0, d1):
                continue                                   # (partial overlaps do not occur with aligned pairs)
            l = lo[pos] if pos < len(lo) else 0
            h = hi[pos] if pos < len(hi) else 1
            if l == 1 or h == 0:                           # low result reads the high half and / or high result reads the low half
                out.append((path, kern, s.strip()))
            elif strict and op.endswith('_f32'):           # in place at all
                out.append((path, kern, s.strip()))
    return out


def main():
    args = sys.argv[1:]
    strict = '--strict' in args
    bad = []
    for p in args:
        if p != '--strict':
            bad += audit(p, strict)
    for path, kern, txt in bad[:200]:
        print('%s: %s: %s' % (path, kern, txt))
    print('%d in-place packed instructions %s' % (len(bad), '(strict: any in-place packed fp32)' if strict else 'with crossed halves'))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())

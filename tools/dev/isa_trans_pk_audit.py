"""Finds packed-fp32 (VOP3P: v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 ...) instructions that read a VGPR whose most recent writer
is a transcendental op (v_exp_f32, v_log_f32, v_rcp_f32, v_rsq_f32, v_sqrt_f32, v_sin_f32, v_cos_f32 and their f16 forms).

Why: on gfx950 that pair is not safe (profiles/r04_hazard.txt): a transcendental result that the quarter-rate trans pipe is still
producing is interlocked for ordinary VALU consumers but NOT for a VOP3P consumer -- the packed multiply of the bf16x6 attention
kernel's accumulator rescale read a stale alpha in lanes 16-31 / 48-63 whenever hipcc scheduled it close behind the v_exp_f32.  The
product kernels therefore hand every transcendental result to packed arithmetic through an ordinary VALU copy (`vcopy()` in
csrc/common.h), and tests/test_host.py runs this audit over every kernel of the library.

    python tools/dev/isa_trans_pk_audit.py file.s [more.s ...]      (hipcc -S --cuda-device-only output)
Exit status 1 and one line per finding if any pair exists.  The search walks each kernel's instruction stream backwards from the
VOP3P instruction over fall-through and branch predecessors (up to --depth instructions)."""
import argparse
import re
import sys

REG = re.compile(r'\bv(?:\[(\d+):(\d+)\]|(\d+)\b)')
TRANS = ('v_exp_', 'v_log_', 'v_rcp_', 'v_rsq_', 'v_sqrt_', 'v_sin_', 'v_cos_')


def vregs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def kernels(path):
    cur, name = None, None
    for ln in open(path, errors='replace'):
        s = ln.split(';')[0].rstrip()
        if not s.strip():
            continue
        if re.match(r'^[_A-Za-z][\w.$]*:\s*$', s) and not s.startswith('.L'):
            if cur:
                yield name, cur
            name, cur = s.strip()[:-1], []
            continue
        if cur is None:
            continue
        t = s.strip()
        if t.startswith('.L') and t.endswith(':'):
            cur.append(('label', t[:-1], t))
            continue
        if t.startswith('.'):
            continue
        op, _, rest = t.partition(' ')
        cur.append((op, [x.strip() for x in rest.split(',')] if rest else [], t))
        if op == 's_endpgm':
            yield name, cur
            cur, name = None, None
    if cur:
        yield name, cur


def writes(op, ops):
    if op == 'label' or not ops or op.startswith(('s_', 'global_store', 'buffer_store', 'flat_store', 'ds_write', 'ds_store', 'v_cmp')):
        return set()
    return vregs(ops[0])


def audit(path, depth):
    found = []
    for name, ins in kernels(path):
        labels = {o[1]: i for i, o in enumerate(ins) if o[0] == 'label'}
        preds = {}
        for i, (op, ops, _) in enumerate(ins):
            if op in ('s_branch',) or op.startswith('s_cbranch'):
                tgt = ops[-1] if ops else None
                if tgt in labels:
                    preds.setdefault(labels[tgt], []).append(i)
        for i, (op, ops, txt) in enumerate(ins):
            if not op.startswith('v_pk_') or len(ops) < 2:
                continue
            for r in vregs(' '.join(ops[1:])):
                # backward search for the last writer of r on every path
                stack, seen = [(i - 1, 0)], set()
                while stack:
                    j, d = stack.pop()
                    while j >= 0 and d <= depth and j not in seen:
                        seen.add(j)
                        jop, jops, jtxt = ins[j]
                        if jop == 'label':
                            for pj in preds.get(j, []):
                                stack.append((pj, d))
                        elif jop == 's_branch':
                            break                       # fall-through does not reach past an unconditional branch
                        elif r in writes(jop, jops):
                            if jop.startswith(TRANS):
                                found.append((path, name, d, jtxt, txt))
                            break
                        j -= 1
                        d += 1
    return found


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('asm', nargs='+')
    ap.add_argument('--depth', type=int, default=400)
    a = ap.parse_args()
    bad = []
    for p in a.asm:
        bad += audit(p, a.depth)
    uniq = sorted(set(bad))
    for path, name, d, prod, cons in uniq:
        print('%s: %s: %d instructions apart:  %s   ->   %s' % (path, name, d, prod, cons))
    print('%d transcendental -> packed-fp32 pairs' % len(uniq))
    return 1 if uniq else 0


if __name__ == '__main__':
    sys.exit(main())

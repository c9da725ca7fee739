"""MFMA hazard audit of a gfx950 kernel's ISA (VERDICT r3 next #1a).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o k.s file.hip
    python tools/dev/isa_hazard_audit.py k.s attn_x6_d32_kernel [--window 24]

For every v_mfma in the named kernel it walks forward through the straight-line instruction stream (falling through labels, stopping
at the kernel end; loop back edges are followed once) and reports, per hazard class, the SMALLEST number of wait states the
compiler left between the MFMA and a later instruction that touches the same registers.  Wait states are counted the way LLVM's
GCNHazardRecognizer counts them: every issued instruction is one, `s_nop N` is N + 1.

Classes (register overlap between the MFMA and the later instruction):
  dst->valu_r   MFMA vDst  -> VALU / VMEM / LDS / export READ      (RAW; gfx950 table for an 8-pass XDL op: 12, 16-pass: 20)
  dst->valu_w   MFMA vDst  -> VALU WRITE                           (WAW; same table)
  dst->mfma_ab  MFMA vDst  -> MFMA SrcA / SrcB read                (12 / 20)
  dst->mfma_c   MFMA vDst  -> MFMA SrcC read, other vDst / overlap (10 / 18; the exact same vDst back to back needs 0-2)
  ab->valu_w    MFMA SrcA/B -> VALU / VMEM-return WRITE            (WAR: NOT in any table -- operands are read "at issue")
  c->valu_w     MFMA SrcC  -> VALU WRITE                           (WAR)
  valu_w->ab    VALU WRITE -> MFMA SrcA / SrcB read                (RAW, distance looking BACK from the MFMA)
  valu_w->c     VALU WRITE -> MFMA SrcC read
"""
import argparse
import re
import sys
from collections import defaultdict

REG = re.compile(r'\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)')


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        kind = m.group(1)
        if m.group(2) is not None:
            out.update((kind, i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((kind, int(m.group(4))))
    return out


def parse(path, kernel):
    lines = open(path).read().split('\n')
    start = None
    for i, ln in enumerate(lines):
        if ln.startswith('_Z') and kernel in ln and ln.rstrip().split(';')[0].strip().endswith(':'):
            start = i
            break
    if start is None:
        sys.exit('kernel %s not found' % kernel)
    ins, labels = [], {}
    for ln in lines[start + 1:]:
        s = ln.split(';')[0].strip()
        if not s:
            continue
        if s.endswith(':'):
            labels[s[:-1]] = len(ins)
            continue
        if s.startswith('.'):
            continue
        op, _, rest = s.partition(' ')
        ops = [x.strip() for x in rest.split(',')] if rest else []
        ins.append((op, ops, s))
        if op == 's_endpgm':
            break
    return ins, labels


def classify(op, ops):
    """returns (reads, writes, kind)"""
    if op.startswith('v_mfma') or op.startswith('v_smfma'):
        return regs(' '.join(ops[1:])), regs(ops[0]), 'mfma'
    if op.startswith(('s_', '.')):
        return set(), set(), 'salu'
    if op.startswith(('global_load', 'buffer_load', 'flat_load', 'scratch_load')):
        return regs(' '.join(ops[1:])), regs(ops[0]), 'vmem_load'
    if op.startswith(('global_store', 'buffer_store', 'flat_store', 'scratch_store', 'global_atomic', 'buffer_atomic')):
        return regs(' '.join(ops)), set(), 'vmem_store'
    if op.startswith('ds_'):
        if 'read' in op or 'load' in op or 'bpermute' in op or 'permute' in op or 'swizzle' in op:
            return regs(' '.join(ops[1:])), regs(ops[0]), 'lds'
        return regs(' '.join(ops)), set(), 'lds'
    if op.startswith('v_'):
        if op.startswith('v_cmp') and not op.startswith('v_cmpx'):
            return regs(' '.join(ops)), set(), 'valu'
        if op.startswith(('v_readfirstlane', 'v_readlane')):
            return regs(' '.join(ops[1:])), set(), 'valu'
        if op.startswith('v_accvgpr'):
            return regs(' '.join(ops[1:])), regs(ops[0]), 'valu'
        return regs(' '.join(ops[1:])), regs(ops[0]), 'valu'
    return set(), set(), 'other'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('asm')
    ap.add_argument('kernel')
    ap.add_argument('--window', type=int, default=24)
    ap.add_argument('--show', type=int, default=3, help='print this many closest instances per class')
    a = ap.parse_args()
    ins, labels = parse(a.asm, a.kernel)
    info = [classify(op, ops) for op, ops, _ in ins]
    n = len(ins)

    def succ(i):
        op, ops, _ = ins[i]
        out = []
        if op == 's_endpgm':
            return out
        if op == 's_branch':
            return [labels[ops[0]]] if ops[0] in labels else []
        if op.startswith('s_cbranch') and ops and ops[-1] in labels:
            out.append(labels[ops[-1]])
        if i + 1 < n:
            out.append(i + 1)
        return out

    def ws(i):
        op, ops, _ = ins[i]
        if op == 's_nop':
            return int(ops[0], 0) + 1
        return 1

    found = defaultdict(list)     # class -> [(distance, mfma idx, other idx)]
    for i, (op, ops, _) in enumerate(ins):
        if info[i][2] != 'mfma':
            continue
        dst = regs(ops[0])
        sa, sb = regs(ops[1]), regs(ops[2])
        sc = regs(ops[3]) if len(ops) > 3 else set()
        # forward walk (DFS over the CFG, bounded by the window)
        seen = {}
        stack = [(s, 0) for s in succ(i)]
        while stack:
            jx, d = stack.pop()
            if d > a.window or (jx in seen and seen[jx] <= d):
                continue
            seen[jx] = d
            r, w, kind = info[jx]
            jop, jops, _ = ins[jx]
            if kind == 'mfma':
                ja, jb = regs(jops[1]), regs(jops[2])
                jc = regs(jops[3]) if len(jops) > 3 else set()
                jd = regs(jops[0])
                if dst & (ja | jb):
                    found['dst->mfma_ab'].append((d, i, jx))
                if dst & jc:
                    found['dst->mfma_c(same vDst)' if jc == dst and jd == dst else 'dst->mfma_c(overlap)'].append((d, i, jx))
                if (sa | sb) & jd and jd != dst:
                    found['ab->mfma_dst_w'].append((d, i, jx))
            else:
                if dst & r and kind in ('valu', 'vmem_store', 'vmem_load', 'lds'):
                    found['dst->%s_r' % ('valu' if kind == 'valu' else 'mem')].append((d, i, jx))
                if dst & w:
                    found['dst->%s_w' % ('valu' if kind == 'valu' else 'memret')].append((d, i, jx))
                if (sa | sb) & w:
                    found['ab->%s_w' % ('valu' if kind == 'valu' else 'memret')].append((d, i, jx))
                if sc & w and not (dst & w):
                    found['c->valu_w'].append((d, i, jx))
            for s in succ(jx):
                stack.append((s, d + ws(jx)))
        # backward: closest VALU write feeding this MFMA (straight-line only)
        d = 0
        for jx in range(i - 1, max(-1, i - 64), -1):
            r, w, kind = info[jx]
            if ins[jx][0].startswith(('s_cbranch', 's_branch')):
                break
            if kind == 'valu':
                if w & (sa | sb):
                    found['valu_w->ab'].append((d, i, jx))
                    break
                if w & sc:
                    found['valu_w->c'].append((d, i, jx))
                    break
            d += ws(jx)
            if d > a.window:
                break
    nm = sum(1 for x in info if x[2] == 'mfma')
    print('%s: %d instructions, %d MFMAs' % (a.kernel, n, nm))
    for cls in sorted(found):
        lst = sorted(found[cls])
        print('  %-24s min %2d wait states   (%d instances within %d)' % (cls, lst[0][0], len(lst), a.window))
        shown = set()
        for d, i, jx in lst:
            if len(shown) >= a.show:
                break
            key = (ins[i][2], ins[jx][2])
            if key in shown:
                continue
            shown.add(key)
            print('       %2d : [%d] %s' % (d, i, ins[i][2]))
            print('            [%d] %s' % (jx, ins[jx][2]))


if __name__ == '__main__':
    main()

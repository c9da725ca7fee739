"""ISA-level bisect of the failing instruction order of the bf16x6 attention kernel (VERDICT r3 next #1a).

tools/dev/x6_hazard.hip is compiled to assembly; the text of ONE kernel (default: hz_kernel<1, 2>, which fails deterministically on
MI355X) is edited -- the schedule and the register allocation stay exactly what hipcc produced, only the named change is made --
and every edit is assembled into its own code object that tools/dev/x6_hazard_mod loads with hipModuleLoad:

    python tools/dev/isa_bisect.py [--kernel hz_kernelILi1ELi2E] [--out tools/dev/bisect]
    tools/dev/x6_hazard_mod tools/dev/bisect/*.co

Edits (each applied to the unmodified kernel):
  e0_control            nothing (must fail like the compiled-in kernel)
  e1_no_nops            the inline-asm `s_nop 15` pairs removed everywhere (same order, no idle time)
  e2_nops_loop_only     kept in the main loop, removed from the peeled / tail copies
  e3_nops_tail_only     removed from the main loop, kept elsewhere
  e4_nop_before_mfma    s_nop 7 in front of every v_mfma
  e5_nop_after_mfma     s_nop 7 behind every v_mfma
  e6_vmcnt0_before_mfma s_waitcnt vmcnt(0) in front of every v_mfma (no load in flight when a matrix instruction issues)
  e7_vmcnt0_before_load s_waitcnt vmcnt(0) in front of every global_load (loads strictly one at a time)
  e8_nop_before_load    s_nop 7 in front of every global_load
  e9_pk_mul_as_two      every v_pk_mul_f32 ... op_sel_hi:[1,0] as two v_mul_f32
  e10_nop_after_valu_w  s_nop 1 behind every v_perm_b32 (the last writers of the P planes)
  e11_lgkm0_top         s_waitcnt vmcnt(0) lgkmcnt(0) at every loop header
"""
import argparse
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = '/opt/rocm/lib/llvm/bin'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--kernel', default='hz_kernelILi1ELi2E')
    ap.add_argument('--out', default=os.path.join(ROOT, 'tools', 'dev', 'bisect'))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    src = os.path.join(a.out, 'hz.s')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-S',
                           '--cuda-device-only', '-o', src, os.path.join(ROOT, 'tools', 'dev', 'x6_hazard.hip')],
                          stderr=subprocess.DEVNULL)
    lines = open(src).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and a.kernel in l and l.rstrip().endswith(':') or
                 (l.startswith('_Z') and a.kernel in l and ': ;' in l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    body = lines[start + 1:end + 1]

    def is_op(l, prefix):
        return l.strip().startswith(prefix)

    def asm_nop_blocks(b):
        """index ranges (first, last) of the lines between ;;#ASMSTART and ;;#ASMEND"""
        out, i = [], 0
        while i < len(b):
            if '#ASMSTART' in b[i]:
                j = i
                while '#ASMEND' not in b[j]:
                    j += 1
                out.append((i, j))
                i = j
            i += 1
        return out

    loop_lo = next(i for i, l in enumerate(body) if 'Inner Loop Header' in l)
    loop_hi = next(i for i in range(loop_lo, len(body)) if is_op(body[i], 's_cbranch_scc1'))

    def drop_nops(b, where):
        out = []
        for i, l in enumerate(b):
            inside = loop_lo <= i <= loop_hi
            if l.strip().startswith('s_nop 15') and any(lo < i < hi for lo, hi in blocks):
                if where == 'all' or (where == 'loop' and inside) or (where == 'tail' and not inside):
                    continue
            out.append(l)
        return out
    blocks = asm_nop_blocks(body)

    def before(b, prefix, ins):
        out = []
        for l in b:
            if is_op(l, prefix):
                out.append('\t' + ins)
            out.append(l)
        return out

    def after(b, prefix, ins):
        out = []
        for l in b:
            out.append(l)
            if is_op(l, prefix):
                out.append('\t' + ins)
        return out

    def pk_as_two(b):
        out = []
        pat = re.compile(r'v_pk_mul_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel_hi:\[1,0\]')
        for l in b:
            m = pat.search(l)
            if m:
                d0, d1, s0, s1, t0, _ = (int(x) for x in m.groups())
                out.append('\tv_mul_f32_e32 v%d, v%d, v%d' % (d0, t0, s0))
                out.append('\tv_mul_f32_e32 v%d, v%d, v%d' % (d1, t0, s1))
            else:
                out.append(l)
        return out

    def at_loop_tops(b, ins):
        out = []
        for l in b:
            out.append(l)
            if 'Loop Header' in l:
                out.append('\t' + ins)
        return out

    def ds_split(b):
        out = []
        pat = re.compile(r'ds_write2st64_b32 (v\d+), (v\d+), (v\d+)(.*)')
        for l in b:
            m = pat.search(l)
            if m:
                a, d0, d1, rest = m.groups()
                o0 = re.search(r'offset0:(\d+)', rest)
                o1 = re.search(r'offset1:(\d+)', rest)
                o0 = int(o0.group(1)) if o0 else 0
                o1 = int(o1.group(1)) if o1 else 0
                out.append('\tds_write_b32 %s, %s offset:%d' % (a, d0, o0 * 256))
                out.append('\tds_write_b32 %s, %s offset:%d' % (a, d1, o1 * 256))
            else:
                out.append(l)
        return out

    def before_first_ds_write(b, ins):
        """in front of the first ds_write2st64 of every run of them"""
        out, prev = [], False
        for l in b:
            is_w = is_op(l, 'ds_write2st64')
            if is_w and not prev:
                out.append('\t' + ins)
            out.append(l)
            if l.strip() and not l.strip().startswith(';'):
                prev = is_w
        return out

    def after_barrier(b, fn):
        """applies fn to the lines behind the first s_barrier (the LDS merge of the four key quarters)"""
        i = next(k for k, l in enumerate(b) if is_op(l, 's_barrier'))
        return b[:i + 1] + fn(b[i + 1:])

    PK = re.compile(r'v_pk_(mul|add)_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\](.*)')

    def scalarize(l):
        """a v_pk_{mul,add}_f32 as two scalar ops through two scratch registers (v100, v101: dead behind the key loop)"""
        m = PK.search(l)
        if not m:
            return None
        op = m.group(1)
        d0, d1, a0, a1, b0, b1 = (int(m.group(i)) for i in range(2, 8))
        rest = m.group(8)
        sel = re.search(r'op_sel:\[(\d),(\d)\]', rest)
        selh = re.search(r'op_sel_hi:\[(\d),(\d)\]', rest)
        sa, sb = (int(sel.group(1)), int(sel.group(2))) if sel else (0, 0)
        ha, hb = (int(selh.group(1)), int(selh.group(2))) if selh else (1, 1)
        a, b = (a0, a1), (b0, b1)
        return ['\tv_%s_f32_e32 v100, v%d, v%d' % (op, a[sa], b[sb]), '\tv_%s_f32_e32 v101, v%d, v%d' % (op, a[ha], b[hb]),
                '\tv_mov_b32_e32 v%d, v100' % d0, '\tv_mov_b32_e32 v%d, v101' % d1]

    def scalarize_set(t, which):
        out, k = [], 0
        for l in t:
            r = scalarize(l) if is_op(l, 'v_pk_') else None
            if r is not None:
                if which is None or k in which:
                    out += r
                else:
                    out.append(l)
                k += 1
            else:
                out.append(l)
        return out

    npk = sum(1 for l in body[next(k for k, l in enumerate(body) if is_op(l, 's_barrier')):] if is_op(l, 'v_pk_') and PK.search(l))

    edits = {
        'e0_control': body,
        'e1_no_nops': drop_nops(body, 'all'),
        'e2_nops_loop_only': drop_nops(body, 'tail'),
        'e3_nops_tail_only': drop_nops(body, 'loop'),
        'e4_nop_before_mfma': before(body, 'v_mfma', 's_nop 7'),
        'e5_nop_after_mfma': after(body, 'v_mfma', 's_nop 7'),
        'e6_vmcnt0_before_mfma': before(body, 'v_mfma', 's_waitcnt vmcnt(0)'),
        'e7_vmcnt0_before_load': before(body, 'global_load', 's_waitcnt vmcnt(0)'),
        'e8_nop_before_load': before(body, 'global_load', 's_nop 7'),
        'e9_pk_mul_as_two': pk_as_two(body),
        'e10_nop_after_valu_w': after(body, 'v_perm_b32', 's_nop 1'),
        'e11_lgkm0_top': at_loop_tops(body, 's_waitcnt vmcnt(0) lgkmcnt(0)'),
        # the LDS merge at the kernel's end: the accumulators go to LDS by ds_write2st64_b32, the first one exactly 12 wait states
        # behind the last MFMA of the chain (hipcc's minimum for XDL write -> LDS read)
        'e12_nop8_before_ds_write_run': before_first_ds_write(body, 's_nop 7'),
        'e13_nop32_before_ds_write_run': before_first_ds_write(body, 's_nop 15\n\ts_nop 15'),
        'e14_ds_write2_as_two': ds_split(body),
        'e15_lgkm0_before_ds_write_run': before_first_ds_write(body, 's_waitcnt lgkmcnt(0)'),
        # behind the barrier: the merge itself (identical code in passing and failing builds)
        'e18_nop64_after_barrier': after_barrier(body, lambda t: ['\ts_nop 15', '\ts_nop 15', '\ts_nop 15', '\ts_nop 15'] + t),
        'e19_nop_before_pk': after_barrier(body, lambda t: before(t, 'v_pk_', 's_nop 3')),
        'e20_nop_after_lgkm_wait': after_barrier(body, lambda t: after(t, 's_waitcnt lgkmcnt', 's_nop 3')),
        'e21_lgkm0_after_ds_read': after_barrier(body, lambda t: after(t, 'ds_read2st64', 's_waitcnt lgkmcnt(0)')),
        'e22_nop_before_exp': after_barrier(body, lambda t: before(t, 'v_exp_f32', 's_nop 3')),
        'e23_nop_after_exp': after_barrier(body, lambda t: after(t, 'v_exp_f32', 's_nop 7')),
        'e24_nop_before_store': after_barrier(body, lambda t: before(t, 'global_store', 's_nop 7')),
        'e26_nop_before_ds_read': after_barrier(body, lambda t: before(t, 'ds_read2st64', 's_nop 7')),
        'e27_nop_after_pk': after_barrier(body, lambda t: after(t, 'v_pk_', 's_nop 7')),
        'e28_nop2_before_ds_read': after_barrier(body, lambda t: before(t, 'ds_read2st64', 's_nop 1')),
        'e29_nop4_before_ds_read': after_barrier(body, lambda t: before(t, 'ds_read2st64', 's_nop 3')),
        'e25_second_barrier': after_barrier(body, lambda t: ['\ts_waitcnt vmcnt(0) lgkmcnt(0)', '\ts_barrier'] + t),
        'e16_nop2_before_ds_write_run': before_first_ds_write(body, 's_nop 1'),
        'e17_nop4_before_ds_write_run': before_first_ds_write(body, 's_nop 3'),
    }
    edits['s_all_pk_scalar'] = after_barrier(body, lambda t: scalarize_set(t, None))
    for k in range(npk):
        edits['s_only_pk%02d_scalar' % k] = after_barrier(body, lambda t, k=k: scalarize_set(t, {k}))
        edits['s_all_but_pk%02d_scalar' % k] = after_barrier(body, lambda t, k=k: scalarize_set(t, set(range(npk)) - {k}))
    for name, b in edits.items():
        path = os.path.join(a.out, name + '.s')
        with open(path, 'w') as f:
            f.write('\n'.join(lines[:start + 1] + b + lines[end + 1:]))
        obj = path[:-2] + '.o'
        subprocess.check_call([LLVM + '/clang', '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', path, '-o', obj])
        subprocess.check_call([LLVM + '/ld.lld', '-shared', obj, '-o', path[:-2] + '.co'])
        os.remove(obj)
        print(name, len(b) - len(body), 'lines added')
    with open(os.path.join(a.out, 'KERNEL'), 'w') as f:
        f.write(lines[start].split(':')[0])


if __name__ == '__main__':
    main()

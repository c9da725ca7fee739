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
    }
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

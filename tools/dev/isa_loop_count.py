"""Instruction mix of a kernel's largest loop in a gfx950 assembly file (hipcc -S): VALU / MFMA counts per iteration.
   python tools/dev/isa_loop_count.py /tmp/asm/attention_x6.s attn_x6_d32_kernel"""
import collections
import re
import sys


def loop_mix(path, kernel):
    s = open(path).read()
    m = re.search(r'^(\w*%s\w*):' % re.escape(kernel), s, re.M)
    a = m.start()
    body = s[a:s.index('.Lfunc_end', a)].splitlines()
    labels = {}
    for i, l in enumerate(body):
        mm = re.match(r'^(\.LBB\d+_\d+):', l)
        if mm:
            labels[mm.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        mm = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            loops.append((labels[mm.group(1)], i))
    lo, hi = max(loops, key=lambda t: t[1] - t[0])
    cnt = collections.Counter()
    for l in body[lo:hi]:
        l = l.strip()
        if not l or l.startswith(('.', ';', '//')) or l.endswith(':'):
            continue
        cnt[l.split()[0]] += 1
    return m.group(1), cnt


if __name__ == '__main__':
    name, cnt = loop_mix(sys.argv[1], sys.argv[2])
    valu = sum(v for k, v in cnt.items() if k.startswith('v_') and not k.startswith('v_mfma'))
    print(name, ': VALU', valu, ' MFMA', sum(v for k, v in cnt.items() if k.startswith('v_mfma')),
          ' LDS', sum(v for k, v in cnt.items() if k.startswith('ds_')), ' VMEM', sum(v for k, v in cnt.items() if k.startswith(('global_', 'buffer_'))))
    for k, v in cnt.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 25):
        print('  %-28s %d' % (k, v))

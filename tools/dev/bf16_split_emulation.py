"""Would a bf16-split MFMA path for the GEMM/conv set keep parity?    CPU emulation through the
oracle: every F.linear / F.conv2d (groups == 1) is replaced by the sum of the bf16 partial products a split-operand
`v_mfma_f32_32x32x16_bf16` kernel would accumulate in fp32 -- products of bf16 numbers are exact in fp32, so the only
difference from the real kernel is summation order.  Attention (QK^T, PV) stays fp32, as it would on the device.

    python tools/dev/bf16_split_emulation.py [case] [terms: 3 | 6]

3 terms: a = hi + lo (2 x bf16), products hi*hi + hi*lo + lo*hi          (16 mantissa bits)
6 terms: a = hi + mid + lo (3 x bf16), products of total order <= 2       (24 mantissa bits)
Prints max |logit - reference golden| at stride 4 and the mask pixels that differ, per frame, teacher-forced."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'aot-benchmark_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from common import case_clip, load_case, run_teacher_forced, synth_model_state, unpack_gapmask  # noqa: E402
from oracle.aot_oracle import OracleEngine, OracleModel  # noqa: E402


def split(x, n):
    parts, r = [], x
    for _ in range(n):
        p = r.to(torch.bfloat16).to(torch.float32)
        parts.append(p)
        r = r - p
    return parts


def patched(fn, terms):
    n = 2 if terms == 3 else 3

    def f(x, w, *a, **k):
        groups = k.get('groups', a[4] if len(a) > 4 else 1)
        if groups != 1 or x.dtype != torch.float32:
            return fn(x, w, *a, **k)
        bias = k.pop('bias', None)
        if a:
            bias, a = a[0], a[1:]
        xs, ws = split(x, n), split(w, n)
        out = None
        for i in range(n):
            for j in range(n):
                if i + j > n - 1:
                    continue
                y = fn(xs[i], ws[j], None, *a, **k)
                out = y if out is None else out + y
        if bias is not None:
            out = out + (bias.view(1, -1, 1, 1) if out.dim() == 4 else bias)
        return out
    return f


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else 'c2_r50_aotl'
    terms = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    c, g = load_case(case)
    _, _, sd = synth_model_state(c['model'])
    frames, mask, objs, out_size = case_clip(c, g=g)
    F.linear, F.conv2d = patched(F.linear, terms), patched(F.conv2d, terms)
    eng = OracleEngine(OracleModel(c['model'], sd))
    res = run_teacher_forced(eng, frames, mask, objs, out_size, g, set(c['keep_logits']))
    no = c['num_obj'] + 1
    tot = hard = 0
    for t, (l4, m) in sorted(res.items()):
        bad = m != g['masks'][t - 1]
        tie = unpack_gapmask(g, t, bad.shape)
        tot += int(bad.sum())
        hard += int((bad & ~tie).sum())
        err = np.abs(l4[:no] - g['logits4_%d' % t]).max() if l4 is not None else float('nan')
        print('%s bf16x%d frame %d: max |dlogit4| %.3g, mask pixels differing %d (outside reference near-ties: %d)'
              % (case, terms, t, err, int(bad.sum()), int((bad & ~tie).sum())), flush=True)
    print('TOTAL %s bf16x%d: %d pixels differ, %d outside near-ties, over %d frames' % (case, terms, tot, hard, len(res)))


if __name__ == '__main__':
    main()

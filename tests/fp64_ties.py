"""The reference's own fp64 run of the whole-clip goldens (tests/golden/<case>_fp64.npz, made by tests/golden/make_fp64_ties.py from
the REAL reference) and the classification of an engine's differing mask pixels on it.  Pure numpy: the GPU parity tests and
bench.py's J&F leg both use it; it is test infrastructure, nothing under aot-benchmark_amd/ imports it."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

_FP64_CACHE = {}


def load_fp64_ties(case):
    """tests/golden/<case>_fp64.npz (make_fp64_ties.py: the REAL reference run in double over a whole-clip golden, teacher-forced
    on the fp32 masks) or None for the cases that have none."""
    if case not in _FP64_CACHE:
        path = os.path.join(GOLD, case + '_fp64.npz')
        _FP64_CACHE[case] = dict(np.load(path)) if os.path.exists(path) else None
    return _FP64_CACHE[case]


FP64_GAP_BINS = (5e-5, 1e-4, 2e-4)


def classify_flips(f64, t, pred, ref):
    """Where does an engine mask differ from the fp32 reference's, measured on the fp64 reference of the same frame?
    Returns counts: 'flips' (all differing pixels), 'ref_undecided' (the reference's own fp32 and fp64 argmax differ there),
    'sides_with_fp64' (of those: the engine's id is the fp64 id), 'gap64<5e-05' / '<0.0001' / '<0.0002' (the others, by the fp64
    top-2 gap; each pixel in its smallest bin), 'swap_top2' (the engine's id is one of the fp64 top two), 'outside' (none of the above)."""
    pred, ref = np.asarray(pred).reshape(-1), np.asarray(ref).reshape(-1)
    bad = np.nonzero(pred != ref)[0]
    out = {'flips': int(bad.size), 'ref_undecided': 0, 'sides_with_fp64': 0, 'swap_top2': 0, 'outside': 0}
    out.update({'gap64<%g' % b: 0 for b in FP64_GAP_BINS})
    if not bad.size:
        return out
    idx, gap = f64['idx_%d' % t], f64['gap_%d' % t]
    top1, top2 = f64['top1_%d' % t], f64['top2_%d' % t]
    pos = {int(i): n for n, i in enumerate(idx)}
    undecided = set(int(i) for i in f64['diff_%d' % t])
    for px in bad:
        px = int(px)
        n = pos.get(px)
        if n is not None and pred[px] in (top1[n], top2[n]):
            out['swap_top2'] += 1
        if px in undecided:
            out['ref_undecided'] += 1
            # the fp64 id of a pixel the reference flips is the one the fp32 reference does NOT have: its fp64 top-1 when stored
            if n is None or pred[px] == top1[n]:
                out['sides_with_fp64'] += 1
            continue
        if n is None:
            out['outside'] += 1
            continue
        for b in FP64_GAP_BINS:
            if gap[n] < b:
                out['gap64<%g' % b] += 1
                break
    return out


def add_counts(total, part):
    for k, v in part.items():
        total[k] = total.get(k, 0) + v
    return total

"""fp64 runs of the REAL reference over the three whole-clip goldens (build container only).

    python tests/golden/make_fp64_ties.py [case ...]        # needs /root/reference; writes tests/golden/<case>_fp64.npz

The argmax ids of a frame are decided, on a handful of pixels, by the fp32 summation order of whoever computes the logits:
the reference itself flips them between fp32 and fp64.  This script makes that statement a fixture.  For each 70-frame golden the
reference network is cast to double (every `.float()` of the reference's label path likewise) and driven over the same clip

  * TEACHER-FORCED on the committed fp32 masks (the state of every frame is the fp32 reference's state, only the arithmetic
    differs): per frame the pixels whose fp64 top-2 logit gap at the output size is below 2e-4 are stored sparsely -- flat pixel
    index, gap (float32), fp64 first and second id -- together with the pixels where the fp64 argmax differs from the fp32
    golden mask (`diff_<t>`).  A test can then evaluate "every flip of the HIP engine lies on a pixel the reference cannot
    decide itself" at any threshold up to 2e-4 and count how many flips side with the fp64 reference;
  * FREE-RUNNING on its own fp64 labels: the number of pixels per frame on which the reference in fp64 differs from the
    reference in fp32 (`free_diff`: the reference's own noise level over a whole clip, feedback included).
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, 'aot-benchmark_amd'))

import refdriver  # noqa: E402
from make_golden import CASES  # noqa: E402
from utils.synth import synth_clip, synth_state_dict  # noqa: E402

WHOLE_CLIPS = ('c2_r50_aotl_70', 'c3b_r50_deaotl_70', 'c3_swinb_deaotl_480_70')
GAP_KEPT = 2e-4


def run_fp64(make_engine, frames, mask, objs, out_size, teacher):
    """tools/demo.py:187-235 in double; `teacher` = fp32 golden masks [T-1,H,W] or None (free-running)."""
    eng = make_engine()
    eng.restart_engine()
    recs = []
    with torch.no_grad():
        eng.add_reference_frame(frames[0], mask, frame_step=0, obj_nums=objs)
        for t in range(1, len(frames)):
            t0 = time.time()
            eng.match_propogate_one_frame(frames[t])
            logit = eng.decode_current_logits(out_size)
            assert logit.dtype == torch.float64
            top = torch.topk(logit[0], 2, dim=0)
            gap = (top[0][0] - top[0][1]).flatten()
            label = torch.argmax(torch.softmax(logit, dim=1), dim=1, keepdim=True).double()
            idx = torch.nonzero(gap < GAP_KEPT).flatten()
            recs.append(dict(mask=label[0, 0].to(torch.uint8).clone(), idx=idx.to(torch.int32).numpy(),
                             gap=gap[idx].float().numpy(), top1=top[1][0].flatten()[idx].to(torch.uint8).numpy(),
                             top2=top[1][1].flatten()[idx].to(torch.uint8).numpy()))
            fb = label if teacher is None else teacher[t - 1].view(1, 1, *out_size).double()
            fb = F.interpolate(fb, size=eng.input_size_2d, mode='nearest')
            eng.update_memory(fb)
            if t in (1, len(frames) - 1):
                print('    frame %d: %.1f s' % (t, time.time() - t0), flush=True)
    return recs


def main():
    only = sys.argv[1:] or WHOLE_CLIPS
    torch.set_num_threads(int(os.environ.get('AOT_GOLDEN_THREADS', os.cpu_count() or 1)))
    to_float = torch.Tensor.float
    for name in only:
        c = CASES[name]
        # weights and clip are generated under the fp32 default (the keyed generators draw different numbers in double) ...
        torch.Tensor.float = to_float
        torch.set_default_dtype(torch.float32)
        gold = np.load(os.path.join(HERE, name + '.npz'))
        gm = torch.from_numpy(gold['masks'])
        net, make_engine, cfg = refdriver.build_reference(c['model'], gap=c.get('gap'))
        net.load_state_dict(synth_state_dict(net.state_dict()))
        frames, mask, objs, out_size = synth_clip(c['clip'], c['frames'], c['in_size'], c['out_size'], c['num_obj'])
        # ... then everything is cast.  The reference casts label maps with .float() (utils/image.py:69-74 and friends) and
        # creates its position / one-hot tensors under the default dtype: in the fp64 run both mean double
        net.double()
        frames = [f.double() for f in frames]
        mask = mask.double()
        torch.Tensor.float = lambda self, *a, **k: self.double()
        torch.set_default_dtype(torch.float64)
        out = {}
        print(name, 'teacher-forced fp64', flush=True)
        recs = run_fp64(make_engine, frames, mask, objs, out_size, gm)
        stats = []
        for t, r in enumerate(recs, start=1):
            diff = torch.nonzero((r['mask'] != gm[t - 1]).flatten()).flatten().to(torch.int32).numpy()
            out['idx_%d' % t], out['gap_%d' % t], out['top1_%d' % t], out['top2_%d' % t] = r['idx'], r['gap'], r['top1'], r['top2']
            out['diff_%d' % t] = diff
            g32 = np.unpackbits(gold['gapmask_%d' % t])[:gm[0].numel()].astype(bool)
            stats.append([len(r['idx']), int((r['gap'] < 5e-5).sum()), len(diff), int(g32.sum()), int((~g32[diff]).sum())])
        out['stats'] = np.array(stats)      # per frame: fp64 ties < 2e-4, < 5e-5, fp32-vs-fp64 argmax flips, fp32 ties < 2e-4, flips off them
        print('  per frame [ties64<2e-4, ties64<5e-5, fp32!=fp64, ties32<2e-4, flips outside ties32]: totals', out['stats'].sum(0).tolist(),
              'max', out['stats'].max(0).tolist(), flush=True)
        print(name, 'free-running fp64', flush=True)
        recs = run_fp64(make_engine, frames, mask, objs, out_size, None)
        out['free_diff'] = np.array([int((r['mask'] != gm[t]).sum()) for t, r in enumerate(recs)])
        print('  free-running fp64 vs fp32 reference, differing pixels per frame:', out['free_diff'].tolist(), flush=True)
        np.savez_compressed(os.path.join(HERE, name + '_fp64.npz'), **out)


if __name__ == '__main__':
    main()

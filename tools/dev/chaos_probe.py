"""How fast does a mask perturbation grow through the mask feedback of the REAL reference (build container only)?

Runs the reference engine free-running twice on the same synthetic clip; the second run has `--flip` pixels of the fed-back
label changed at frame `--at`.  Prints the number of differing mask pixels per frame: a growth factor > 1 per frame means the
clip cannot be reproduced free-running across fp32 summation orders (config 3 with the synthetic weights).
`--scale key_substring=factor` multiplies matching synthetic tensors (calibration experiments before they go into
utils/synth.py).

    python tools/dev/chaos_probe.py swinb_deaotl --size 240 432 --frames 24 --at 6 --flip 2
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'aot-benchmark_amd'), os.path.join(ROOT, 'tests', 'golden')]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('model')
    ap.add_argument('--size', type=int, nargs=2, default=(240, 432))
    ap.add_argument('--frames', type=int, default=24)
    ap.add_argument('--at', type=int, default=6)
    ap.add_argument('--flip', type=int, default=2)
    ap.add_argument('--clip', type=int, default=10)
    ap.add_argument('--gap', type=int, default=5)
    ap.add_argument('--obj', type=int, default=10)
    ap.add_argument('--scale', action='append', default=[])
    ap.add_argument('--f64', action='store_true')
    ap.add_argument('--vs64', action='store_true', help='instead of a perturbation: the same clip in fp32 and in fp64')
    ap.add_argument('--smooth', type=int, default=0, help='box-filter the frames with this (odd) kernel size')
    a = ap.parse_args()
    from utils.synth import synth_clip, synth_state_dict
    import refdriver
    torch.set_num_threads(os.cpu_count())
    net, make_engine, cfg = refdriver.build_reference(a.model, gap=a.gap)
    sd = synth_state_dict(net.state_dict())
    for s in a.scale:
        k, f = s.split('=')
        n = 0
        for key in sd:
            if k in key and sd[key].is_floating_point():
                sd[key] = sd[key] * float(f)
                n += 1
        print('scaled %d tensors matching %r by %s' % (n, k, f))
    net.load_state_dict(sd)
    if a.f64:
        net = net.double()
    H, W = a.size
    frames, mask, objs, out_size = synth_clip(a.clip, a.frames, (H, W), (H, W), a.obj)
    if a.smooth:
        frames = [F.avg_pool2d(F.pad(f, (a.smooth // 2,) * 4, mode='circular'), a.smooth, 1) * a.smooth ** 0.5 for f in frames]
    if a.f64:
        frames = [f.double() for f in frames]
        mask = mask.double()

    def run(perturb):
        nonlocal frames, mask
        if a.vs64 and perturb:
            # the reference casts label maps with .float() (utils/image.py:69-74 and friends): in the fp64 run that means double
            torch.Tensor.float = lambda self, *aa, **kk: self.double()
            torch.set_default_dtype(torch.float64)
            net.double()
            frames = [f.double() for f in frames]
            mask = mask.double()
        eng = make_engine()
        eng.restart_engine()
        masks, gaps, keep = [], [], {}
        with torch.no_grad():
            eng.add_reference_frame(frames[0], mask, frame_step=0, obj_nums=objs)
            for t in range(1, a.frames):
                t0 = time.time()
                eng.match_propogate_one_frame(frames[t])
                logit = eng.decode_current_logits(out_size)
                label = torch.argmax(torch.softmax(logit, dim=1), dim=1, keepdim=True).to(frames[0].dtype)
                top2 = torch.topk(logit[0], 2, dim=0)[0]
                gaps.append(int(((top2[0] - top2[1]) < 2e-4).sum()))
                masks.append(label[0, 0].to(torch.uint8).clone())
                if t == a.at + 1:
                    keep['logit'] = logit.clone()
                    keep['gap'] = (top2[0] - top2[1]).clone()
                    e0 = eng.aot_engines[0]
                    keep['lstt'] = [x.clone() for x in e0.curr_lstt_output[0]]
                    keep['l4'] = e0.pred_id_logits.clone()
                fb = label.clone()
                if perturb and t == a.at and not a.vs64:
                    g = torch.Generator().manual_seed(7)
                    ys = torch.randint(0, H, (a.flip,), generator=g)
                    xs = torch.randint(0, W, (a.flip,), generator=g)
                    for y, x in zip(ys.tolist(), xs.tolist()):
                        fb[0, 0, y, x] = (fb[0, 0, y, x] + 1) % (a.obj + 1)
                fb = F.interpolate(fb, size=eng.input_size_2d, mode='nearest')
                eng.update_memory(fb)
                if t == a.at:
                    keep['id'] = eng.aot_engines[0].curr_id_embs.clone()
                if t == 1:
                    print('  %.1f s per frame' % (time.time() - t0), flush=True)
        return masks, gaps, keep

    m0, gaps, k0 = run(False)
    m1, _, k1 = run(True)
    d = (k0['logit'] - k1['logit'])[0, :a.obj + 1].abs()
    rel = lambda x, y: '%.2e (std %.2f)' % ((x - y).abs().mean(), x.std())
    print('id_emb after the flip:', rel(k0['id'], k1['id']), '| GPM/LSTT outputs:', [rel(x, y) for x, y in zip(k0['lstt'], k1['lstt'])],
          '| stride-4 logits:', rel(k0['l4'], k1['l4']))
    print('frame at+1: |dlogit| max %.3g mean %.3g; logit std %.3g; pixels with top-2 gap < 1e-4/1e-3/1e-2/1e-1: %s' % (
        d.max(), d.mean(), k0['logit'][0, :a.obj + 1].std(), [int((k0['gap'] < x).sum()) for x in (1e-4, 1e-3, 1e-2, 1e-1)]))
    diff = [int((x != y).sum()) for x, y in zip(m0, m1)]
    hist = [torch.bincount(m.flatten().long(), minlength=a.obj + 1).tolist() for m in (m0[0], m0[-1])]
    print('label histogram first / last frame:', hist)
    print('near-tie pixels per frame:', gaps)
    print('differing pixels per frame, fp32 run against fp64 run, both free-running:' if a.vs64 else
          'differing pixels per frame after flipping %d px of the feedback at frame %d:' % (a.flip, a.at))
    print(diff)


if __name__ == '__main__':
    main()

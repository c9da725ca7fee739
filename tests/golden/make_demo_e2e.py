"""Real-image end-to-end golden (build container only): the REAL reference's `Evaluator.evaluating`
(networks/managers/evaluator.py:209-505 == the loop of tools/demo.py:112-255) on the first six 1080p JPEG frames of
datasets/Demo/images/1001_3iEIq5HBY1s with its 44-object first-frame mask -- five object groups through the reference's
AOTInferEngine, MultiRestrictSize (1080x1920 -> 577x1025 at the default long edge 800 * 1.3) + MultiToTensor, its own VOSTest
bookkeeping, DataLoader collation, fusion / feedback code and save_mask call.

    python tests/golden/make_demo_e2e.py            # needs /root/reference; writes tests/golden/demo_1001/{*.jpg, *.png, golden.npz}

What is NOT the reference (as in make_golden.make_evaluator_loop): cv2 is absent -- `cv2.imread` is PIL's JPEG decoder (the GPU test
decodes the same bytes with PIL too), `cv2.resize` is the oracle's restated INTER_CUBIC; the CUDA-only calls are CPU no-ops.  Weights:
keyed synthetic R50-AOTL (utils/synth.py), as everywhere.  The six JPEGs and the label PNG are copied next to the golden: they are
the reference's demo DATA, the inputs of this fixture.  Stored per propagated frame: the mask handed to save_mask (dense ids 0..44),
and the pixels whose fused top-2 class probabilities are within 1e-3 / within 2e-4 of each other (packed bits).
"""
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'aot-benchmark_amd'))

import refdriver  # noqa: E402
from utils.synth import synth_state_dict  # noqa: E402

SEQ = '1001_3iEIq5HBY1s'
FRAMES = 6
MODEL = 'r50_aotl'
OUT = os.path.join(HERE, 'demo_1001')


def main():
    from PIL import Image
    from oracle.aot_oracle import cv2_cubic_resize
    torch.set_num_threads(int(os.environ.get('AOT_GOLDEN_THREADS', os.cpu_count() or 1)))
    src_img = os.path.join(refdriver.REF, 'datasets', 'Demo', 'images', SEQ)
    src_lab = os.path.join(refdriver.REF, 'datasets', 'Demo', 'masks', SEQ)
    names = sorted(os.listdir(src_img))[:FRAMES]
    lab_name = names[0].replace('jpg', 'png')
    os.makedirs(OUT, exist_ok=True)
    for n in names:
        shutil.copyfile(os.path.join(src_img, n), os.path.join(OUT, n))
    shutil.copyfile(os.path.join(src_lab, lab_name), os.path.join(OUT, lab_name))
    for n in os.listdir(OUT):
        os.chmod(os.path.join(OUT, n), 0o644)

    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith('__'):
                raise AttributeError(name)
            return _Any(name)

        def __call__(self, *a, **k):
            return None
    cv2 = _Any('cv2')
    cv2.INTER_CUBIC, cv2.INTER_NEAREST, cv2.INTER_LINEAR = 2, 0, 1
    cv2.setNumThreads = lambda n: None
    cv2.resize = lambda img, dsize=None, interpolation=None, **k: cv2_cubic_resize(img, int(dsize[1]), int(dsize[0]))
    # cv2.imread returns BGR uint8; the dataset swaps to RGB (eval_datasets.py:59-63)
    cv2.imread = lambda path, *a: np.ascontiguousarray(np.array(Image.open(path).convert('RGB'))[:, :, ::-1])
    stubs = {'cv2': cv2, 'torchvision': _Any('torchvision'), 'torchvision.transforms': _Any('torchvision.transforms'),
             'torchvision.transforms.functional': _Any('torchvision.transforms.functional')}

    class _Event:
        def __init__(self, enable_timing=False):
            pass

        def record(self):
            pass

        def elapsed_time(self, other):
            return 1.0
    net, _, cfg = refdriver.build_reference(MODEL)
    net.load_state_dict(synth_state_dict(net.state_dict()))
    refdriver._enter()
    saved_mods = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    saved_cuda = {k: getattr(torch.cuda, k) for k in ('Event', 'empty_cache', 'synchronize', 'max_memory_allocated')}
    saved_tcuda = torch.Tensor.cuda
    written, decoded = [], []
    try:
        torch.cuda.Event = _Event
        torch.cuda.empty_cache = lambda: None
        torch.cuda.synchronize = lambda *a, **k: None
        torch.cuda.max_memory_allocated = lambda device=None: 0
        torch.Tensor.cuda = lambda self, *a, **k: self
        import dataloaders.video_transforms as tr
        import networks.managers.evaluator as ev_mod
        from dataloaders.eval_datasets import VOSTest
        with tempfile.TemporaryDirectory() as tmp:
            os.makedirs(os.path.join(tmp, 'images', SEQ))
            os.makedirs(os.path.join(tmp, 'labels', SEQ))
            for n in names:
                shutil.copyfile(os.path.join(OUT, n), os.path.join(tmp, 'images', SEQ, n))
            shutil.copyfile(os.path.join(OUT, lab_name), os.path.join(tmp, 'labels', SEQ, lab_name))
            chain = [tr.MultiRestrictSize(None, 800 * 1.3, False, [1.0], cfg.MODEL_ALIGN_CORNERS), tr.MultiToTensor()]

            def transform(sample):
                for f in chain:
                    sample = f(sample)
                return sample
            ds = VOSTest(os.path.join(tmp, 'images'), os.path.join(tmp, 'labels'), SEQ, names, [lab_name], transform=transform)
            ecfg = types.SimpleNamespace(**cfg.__dict__)
            for k, v in dict(TEST_WORKERS=0, TEST_DATASET_SPLIT='val', TEST_FLIP=False, TEST_FRAME_LOG=False,
                             TEST_LONG_TERM_MEM_GAP=2, TEST_SHORT_TERM_MEM_SKIP=1, MODEL_USE_PREV_PROB=False).items():
                setattr(ecfg, k, v)
            ev = ev_mod.Evaluator.__new__(ev_mod.Evaluator)
            ev.cfg, ev.model, ev.gpu, ev.gpu_num, ev.rank = ecfg, net, 0, 1, 0
            ev.seq_queue = ev.info_queue = None
            ev.dataset = [ds]
            ev.result_root = os.path.join(tmp, 'results')
            ev.source_folder, ev.zip_dir = ev.result_root, os.path.join(tmp, 'results.zip')
            os.makedirs(os.path.join(ev.result_root, SEQ))
            real_build, real_zip = ev_mod.build_engine, ev_mod.zip_folder

            def spy_build(*a, **k):
                e = real_build(*a, **k)
                inner = e.decode_current_logits

                def dec(output_size=None):
                    lg = inner(output_size)
                    prob = torch.softmax(lg, 1)[0]
                    top2 = torch.topk(prob, 2, 0)[0]
                    gap = top2[0] - top2[1]
                    decoded.append((tuple(lg.shape), (gap < 1e-3).numpy(), (gap < 2e-4).numpy(), tuple(e.input_size_2d)))
                    print('  decoded frame %d: logits %s, input %s, near-ties %d / %d' % (
                        len(decoded), tuple(lg.shape), tuple(e.input_size_2d), int(decoded[-1][1].sum()), int(decoded[-1][2].sum())), flush=True)
                    return lg
                e.decode_current_logits = dec
                return e
            ev_mod.build_engine = spy_build
            ev_mod.save_mask = lambda m, path, idx: written.append((os.path.relpath(path, ev.result_root), m.clone(), list(idx)))
            ev_mod.zip_folder = lambda *a, **k: None
            try:
                ev.evaluating()
            finally:
                ev_mod.build_engine, ev_mod.zip_folder = real_build, real_zip
    finally:
        torch.Tensor.cuda = saved_tcuda
        for k, v in saved_cuda.items():
            setattr(torch.cuda, k, v)
        for k, v in saved_mods.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        refdriver._leave()
    assert [w[0] for w in written] == ['%s/%s' % (SEQ, n.replace('jpg', 'png')) for n in names[1:]], [w[0] for w in written]
    assert len(decoded) == FRAMES - 1 and all(w[2] == list(range(45)) for w in written)
    out = {'masks': np.stack([w[1].numpy().astype(np.uint8) for w in written]),
           'ties_1e3': np.packbits(np.stack([d[1] for d in decoded])), 'ties_2e4': np.packbits(np.stack([d[2] for d in decoded])),
           'input_size': np.array(decoded[0][3]), 'logit_shape': np.array(decoded[0][0]), 'names': np.array(names),
           'model': np.array(MODEL), 'gap': np.array(2)}
    np.savez_compressed(os.path.join(OUT, 'golden.npz'), **out)
    print('demo_1001: frames', len(written), 'objects per mask', [len(np.unique(m)) - 1 for m in out['masks']],
          'near-ties (1e-3 / 2e-4)', [(int(d[1].sum()), int(d[2].sum())) for d in decoded], flush=True)


if __name__ == '__main__':
    main()

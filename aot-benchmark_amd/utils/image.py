"""Label helpers on the hot path (reference utils/image.py:69-74), the evaluator's size rule, and the result writers
on the far side of the engine (reference utils/image.py:5-113: palette PNG masks, colour maps, overlays)."""
import threading

import numpy as np
import torch


def davis_palette():
    """The 256-entry palette the reference attaches to every written mask (utils/image.py:6-56), as a flat RGB list:
    entries 0..21 are the PASCAL-VOC bit-interleaved colours with 191 where VOC has 192, entries 22..255 are the grey
    ramp (i, i, i).  Pinned on the reference's table by tests/golden/image_utils.npz."""
    pal = []
    for i in range(256):
        if i >= 22:
            pal += [i, i, i]
            continue
        rgb = [0, 0, 0]
        c = i
        for j in range(8):
            for ch in range(3):
                rgb[ch] |= ((c >> ch) & 1) << (7 - j)
            c >>= 3
        pal += [191 if v == 192 else v for v in rgb]
    return pal


def label2colormap(label):
    """[H, W] ids -> [H, W, 3] uint8 colours: three bits per channel interleaved (utils/image.py:59-66; ids are taken
    modulo 256)."""
    m = np.asarray(label).astype(np.uint8)
    out = np.zeros(m.shape + (3,), dtype=np.uint8)
    for ch, spread in enumerate(((0, 3, 6), (1, 4, 7), (2, 5))):
        v = np.zeros(m.shape, dtype=np.uint8)
        for n, bit in enumerate(spread):
            v |= ((m >> bit) & 1) << (7 - n)
        out[..., ch] = v
    return out


def masked_image(image, colored_mask, mask, alpha=0.7):
    """Overlay of demo.py (utils/image.py:77-82): image / colours are [3, H, W], mask [H, W]; foreground pixels blend."""
    fg = np.broadcast_to(np.asarray(mask)[None] > 0, np.asarray(image).shape)
    return np.where(fg, image * alpha + colored_mask * (1 - alpha), image)


def unsqueeze_ids(mask, squeeze_idx):
    """Maps the engine's dense ids 1..n back to the dataset's object ids (squeeze_idx[k] = dataset id of dense id k;
    utils/image.py:91-97).  Ids without an entry become background."""
    mask = np.asarray(mask)
    if squeeze_idx is None:
        return mask
    lut = np.zeros(256, dtype=np.uint8)
    n = min(len(squeeze_idx), 256)
    lut[1:n] = np.asarray(squeeze_idx[1:n], dtype=np.int64).astype(np.uint8)
    return lut[mask.astype(np.uint8)]


def _write_mask(mask, path, squeeze_idx=None):
    from PIL import Image
    im = Image.fromarray(unsqueeze_ids(mask, squeeze_idx).astype(np.uint8)).convert('P')
    im.putpalette(davis_palette())
    im.save(path)


def save_mask(mask_tensor, path, squeeze_idx=None, wait=False):
    """Writes a predicted label map as a palette PNG from a background thread (utils/image.py:90-105).  Returns the
    thread (joined already when wait=True)."""
    mask = mask_tensor.detach().cpu().numpy().astype('uint8')
    th = threading.Thread(target=_write_mask, args=[mask, path, squeeze_idx])
    th.start()
    if wait:
        th.join()
    return th


def save_image(image, path):
    """[3, H, W] float image in 0..1 -> RGB file (utils/image.py:85-87)."""
    from PIL import Image
    Image.fromarray(np.uint8(np.asarray(image) * 255.).transpose((1, 2, 0))).save(path)


def flip_tensor(tensor, dim=0):
    return torch.flip(tensor, dims=(dim,))


def one_hot_mask(mask, cls_num):
    """[B,1,H,W] (or [B,H,W]) label ids -> [B,cls_num+1,H,W] float one-hot; ids above
    ``cls_num`` (or non-integer values) give an all-zero column, as in the reference."""
    if mask.dim() == 3:
        mask = mask.unsqueeze(1)
    ids = torch.arange(0, cls_num + 1, device=mask.device).view(1, -1, 1, 1)
    return (mask == ids).float()


def _snap(n, stride, plus_one):
    """n -> nearest k*stride (+1): numpy's round-half-to-even on the quotient, as the reference uses np.around."""
    off = 1 if plus_one else 0
    if (n - off) % stride == 0:
        return n
    return int(np.around((n - off) / stride) * stride + off)


def restrict_size(h, w, max_short_edge=None, max_long_edge=800, scale=1.0, align_corners=True, max_stride=16):
    """Network input size of an h x w frame (the rule of MultiRestrictSize, dataloaders/video_transforms.py:612-653):
    cap the short edge, then the long edge (both by uniform float scaling), apply the multi-scale factor with truncation,
    then snap each side to k*stride (+1 when the model aligns corners).  Pinned on the reference class by
    tests/golden/transforms.json."""
    fh, fw = float(h), float(w)
    if max_short_edge is not None and min(h, w) > max_short_edge:
        r = float(max_short_edge) / min(h, w)
        fh, fw = r * fh, r * fw
    if max_long_edge is not None and max(fh, fw) > max_long_edge:
        r = float(max_long_edge) / max(fh, fw)
        fh, fw = r * fh, r * fw
    return (_snap(int(fh * scale), max_stride, align_corners), _snap(int(fw * scale), max_stride, align_corners))

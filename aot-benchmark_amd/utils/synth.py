"""Deterministic synthetic weights and clips.

There is no network access for checkpoints or datasets, so parity fixtures and
``bench.py`` use *keyed* synthetic weights: every tensor is generated from its own
``state_dict`` key, so the reference model (in the build container, when the goldens are
made) and this engine (on the GPU box) get bit-identical parameters without shipping
60 MB of weights.  The recipe is calibrated so that the LSTT actually matters to the
output (random default init makes the attention maps near-uniform and the encoder
dominate; see SURVEY.md section 7 'hard parts').
"""
import zlib

import torch


def _gen(key, salt=0xA07):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) ^ salt) & 0x7FFFFFFF)
    return g


def synth_tensor(key, ref, all_keys):
    """One synthetic tensor for ``key`` with the shape of ``ref``."""
    shape = tuple(ref.shape)
    g = _gen(key)
    leaf = key.rsplit('.', 1)[-1]
    if not ref.is_floating_point():
        return ref.detach().clone()          # index buffers (Swin relative_position_index) are structural: keep
    if leaf == 'running_mean':
        return 0.1 * torch.randn(shape, generator=g)
    if leaf == 'running_var':
        return 1.0 + 0.1 * torch.rand(shape, generator=g)
    if leaf == 'relative_position_bias_table':
        return 0.5 * torch.randn(shape, generator=g)
    if leaf == 'relative_emb_v':
        return 0.05 * torch.randn(shape, generator=g)
    if leaf == 'mask_token':
        return torch.randn(shape, generator=g)
    if ref.dim() >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        w = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
        if key == 'patch_wise_id_bank.weight':
            w = torch.randn(shape, generator=g) / float(shape[-1])          # id_emb std ~ 1
        elif key == 'encoder_projector.weight':
            w = w * 0.7071
        elif 'relative_emb_k' in key:
            w = w * 1.5
        elif key.endswith(('linear_Q.weight', 'linear_K.weight', 'linear_QK.weight')) or '.linear_QV.' in key:
            w = w * 1.6                                                     # sharper attention maps
        elif key.startswith('decoder.conv_out'):
            w = w * 4.0                                                     # spread the logits
        return w
    # 1-D tensors: a bias of a conv/linear if the sibling weight is a matrix, else a norm parameter
    stem = key.rsplit('.', 1)[0]
    sib = stem + '.weight'
    if leaf == 'bias':
        is_lin = sib in all_keys and all_keys[sib].dim() >= 2
        return (0.02 if is_lin else 0.1) * torch.randn(shape, generator=g) * _swin_out_norm_gain(stem)
    w = 1.0 + 0.1 * torch.randn(shape, generator=g)
    # damp the residual branch of every encoder block so activations stay O(1) through the backbone
    if key.startswith('encoder.') and (stem.endswith('.bn3') or _is_mbv2_last_bn(stem, all_keys)):
        w = w * 0.25
    return w * _swin_out_norm_gain(stem)


def _swin_out_norm_gain(stem):
    """Round 4 calibration of the Swin trunk.  The per-stage output LayerNorms of the Swin
    encoder (`encoder.norm0/1/2`, swin_transformer.py:630-633) hand the decoder unit-variance shortcuts, where the ResNet
    trunks hand it a residual stream of std 4-8.  With unit-variance shortcuts the decoder's logits respond ~10x more
    strongly to the memory read-out, and a one-pixel change of a fed-back mask changes ~5 pixels of the next mask: the
    free-running clip is chaotic for the reference itself (`tools/dev/chaos_probe.py`, `profiles/r04_swinb_chaos_probe.txt`).
    Weight AND bias of those three norms x3 puts the Swin models in the ResNet models' regime (growth factor < 1)."""
    return 3.0 if stem in ('encoder.norm0', 'encoder.norm1', 'encoder.norm2') else 1.0


def _is_mbv2_last_bn(stem, all_keys):
    # encoder.features.N.conv.K where K is the last index of the block's Sequential
    parts = stem.split('.')
    if len(parts) == 5 and parts[1] == 'features' and parts[3] == 'conv' and parts[4].isdigit():
        nxt = '.'.join(parts[:4] + [str(int(parts[4]) + 1)])
        return not any(k.startswith(nxt + '.') for k in all_keys)
    return False


def synth_state_dict(reference_state_dict):
    """Keyed synthetic replacement for every entry of ``reference_state_dict``
    (only names, shapes and dtypes are used)."""
    return {k: synth_tensor(k, v, reference_state_dict) for k, v in reference_state_dict.items()}


def align_size(h, w, align_corners=True, stride=16):
    """Size rule of the evaluator's MultiRestrictSize (reference
    dataloaders/video_transforms.py:640-655): (x-1) % 16 == 0 with align_corners, else x % 16 == 0."""
    def fix(x):
        if align_corners:
            return x if (x - 1) % stride == 0 else int(round((x - 1) / stride) * stride + 1)
        return x if x % stride == 0 else int(round(x / stride) * stride)
    return fix(h), fix(w)


def synth_clip(k, num_frames, in_size=(481, 849), out_size=(480, 854), num_obj=10, device='cpu'):
    """Clip ``k`` of SURVEY.md section 8d: frame t = roll(I0, (2t, 3t)) + 0.05*noise, first-frame
    mask = grid of rectangles labelled 1..num_obj on background 0.  Returns
    (frames list of [1,3,H,W] float32, mask [1,1,H,W] float32 at in_size, obj_nums, out_size)."""
    H, W = in_size
    g = torch.Generator()
    g.manual_seed(1000 + k)
    base = torch.randn(3, H, W, generator=g)
    # add low-frequency structure so objects are distinguishable
    yy = torch.linspace(0, 6.28318, H).view(1, H, 1)
    xx = torch.linspace(0, 6.28318, W).view(1, 1, W)
    base = base + 1.5 * torch.sin(yy * (1 + k % 3)) * torch.cos(xx * 2) + torch.sin(xx * 3 + yy)
    mask = torch.zeros(1, 1, H, W)
    rows = 2 if num_obj > 1 else 1
    cols = (num_obj + rows - 1) // rows
    ch, cw = H // rows, W // cols
    rh, rw = int(ch * 0.6), int(cw * 0.8)
    for o in range(num_obj):
        r, c = divmod(o, cols)
        y0 = r * ch + (ch - rh) // 2
        x0 = c * cw + (cw - rw) // 2
        mask[0, 0, y0:y0 + rh, x0:x0 + rw] = o + 1
        base[:, y0:y0 + rh, x0:x0 + rw] += 0.8 * torch.randn(3, 1, 1, generator=g)   # per-object tint
    frames = []
    for t in range(num_frames):
        f = torch.roll(base, shifts=(2 * t, 3 * t), dims=(1, 2)) + 0.05 * torch.randn(3, H, W, generator=g)
        frames.append(f.unsqueeze(0).contiguous().to(device))
    return frames, mask.to(device), [num_obj], tuple(out_size)

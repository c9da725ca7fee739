"""Checkpoints with the reference's conventions (utils/checkpoint.py): weights accepted as {'state_dict': ...},
{'model': ...} or a raw dict with a leading 'module.' stripped (:94-121); training checkpoints {'state_dict',
'optimizer'[, 'scaler']} written and resumed in the same files and layout (:13-91,124-160)."""
import torch


def get_device(gpu=None):
    """The device checkpoints are mapped to (utils/checkpoint.py:7-10)."""
    if torch.cuda.is_available():
        return torch.device('cuda', gpu) if gpu is not None else torch.device('cuda')
    return torch.device('cpu')


def load_network(net, pretrained_dir, gpu=None, trusted=False):
    """Only tensors are needed (a state_dict, optionally wrapped in {'state_dict': ...} / {'model': ...}), so the file is
    read with weights_only=True: a downloaded checkpoint cannot run pickled code.  trusted=True restores torch's full
    unpickler for old checkpoints that carry other objects next to the weights."""
    device = torch.device('cpu') if (gpu is None or gpu < 0 or not torch.cuda.is_available()) \
        else torch.device('cuda', gpu)
    blob = torch.load(pretrained_dir, map_location='cpu', weights_only=not trusted)
    return load_state(net, blob, device)


def load_state(net, blob, device=None):
    for key in ('state_dict', 'model'):
        if isinstance(blob, dict) and key in blob:
            blob = blob[key]
            break
    own = net.state_dict()
    removed = []
    for k, v in blob.items():
        if k in own:
            own[k] = v
        elif k.startswith('module.') and k[7:] in own:
            own[k[7:]] = v
        else:
            removed.append(k)
    net.load_state_dict(own)
    if device is not None:
        net = net.to(device)
    return net, removed


# ---- training checkpoints (reference utils/checkpoint.py:13-91,124-160): {'state_dict', 'optimizer'[, 'scaler']} in
# `save_step_<step>.pth`, the optimiser part in torch.optim's state-dict layout (utils/optim.py writes the same) ------------
def _read_training_blob(path, trusted):
    # an optimiser state dict is tensors, numbers, strings, lists and dicts: still loadable with weights_only=True
    return torch.load(path, map_location='cpu', weights_only=not trusted)


def load_network_and_optimizer(net, opt, pretrained_dir, gpu=None, scaler=None, trusted=False):
    """Resume: weights by key (a leading 'module.' stripped), optimiser state by position of the parameter groups."""
    device = torch.device('cpu') if (gpu is None or gpu < 0 or not torch.cuda.is_available()) \
        else torch.device('cuda', gpu)
    blob = _read_training_blob(pretrained_dir, trusted)
    net, removed = load_state(net, {'state_dict': blob['state_dict']}, device)
    opt.load_state_dict(blob['optimizer'])
    if scaler is not None and 'scaler' in blob:
        scaler.load_state_dict(blob['scaler'])
    return net, opt, removed


def load_network_and_optimizer_v2(net, opt, pretrained_dir, gpu=None, scaler=None, trusted=False):
    """Resume when the set of trainable parameters changed: the saved optimiser groups are matched to the live ones by their
    'name' (one parameter per group, utils/learning.py); groups the checkpoint does not know keep their fresh state, saved
    groups without a live counterpart are dropped."""
    device = torch.device('cpu') if (gpu is None or gpu < 0 or not torch.cuda.is_available()) \
        else torch.device('cuda', gpu)
    blob = _read_training_blob(pretrained_dir, trusted)
    net, removed = load_state(net, {'state_dict': blob['state_dict']}, device)
    live = opt.state_dict()
    slot = {g['name']: g['params'][0] for g in live['param_groups']}
    saved = blob['optimizer']
    state, taken = dict(live['state']), {}
    for g in saved['param_groups']:
        if g['name'] not in slot:
            continue
        old = g['params'][0]
        g = dict(g)
        g['params'] = [slot[g['name']]]
        taken[g['name']] = g
        if old in saved['state']:
            state[slot[g['name']]] = saved['state'][old]
    groups = [taken.get(g['name'], g) for g in live['param_groups']]
    opt.load_state_dict({'state': state, 'param_groups': groups})
    if scaler is not None and 'scaler' in blob:
        scaler.load_state_dict(blob['scaler'])
    return net, opt, removed


def save_network(net, opt, step, save_path, max_keep=8, backup_dir='./saved_models', scaler=None):
    """Writes `<save_path>/save_step_<step>.pth` (into backup_dir when save_path cannot be written) and keeps only the
    max_keep highest steps there."""
    import pathlib
    blob = {'state_dict': net.state_dict(), 'optimizer': opt.state_dict()}
    if scaler is not None:
        blob['scaler'] = scaler.state_dict()
    where = pathlib.Path(save_path)
    try:
        where.mkdir(parents=True, exist_ok=True)
        torch.save(blob, where / ('save_step_%s.pth' % step))
    except Exception:
        where = pathlib.Path(backup_dir)
        where.mkdir(parents=True, exist_ok=True)
        torch.save(blob, where / ('save_step_%s.pth' % step))
    steps = sorted(int(f.stem.rsplit('_', 1)[-1]) for f in where.glob('save_step_*.pth'))
    for old in steps[:max(0, len(steps) - max_keep)]:
        f = where / ('save_step_%d.pth' % old)
        if f.exists():
            f.unlink()

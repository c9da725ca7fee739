"""Checkpoint loading with the reference's key convention (utils/checkpoint.py:94-121):
accepts {'state_dict': ...}, {'model': ...} or a raw dict, strips a leading 'module.'."""
import torch


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

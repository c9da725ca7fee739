"""Learning-rate schedule and parameter groups of the trainer (reference utils/learning.py:4-90) -- host logic."""
import math


def adjust_learning_rate(optimizer, base_lr, p, itr, max_itr, restart=1, warm_up_steps=1000, is_cosine_decay=False, min_lr=1e-5,
                         encoder_lr_ratio=1.0, freeze_params=[]):
    """Linear warm-up, then polynomial (power p) or cosine decay to min_lr, optionally restarted; encoder groups get a
    reduced rate, frozen groups rate 0 and no weight decay.  Returns the base rate of this iteration."""
    if restart > 1:
        each = int(math.ceil(float(max_itr) / restart))
        itr, warm_up_steps, max_itr = itr % each, warm_up_steps / restart, each
    span = base_lr - min_lr
    if itr < warm_up_steps:
        now_lr = min_lr + span * itr / warm_up_steps
    else:
        t, horizon = itr - warm_up_steps, max_itr - warm_up_steps
        if is_cosine_decay:
            now_lr = min_lr + span * (math.cos(math.pi * t / (horizon + 1)) + 1.) * 0.5
        else:
            now_lr = min_lr + span * (1 - t / (horizon + 1)) ** p
    for group in optimizer.param_groups:
        name = group['name']
        group['lr'] = (now_lr - min_lr) * encoder_lr_ratio + min_lr if (encoder_lr_ratio != 1.0 and 'encoder.' in name) else now_lr
        if any(f in name for f in freeze_params):
            group['lr'] = 0
            group['weight_decay'] = 0
    return now_lr


def get_trainable_params(model, base_lr, weight_decay, use_frozen_bn=False, exclusive_wd_dict={}, no_wd_keys=[], verbose=False):
    """One parameter group per trainable tensor, named by its key.  Weight decay: the per-key override if one matches; 1-D
    tensors (norm scales, biases) get none -- except frozen-BN scales inside the encoder when frozen BN is in use --; any
    other tensor whose key contains an exempted substring gets none."""
    groups, seen, total = [], set(), 0
    for key, value in model.named_parameters():
        if value in seen:
            continue
        total += value.numel()
        if not value.requires_grad:
            continue
        seen.add(value)
        wd = next((v for k, v in exclusive_wd_dict.items() if k in key), weight_decay)
        if value.dim() == 1:
            if 'bias' in key or not use_frozen_bn or 'encoder.' not in key:
                wd = 0.
        elif any(k in key for k in no_wd_keys):
            wd = 0.
        groups.append({'params': [value], 'lr': base_lr, 'weight_decay': wd, 'name': key})
    if verbose:
        print('Total Param: {:.2f}M'.format(total / 1e6))
    return groups


def freeze_params(module):
    """Stops the gradients of every parameter of `module` (reference utils/learning.py:93-95)."""
    for p in module.parameters():
        p.requires_grad = False


def calculate_params(state_dict):
    """Prints and returns the number of distinct parameters of a state dict, in millions (utils/learning.py:98-106: shared
    tensors count once)."""
    seen, total = set(), 0
    for v in state_dict.values():
        if id(v) not in seen:
            seen.add(id(v))
            total += v.numel()
    print('Total Param: {:.2f}M'.format(total / 1e6))
    return total / 1e6

"""The data-parallel training step of BASELINE config 5 (R50-DeAOTL, configs/pre_ytb_dav.py: 465 x 465 crops, 5 frames, 2 samples per
GPU) under torch.distributed with backend 'nccl' (= RCCL over xGMI), one process per GPU:

    python tools/dev/train_ddp.py --gpus N [--model r50_deaotl] [--precision bf16] [--steps 6]
    (or: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/dev/train_ddp.py ...)

networks/managers/trainer.py::TrainStep over utils/flat_state.py: gradients accumulate into flat bucket memory, a bucket's
all-reduce leaves from the post-accumulate-grad hook while backward runs, clip + AdamW + EMA are three launches.  Synthetic clips,
keyed synthetic weights; prints one JSON line (ms per step = max over ranks, frames / s of the whole job, buckets issued inside
backward, peak memory).  Works at N = 1 (a one-rank RCCL group: what the builder's box can run)."""
import argparse
import json
import os
import subprocess
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(R, 'aot-benchmark_amd'), os.path.join(R, 'tests')]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--model', default='r50_deaotl')
    ap.add_argument('--stage', default='pre_ytb_dav', help='engine config module under configs/ (BASELINE config 5: pre_ytb_dav)')
    ap.add_argument('--precision', default='bf16', choices=['f32', 'bf16'])
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--batch', type=int, default=2, help='samples per GPU (TRAIN_BATCH_SIZE 16 over 8 GPUs)')
    ap.add_argument('--frames', type=int, default=5)
    ap.add_argument('--size', type=int, default=465)
    ap.add_argument('--bucket-mb', type=float, default=32.0)
    ap.add_argument('--backend', default='nccl')
    ap.add_argument('--profile', default=None, help='after the timed steps: one more step under torch.profiler; device kernels by '
                    'the Python line / autograd node that launched them -> this text file')
    a = ap.parse_args()
    if 'WORLD_SIZE' not in os.environ:
        import socket
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ); env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        raise SystemExit(subprocess.call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
                                          '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:],
                                         env=env))
    import torch
    import torch.distributed as dist
    import importlib
    from networks.models import build_vos_model
    from utils.synth import synth_state_dict
    from networks.engines import build_engine
    from networks.managers.trainer import TrainStep
    from utils.synth import synth_clip
    world, rank, local = int(os.environ['WORLD_SIZE']), int(os.environ['RANK']), int(os.environ.get('LOCAL_RANK', 0))
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    dist.init_process_group(a.backend, rank=rank, world_size=world, device_id=dev if a.backend == 'nccl' else None)
    # BASELINE config 5's stage (configs/pre_ytb_dav.py) over the model's preset; the same keyed synthetic weights on every rank
    cfg = importlib.import_module('configs.' + a.stage).EngineConfig('train_ddp', a.model)
    model = build_vos_model(cfg.MODEL_VOS, cfg)
    model.load_state_dict(synth_state_dict(model.state_dict()))
    S = a.size if cfg.MODEL_ALIGN_CORNERS else a.size // 16 * 16
    model = model.to(dev).train()
    engine = build_engine(cfg.MODEL_ENGINE, phase='train', aot_model=model, gpu_id=local, long_term_mem_gap=cfg.TRAIN_LONG_TERM_MEM_GAP).train()
    step_fn = TrainStep(cfg, model, engine, precision=a.precision, bucket_mb=a.bucket_mb, ema=rank == 0)
    bs, T = a.batch, a.frames
    frames, masks, objs = [], [], []
    for b in range(bs):                                              # this rank's samples: distinct clips per rank
        f, m, o, _ = synth_clip(40 + rank * bs + b, T, (S, S), (S, S), 3 + b, device=dev)
        frames.append(torch.cat(f, 0))
        masks.append(m.expand(T, -1, -1, -1))
        objs.append(3 + b)
    all_frames = torch.stack(frames, 1).reshape(T * bs, 3, S, S).contiguous()
    all_masks = torch.stack(masks, 1).reshape(T * bs, 1, S, S).contiguous().float()
    losses = [float(step_fn(all_frames, all_masks, objs, 0)[0])]           # step 0: allocations, weight packs
    torch.cuda.synchronize(dev)
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step_fn(all_frames, all_masks, objs, i + 1)[0]
        losses.append(loss)
    torch.cuda.synchronize(dev)
    dist.barrier()
    dt = torch.tensor([(time.perf_counter() - t0) / a.steps], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    st = step_fn.state
    # replicas must still hold identical parameters: a checksum of the flat buffer, compared across ranks
    chk = st.flat_p.double().sum().view(1)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({'metric': 'training step, %s, %s' % (a.model, a.precision), 'ms_per_step': round(float(dt) * 1e3, 1),
                          'frames_per_s': round(world * bs * T / float(dt), 1), 'n_gpus': world, 'backend': a.backend,
                          'batch_per_gpu': bs, 'frames': T, 'size': S, 'buckets': len(st.buckets),
                          'buckets_issued_inside_backward': st.launched_in_backward, 'bucket_mb': a.bucket_mb,
                          'params_m': round(st.total / 1e6, 2), 'replicas_identical': bool(float(lo) == float(hi)),
                          'losses': [round(float(x), 4) for x in losses], 'grad_norm_last': round(st.grad_norm(), 3),
                          'peak_mem_gib': round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2)}))
    if a.profile and rank == 0:
        from networks.layers import train_ops
        train_ops.MATMUL_LOG = {}
        profile_step(lambda: step_fn(all_frames, all_masks, objs, a.steps + 1), a.profile)
        with open(a.profile, 'a') as f:
            f.write('\nproducts on the strided kernel in that step (count x [bt, m, k, n], alpha, strides; GFLOP each):\n')
            for (bt, m, k, n, alpha, sa, sb), cnt in sorted(train_ops.MATMUL_LOG.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][2] * kv[0][3]):
                f.write('%4d x [%d, %d, %d, %d] alpha %g  a%s b%s  %.3f GF\n' % (cnt, bt, m, k, n, alpha, sa, sb, 2e-9 * bt * m * k * n))
        train_ops.MATMUL_LOG = None
    dist.destroy_process_group()


def profile_step(run, path):
    """One step under torch.profiler (CPU + device, Python stacks): every device kernel attributed to what launched it -- in
    forward the innermost frame inside aot-benchmark_amd, in backward the autograd node -- and summed: which glue costs what."""
    import collections
    import torch
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        run()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        ks = getattr(e, 'kernels', None)
        if not ks:
            continue
        node, p = None, e
        while p is not None:
            if p.name.startswith('autograd::engine::evaluate_function'):
                node = p.name.split(': ')[-1]
                break
            p = p.cpu_parent
        site = node
        if site is None:
            site = '?'
            q = e
            while q is not None and site == '?':
                for fr in (q.stack or []):
                    if 'aot-benchmark_amd' in fr:
                        site = fr.split('aot-benchmark_amd/')[-1][:70]
                        break
                q = q.cpu_parent
        for k in ks:
            name = k.name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0][-60:]
            if 'elementwise' in name or 'Functor' in k.name:
                name = 'torch:' + e.name
            a = agg[(name, site)]
            a[0] += 1
            a[1] += k.duration
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    total = sum(v[1] for _, v in rows)
    with open(path, 'w') as f:
        f.write('device time of one step by (kernel, launching site): %.1f ms in %d launches\n' % (total / 1e3, sum(v[0] for _, v in rows)))
        for (name, site), (n, us) in rows[:150]:
            f.write('%8.1f us %5d x  %-46s %s\n' % (us, n, name[:46], site))
        by_site = collections.defaultdict(lambda: [0, 0.0])
        for (name, site), (n, us) in rows:
            if name.startswith('torch:'):
                by_site[site][0] += n
                by_site[site][1] += us
        f.write('\ntorch glue only, by site: %.1f ms\n' % (sum(v[1] for v in by_site.values()) / 1e3))
        for site, (n, us) in sorted(by_site.items(), key=lambda kv: -kv[1][1])[:60]:
            f.write('%8.1f us %5d x  %s\n' % (us, n, site))


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""bench.py -- frames/sec of the AOT hot path on synthetic 480p 10-object clips (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A step = one propagated frame of the R50-AOTL engine at 481x849 input / 480x854 output, 10 objects:
match_propogate_one_frame -> decode_current_logits -> softmax/argmax/nearest-resize -> update_memory,
the exact call sequence of the reference's evaluator (networks/managers/evaluator.py:325-446,
tools/demo.py:219-235).  Clips are 70 frames (the long-term bank grows from 1 to 14 frames, gap 5); frames
are resident in HBM before the timed region.  Per-video inference is embarrassingly parallel, so each GPU
runs --streams S clips concurrently, one HIP stream and one engine (memory bank, scratch) each, sharing the
weights: most kernels of one 480p frame cannot fill 256 CUs on their own.  `value` is the whole-job
throughput; `config.single_stream_fps` is the same job with S = 1 (one clip at a time, the reference's
evaluation mode).  Clips shard over ranks with no data-path collective
("weak" scaling: every rank runs K frames); RCCL is used only for the barrier, the max-over-ranks time and
one all_gather of a small stats vector (replaces the reference's mp.Queue, evaluator.py:507-531).

The JSON line also carries
  roofline     -- the long-term attention kernel (attn_fwd_d32_pipe_kernel) timed live with HIP events on its
                  stream: achieved = 4*N*T*C FLOP per launch / mean launch time, against the 157.3 TFLOP/s fp32
                  MFMA peak;
  cpu_baseline -- the CPU oracle (oracle/aot_oracle.py, a port of the reference's algorithm; the reference
                  itself cannot travel to the GPU box) timed on the host cores on the first frames of the same clip.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'aot-benchmark_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

MODEL = 'r50_aotl'
IN_SIZE, OUT_SIZE, NUM_OBJ, CLIP_FRAMES = (481, 849), (480, 854), 10, 70
FP32_MFMA_PEAK_TF = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md


def shard_clips(num_clips, rank, world):
    """clip i -> rank i mod world (equal-length synthetic clips)."""
    return [i for i in range(num_clips) if i % world == rank]


def gather_stats(stats, world):
    """One all_gather of a small float64 vector per rank; returns [world, len]."""
    if world == 1:
        return stats.unsqueeze(0).cpu()
    out = [torch.zeros_like(stats) for _ in range(world)]
    dist.all_gather(out, stats)
    return torch.stack(out).cpu()


def build_model(device):
    from networks.engines import build_engine
    from networks.models import build_vos_model
    from utils.synth import synth_state_dict
    cfg = importlib.import_module('configs.models.' + MODEL).ModelConfig()
    model = build_vos_model(cfg.MODEL_VOS, cfg)
    sd = synth_state_dict(model.state_dict())
    model.load_state_dict(sd)
    model = model.to(device).eval()
    engine = build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=model, gpu_id=device.index or 0,
                          long_term_mem_gap=cfg.TEST_LONG_TERM_MEM_GAP)
    return cfg, model, engine, sd


def one_frame(engine, img):
    """The per-frame body of the evaluator loop; the predicted mask feeds the memory update on device."""
    import aot_hip
    engine.match_propogate_one_frame(img)
    logit = engine.decode_current_logits(OUT_SIZE)
    # softmax -> mean over the (single) augmentation -> argmax, then the nearest-resized label feedback
    # (evaluator.py:332-352,394-408) as the two device kernels of csrc/prepost.hip
    label, aug_labels, _ = aot_hip.fuse_probs(logit, [False])
    engine.update_memory(aot_hip.label_resize(aug_labels[0], engine.input_size_2d[0], engine.input_size_2d[1]))
    return label


class ClipRunner:
    """Walks clips frame by frame; a new clip (restart + reference frame) starts whenever one is exhausted."""

    def __init__(self, engine, clips):
        self.engine, self.clips = engine, clips
        self.ci, self.t = -1, CLIP_FRAMES

    def step(self):
        if self.t >= CLIP_FRAMES:
            self.ci = (self.ci + 1) % len(self.clips)
            frames, mask, objs = self.clips[self.ci]
            self.frames = frames
            self.engine.restart_engine()
            self.engine.add_reference_frame(frames[0], mask, objs, frame_step=0)
            self.t = 1
        one_frame(self.engine, self.frames[self.t])
        self.t += 1


def attention_roofline(engine, clip, device):
    """Instrumented pass over one clip: HIP events on the launch stream around every long-term/self attention
    MFMA kernel launch (the merge launch is outside the bracket)."""
    import aot_hip
    recs = []

    def probe(phase, nq, t, heads):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        if phase == 0:
            recs.append([ev, None, 4.0 * nq * t * heads * 32])
        else:
            recs[-1][1] = ev
    frames, mask, objs = clip
    engine.restart_engine()
    engine.add_reference_frame(frames[0], mask, objs, frame_step=0)
    aot_hip.attn_probe = probe
    try:
        for t in range(1, len(frames)):
            one_frame(engine, frames[t])
    finally:
        aot_hip.attn_probe = None
    torch.cuda.synchronize(device)
    ms = sum(a.elapsed_time(b) for a, b, _ in recs)
    flop = sum(f for _, _, f in recs)
    n = len(recs)
    traffic = None     # HBM/fabric bytes per launch from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE), see profiles/
    tp = os.path.join(ROOT, 'profiles', 'r01c_attn_traffic.json')
    if os.path.exists(tp):
        with open(tp) as f:
            traffic = round(json.load(f)['traffic_bytes_per_launch'])
    return {'bound': 'mfma', 'achieved': round(flop / (ms * 1e-3) / 1e12, 2), 'peak': FP32_MFMA_PEAK_TF,
            'unit': 'TFLOP/s', 'frac': round(flop / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, 4), 'traffic': traffic,
            'kernel': 'attn_fwd_d32_pipe_kernel', 'launches': n, 'avg_launch_us': round(ms * 1e3 / n, 2),
            'gflop_per_launch': round(flop / n / 1e9, 3)}


def jf_vs_reference(device):
    """J&F of this engine's FREE-RUNNING masks against the real reference's masks on the committed golden clip of
    BASELINE config 2 (tests/golden/c2_r50_aotl.npz: R50-AOTL, 481x849, 10 objects, 6 propagated frames)."""
    import numpy as np
    from utils.metric import jf_per_object
    from utils.synth import synth_clip
    gp = os.path.join(ROOT, 'tests', 'golden', 'c2_r50_aotl.npz')
    if not os.path.exists(gp):
        return None
    gold = np.load(gp)['masks']
    cfg, model, engine, _ = build_model(device)
    frames, mask, objs, _ = synth_clip(0, gold.shape[0] + 1, IN_SIZE, OUT_SIZE, NUM_OBJ, device=device)
    engine.restart_engine()
    engine.add_reference_frame(frames[0], mask, objs, frame_step=0)
    js, fs, diff = [], [], 0
    for t in range(1, len(frames)):
        label = one_frame(engine, frames[t])[0, 0].long().cpu()
        ref = torch.from_numpy(gold[t - 1].astype('int64'))
        j, f = jf_per_object(label, ref, NUM_OBJ)
        js.append(j)
        fs.append(f)
        diff += int((label != ref).sum())
    J, Fm = sum(js) / len(js), sum(fs) / len(fs)
    return {'J': round(J, 6), 'F': round(Fm, 6), 'J&F': round((J + Fm) / 2, 6), 'frames': len(js),
            'pixels_differing': diff, 'of_pixels': int(gold.size), 'clip': 'tests/golden/c2_r50_aotl.npz (free-running)'}


def cpu_baseline(sd, budget_s=20.0, max_frames=12):
    """Oracle (CPU port of the reference algorithm) on the first frames of clip 0, same FPS definition."""
    from oracle.aot_oracle import OracleEngine, OracleModel
    from utils.synth import synth_clip
    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    frames, mask, objs, _ = synth_clip(0, max_frames + 1, IN_SIZE, OUT_SIZE, NUM_OBJ)
    eng = OracleEngine(OracleModel(MODEL, {k: v.cpu() for k, v in sd.items()}))
    done, spent = 0, 0.0
    with torch.no_grad():
        eng.add_reference_frame(frames[0], mask, objs)
        for t in range(1, max_frames + 1):
            t0 = time.perf_counter()
            eng.match_propogate_one_frame(frames[t])
            logit = eng.decode_current_logits(OUT_SIZE)
            label = torch.argmax(torch.softmax(logit, 1), 1, keepdim=True).float()
            eng.update_memory(F.interpolate(label, size=eng.input_size_2d, mode='nearest'))
            spent += time.perf_counter() - t0
            done += 1
            if spent > budget_s:
                break
    return {'value': round(done / spent, 3), 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
            'sample': 'frames 1..%d of clip 0 (481x849, 10 objects, bank M<=%d), oracle/aot_oracle.py fp32, %d torch threads'
                      % (done, 1 + done // 5, threads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3 * (CLIP_FRAMES - 1), help='propagated frames timed per GPU (default: one full 70-frame clip per stream)')
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--streams', type=int, default=3, help='clips processed concurrently per GPU (one HIP stream each)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU: the AOT hot path has no CPU fallback')
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)   # RCCL over xGMI

    from utils.synth import synth_clip
    cfg, model, engine, sd = build_model(device)
    S = max(1, args.streams)
    nclips_stream = max(1, -(-args.steps // (S * (CLIP_FRAMES - 1))))
    nclips_rank = S * nclips_stream
    my_ids = shard_clips(nclips_rank * world, rank, world)
    clips = []
    for cid in my_ids:
        frames, mask, objs, _ = synth_clip(cid, CLIP_FRAMES, IN_SIZE, OUT_SIZE, NUM_OBJ, device=device)
        clips.append((frames, mask, objs))
    from networks.engines import build_engine
    engines = [engine] + [build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=model, gpu_id=device.index or 0,
                                       long_term_mem_gap=cfg.TEST_LONG_TERM_MEM_GAP) for _ in range(S - 1)]
    streams = [torch.cuda.Stream(device) for _ in range(S)]

    def timed(runners, steps):
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in range(steps):
            with torch.cuda.stream(streams[i % len(runners)]):
                runners[i % len(runners)].step()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    with torch.no_grad():
        # priming (setup, untimed): one full clip per stream so the caching allocator, the per-stream scratch and the
        # memory banks have reached their steady-state size -- a growing allocator calls hipMalloc, which
        # synchronises the device and serialises the concurrently running clips (395 vs 445 fps on first/second pass)
        prime = [ClipRunner(engines[i], clips[i:i + 1]) for i in range(S)]
        for i in range(S * (CLIP_FRAMES - 1)):
            with torch.cuda.stream(streams[i % S]):
                prime[i % S].step()
        # warmup: W frames spread over the streams
        warm = [ClipRunner(engines[i], clips[i:i + 1]) for i in range(S)]
        for i in range(max(args.warmup, S)):
            with torch.cuda.stream(streams[i % S]):
                warm[i % S].step()
        runners = [ClipRunner(engines[i], clips[i::S]) for i in range(S)]
        elapsed = timed(runners, args.steps)
        single = None
        if S > 1 and rank == 0 and world == 1:      # the same job one clip at a time, for reference
            single = args.steps / timed([ClipRunner(engines[0], clips[:1])], args.steps)

        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tmax = float(tmax.item())
        stats = gather_stats(torch.tensor([elapsed, float(args.steps), torch.cuda.max_memory_allocated(device) / 2**30],
                                          dtype=torch.float64, device=device), world)

        roof = None
        if rank == 0 and not args.no_roofline:
            with torch.cuda.stream(streams[0]):
                roof = attention_roofline(engines[0], clips[0], device)

    base = jf = None
    if rank == 0:
        with torch.no_grad():
            jf = jf_vs_reference(device)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base = cpu_baseline(sd)
    if world > 1:
        dist.barrier()

    if rank == 0:
        total_frames = float(stats[:, 1].sum())
        line = {
            'metric': 'frames/sec, 480p 10-object synthetic clips; J&F vs reference',
            'value': round(total_frames / tmax, 2), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(tmax / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'R50-AOTL inference, 480p (481x849 in, 480x854 out) 10-object synthetic clips, '
                                   '70 frames/clip, long-term gap 5 (configs[1])',
                       'frames_per_clip': CLIP_FRAMES, 'clips_per_gpu': nclips_rank, 'streams_per_gpu': S,
                       'single_stream_fps': None if single is None else round(single, 2),
                       'parallelism': 'clip-sharded dp%d x %d concurrent clips per GPU' % (world, S),
                       'weights': 'keyed synthetic (utils/synth.py)', 'peak_mem_gib': round(float(stats[:, 2].max()), 2),
                       'timed_region': 'wall clock incl. reference-frame setup of each clip; steady state (one untimed '
                                       'priming clip per stream before the warmup steps)',
                       'jf_vs_reference': jf},
            'roofline': roof, 'cpu_baseline': base,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

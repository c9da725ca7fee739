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
throughput.  Clips shard over ranks with no data-path collective ("weak" scaling: every rank runs K frames);
RCCL is used only for the barrier, the max-over-ranks time and one all_gather of a small stats vector (replaces
the reference's mp.Queue, evaluator.py:507-531).

The JSON line also carries
  roofline       -- the long-term / self attention kernel of the timed arithmetic (attn_x6_d32_kernel; the gated kernel for
                    DeAOT models) timed live with HIP events on its stream over one clip's launch mix: achieved = algorithmic
                    FLOP per launch / mean launch time against the roof of the arithmetic (bf16x6: 2500 / 6 TF-equivalent;
                    fp32: 157.3 TF), fabric bytes per launch from the committed rocprofv3 PMC passes of the kernel as built
                    (profiles/r06_*traffic.json); roofline.gemm: the same for every conv / linear call of the clip;
                    roofline.other_arithmetic: the other kernel family;
  cpu_baseline   -- the CPU oracle (oracle/aot_oracle.py, a port of the reference's algorithm; the reference itself
                    cannot travel to the GPU box) timed on the host cores on the first frames of the same clip;
  config.whole_clip      -- whole 70-frame clips on every stream;
  config.single_stream   -- ONE clip at a time over a whole clip (the reference's evaluation mode; the look-ahead encoder on the
                            engine's side stream), `windows_fps` = the same in the --steps form, `online` = no look-ahead at
                            all, timed with device events per frame as evaluator.py:325-330,444-446,486-498 does;
  config.fp32_exact      -- the same plan on exact fp32 products;
  config.jf_vs_reference -- free-running masks against the real reference's over the 70-frame golden, every differing
                            pixel classified on the reference's own near-tie map and on its fp64 run (non-zero exit on a
                            pixel outside the near-ties);
  config.other_configs   -- BASELINE config 3 (SwinB-DeAOTL, 480 x 848) and R50-DeAOTL as sub-runs of the same script.
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
BF16_MFMA_PEAK_TF = 2500.0     # dense bf16 (same guide)


X6_WHAT = ('conv / linear layers with >= %d tiles of 64x64 on aot_conv2d_bf16x6_f32 (three truncated bf16 planes per operand, six of the '
           'nine partial products, fp32 accumulation; register-staged tile kernels gemm_x6rd / gemm_x6r, split-K through '
           'aot_conv2d_bf16x6k_f32 for the long-K layers of under-filled maps); long-term and self-attention on aot_attn_x6_f32 (aot_gated_attn_x6_f32 for DeAOT) '
           'over the memory bank kept pre-split by aot_attn_pack_x6_f32; everything else as in the fp32 family')


def aot_hip_x6_min_tiles():
    import aot_hip
    return aot_hip.X6_MIN_TILES


COLLECTIVES = {'all_gather': 0, 'all_reduce': 0, 'barrier': 0}      # what this rank issued (the sharding tests read it in --dry-run)


def pin_host(rank, world, local_rank, dry=False):
    """One Python thread drives one GPU (three graph launches per frame and stream): keep it -- and the torch CPU threads of the
    rank-0-only legs -- on the cores of the GPU's own NUMA node, split evenly between the ranks that share the node, so that eight
    ranks do not migrate over each other and rank 0's CPU legs cannot spill onto the cores of a rank that is still being timed.
    Best effort: returns a description, never raises."""
    ncpu = os.cpu_count() or 1
    info = {'threads': torch.get_num_threads(), 'affinity': 'unchanged'}
    if world == 1:
        return info                      # a single rank keeps torch's own defaults
    try:
        ncpu = len(os.sched_getaffinity(0))          # the cores this process may actually use (cgroup / affinity aware)
    except (AttributeError, OSError):
        pass
    info['threads'] = max(1, min(info['threads'], ncpu // world))    # never more than torch's own default
    try:
        torch.set_num_threads(info['threads'])
        if dry or not hasattr(os, 'sched_setaffinity'):
            return info
        prop = torch.cuda.get_device_properties(local_rank)
        bdf = '%04x:%02x:%02x.0' % (getattr(prop, 'pci_domain_id', 0), prop.pci_bus_id, prop.pci_device_id)
        node = -1
        pth = '/sys/bus/pci/devices/%s/numa_node' % bdf
        if os.path.exists(pth):
            node = int(open(pth).read().strip())
        cpus = sorted(os.sched_getaffinity(0))
        if node >= 0 and os.path.exists('/sys/devices/system/node/node%d/cpulist' % node):
            lst = []
            for part in open('/sys/devices/system/node/node%d/cpulist' % node).read().strip().split(','):
                a, _, b = part.partition('-')
                lst += list(range(int(a), int(b or a) + 1))
            cpus = sorted(set(lst) & set(cpus)) or cpus
        # the ranks whose GPUs hang off this node share its cores in rank order (GPUs assumed evenly spread over the nodes;
        # node unknown: all ranks share all cores evenly)
        per_node = world if node < 0 else max(1, world // _numa_nodes())
        share = max(1, len(cpus) // per_node)
        k = local_rank % per_node
        mine = cpus[k * share:(k + 1) * share] or cpus
        os.sched_setaffinity(0, mine)
        info.update(affinity='numa node %d, cpus %d-%d (%d)' % (node, mine[0], mine[-1], len(mine)), threads=min(info['threads'], len(mine)))
        torch.set_num_threads(info['threads'])
    except Exception as e:           # noqa: BLE001
        info['affinity'] = 'unchanged (%s)' % type(e).__name__
    return info


def _numa_nodes():
    try:
        return max(1, len([d for d in os.listdir('/sys/devices/system/node') if d.startswith('node') and d[4:].isdigit()]))
    except OSError:
        return 1


def shard_clips(num_clips, rank, world):
    """clip i -> rank i mod world (equal-length synthetic clips)."""
    return [i for i in range(num_clips) if i % world == rank]


def gather_stats(stats, world):
    """One all_gather of a small float64 vector per rank; returns [world, len]."""
    if world == 1:
        return stats.unsqueeze(0).cpu()
    out = [torch.zeros_like(stats) for _ in range(world)]
    COLLECTIVES['all_gather'] += 1
    dist.all_gather(out, stats)
    return torch.stack(out).cpu()


def build_model(device, graph=False, gemm_table='latency', mfma='f32'):
    from networks.engines import build_engine
    from networks.models import build_vos_model
    from utils.synth import synth_state_dict
    cfg = importlib.import_module('configs.models.' + MODEL).ModelConfig()
    model = build_vos_model(cfg.MODEL_VOS, cfg)
    sd = synth_state_dict(model.state_dict())
    model.load_state_dict(sd)
    model = model.to(device).eval()
    engine = build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=model, gpu_id=device.index or 0,
                          long_term_mem_gap=cfg.TEST_LONG_TERM_MEM_GAP, graph=graph, gemm_table=gemm_table, mfma=mfma)
    return cfg, model, engine, sd


def one_frame(engine, img, feedback=None):
    """The per-frame body of the evaluator loop; the predicted mask feeds the memory update on device.  feedback (J&F leg
    of a chaotic clip only): label [1,1,H,W] -> the label map to memorise."""
    import aot_hip
    engine.match_propogate_one_frame(img)
    if feedback is None and not os.environ.get('AOT_NO_TAIL'):
        # decode -> softmax -> mean over the (single) augmentation -> argmax -> nearest-resized label feedback
        # (aot_engine.py:356-380, evaluator.py:332-352,394-408): one replay, the tail as ONE kernel (aot_frame_tail_f32),
        # bit-identical to decode_current_logits + aot_hip.fuse_probs + aot_hip.label_resize
        label, fb = engine.decode_current_labels(OUT_SIZE)
        engine.update_memory(fb)
        return label
    logit = engine.decode_current_logits(OUT_SIZE)
    label, aug_labels, _ = aot_hip.fuse_probs(logit, [False])
    fb = aug_labels[0] if feedback is None else feedback(label)
    engine.update_memory(aot_hip.label_resize(fb, engine.input_size_2d[0], engine.input_size_2d[1]))
    return label


OVERLAP_ENCODE = -1         # --overlap-encode: 1 / 0 = every StreamClip's look-ahead batches do / do not go to the engine's side stream;
                            # -1 (default) = in the one-clip-at-a-time legs only (measured: +9 % there, -3 % with three clips per GPU)


class StreamClip:
    """One clip on one HIP stream with its own engine.  `t` is the next frame to propagate (1 .. CLIP_FRAMES-1)."""

    def __init__(self, engine, stream, clip):
        self.engine, self.stream, self.clip = engine, stream, clip
        self.t = None
        self.feedback = None    # see one_frame()
        self.ahead = 0          # > 1: the encoder runs over the next `ahead` frames of the clip as one batch (engine.encode_ahead)
        self.overlap = OVERLAP_ENCODE == 1    # the batch AFTER the one being propagated is encoded meanwhile, on the engine's side stream
        self._encoded = 0       # frames from t on whose features are waiting in the engine
        self._prefetched = 0    # frames behind those whose batch is in flight on the side stream

    def restart(self):
        """restart_engine + add_reference_frame: per-clip set-up, never inside the timed region (the reference's
        FPS excludes the first frame, evaluator.py:325-330,444-446)."""
        frames, mask, objs = self.clip
        with torch.cuda.stream(self.stream):
            self.engine.restart_engine()
            self.engine.add_reference_frame(frames[0], mask, objs, frame_step=0)
        self.t = 1
        self._encoded = self._prefetched = 0

    def drop_ahead(self):
        """Forgets features encoded ahead of time: a timed window pays for the encoder of every frame it propagates."""
        if self._encoded or self._prefetched:
            with torch.cuda.stream(self.stream):
                self.engine.encode_ahead([])
            self._encoded = self._prefetched = 0

    def _issue(self, first, limit):
        """Encodes frames [first, first + n) as one batch, n = min(look-ahead, frames left in the clip, limit); returns n (0: no batch)."""
        frames = self.clip[0]
        n = min(self.ahead, len(frames) - first, limit)
        if n > 1:
            self.engine.encode_ahead(list(frames[first:first + n]), overlap=self.overlap)
            return n
        return 0

    def step(self, remaining=None):
        """Propagates frame t.  `remaining` = frames still to come in the caller's window (this one included): the batches
        encoded ahead never reach past it."""
        frames = self.clip[0]
        left = remaining if remaining is not None else len(frames)
        with torch.cuda.stream(self.stream):
            if self.ahead > 1:
                if self._encoded == 0:
                    self._encoded = self._issue(self.t, left)
                if self.overlap and self._encoded and self._prefetched == 0:      # the batch after this one, beside its propagation
                    self._prefetched = self._issue(self.t + self._encoded, left - self._encoded)
            label = one_frame(self.engine, frames[self.t], self.feedback)
        self._encoded = max(0, self._encoded - 1)
        if self._encoded == 0 and self._prefetched:
            self._encoded, self._prefetched = self._prefetched, 0
        self.t += 1
        return label

    def advance_to(self, t):
        while self.t < t:
            self.step(t - self.t)


def plan_windows(steps, streams, frames=CLIP_FRAMES - 1):
    """Which frames of a clip the `steps` timed frames are.  Returns a list of passes; a pass is a list of rounds; a round
    holds one (first_frame, count) window per stream.  Every pass uses a fresh clip per stream; within a pass the
    windows are spread evenly over frames 1..`frames` so that the timed frames sample the clip's bank-size
    distribution (M = 1 + (t-1)//gap) whatever `steps` is: a full pass is the whole clip on every stream, a short run
    is a few windows centred on equally spaced quantiles of the clip."""
    passes, left = [], steps
    while left > 0:
        kp = min(left, streams * frames)
        left -= kp
        rounds = 1 if kp >= streams * frames else max(1, round(3 / streams))
        nw = rounds * streams
        base, extra = divmod(kp, nw)
        wins = []
        for j in range(nw):
            n = base + (1 if j < extra else 0)
            centre = 1 + (j + 0.5) / nw * frames
            first = max(1, min(int(round(centre - n / 2.0)), frames + 1 - n))
            wins.append((first, n))
        passes.append([[wins[r * streams + i] for i in range(streams)] for r in range(rounds)])
    return passes


def bank_frames_at(t, gap):
    """Frames in the long-term bank while frame t is matched (reference frame at step 0, one more every `gap` steps)."""
    return 1 + (t - 1) // gap


def attention_roofline(engine, clip, device):
    """Instrumented pass over one clip: HIP events on the launch stream around every long-term / self attention call of
    the LSTT / GPM stack -- aot_hip.attention = the MFMA kernel attn_fwd_d32_pipe_kernel (AOT), aot_hip.gated_attention =
    attn_fwd_wide_coop_kernel<8> (DeAOT), each plus the small partial-merge launch when the bank is long enough to be
    split over workgroups.  bench.py wraps the binding; the product carries no hook."""
    import aot_hip
    recs = []
    deaot = 'deaot' in MODEL
    x6 = getattr(engine, 'mfma', 'f32') == 'bf16x6'
    name = ('gated_attention' if deaot else 'attention') + ('_x6' if x6 else '')
    real = getattr(aot_hip, name)

    def timed(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        r = real(*a, **kw)
        e1.record(torch.cuda.current_stream())
        q = a[0]
        kvb = 6.0 if x6 else 4.0     # bytes per bank element: fp32, or three bf16 planes
        if deaot and x6:    # (q, bank, gate, out, T, scale)
            T, dv = a[4], a[3].shape[1]
            flop = 2.0 * q.shape[0] * T * (q.shape[1] + dv)
            byts = kvb * T * kw.get('B', 1) * (q.shape[1] + dv) + 4.0 * q.shape[0] * (q.shape[1] + 2 * dv)
        elif deaot:         # (q, k, v, gate, out, T, scale): 2*N*T*(128 + dv) FLOP; bytes: K and V rows once, q / gate / out
            T, dv = a[5], a[4].shape[1]
            flop = 2.0 * q.shape[0] * T * (q.shape[1] + dv)
            byts = 4.0 * (T * kw.get('B', 1) * (q.shape[1] + dv) + q.shape[0] * (q.shape[1] + 2 * dv))
        elif x6:            # (q, bank, out, T, H, scale)
            T, H = a[3], a[4]
            flop = 4.0 * q.shape[0] * T * H * 32
            byts = (2 * kvb * T * kw.get('B', 1) + 8.0 * q.shape[0]) * H * 32
        else:               # (q, k, v, out, T, H, scale): 4*N*T*C FLOP; bytes 8*T*C + 8*N*C
            T, H = a[4], a[5]
            flop = 4.0 * q.shape[0] * T * H * 32
            byts = 8.0 * (T * kw.get('B', 1) + q.shape[0]) * H * 32
        recs.append((e0, e1, flop, byts))
        return r
    # ... and around every conv / linear call (the GEMM family: aot_hip.conv2d -- which aot_hip.linear and the fp32 fall-back of the
    # stem go through --, the four-channel stem entry and the GroupNorm-partials linear): 2 * M * K * N FLOP each
    grecs, depth = [], [0]
    greal = {n: getattr(aot_hip, n) for n in ('conv2d', 'conv2d_c4', 'linear_gn_x6')}

    def gemm_timed(fname, flop_of):
        fn = greal[fname]

        def run(*a, **kw):
            if depth[0]:                     # conv2d reached from conv2d_c4's fall-back: already inside a timed call
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            depth[0] += 1
            e0.record(torch.cuda.current_stream())
            try:
                r = fn(*a, **kw)
            finally:
                depth[0] -= 1
            e1.record(torch.cuda.current_stream())
            grecs.append((e0, e1, flop_of(*a, **kw)))
            return r
        return run

    def conv_flop(x, w, bias, out, H, W, Cin, OH, OW, Cout, KH=1, KW=1, *a, **kw):
        return 2.0 * kw.get('B', 1) * OH * OW * KH * KW * Cin * Cout

    def c4_flop(x, w, bias, out, H, W, OH, OW, Cout, KH, KW, *a, **kw):
        return 2.0 * kw.get('B', 1) * OH * OW * KH * KW * 3 * Cout       # (the fourth input channel is padding)

    def lgn_flop(x, w, bias, out, *a, **kw):
        return 2.0 * x.shape[0] * x.shape[1] * out.shape[1]
    frames, mask, objs = clip
    engine.restart_engine()
    engine.add_reference_frame(frames[0], mask, objs, frame_step=0)
    setattr(aot_hip, name, timed)
    for fname, fl in (('conv2d', conv_flop), ('conv2d_c4', c4_flop), ('linear_gn_x6', lgn_flop)):
        setattr(aot_hip, fname, gemm_timed(fname, fl))
    try:
        for t in range(1, len(frames)):
            one_frame(engine, frames[t])
    finally:
        setattr(aot_hip, name, real)
        for fname, fn in greal.items():
            setattr(aot_hip, fname, fn)
    torch.cuda.synchronize(device)
    gms = sum(a.elapsed_time(b) for a, b, _ in grecs)
    gflop = sum(f for _, _, f in grecs)
    gtf = gflop / max(gms, 1e-9) / 1e9
    gemm = {'bound': 'mfma', 'achieved': round(gtf, 2), 'unit': 'TFLOP/s (fp32-equivalent)' if x6 else 'TFLOP/s',
            'peak': round(BF16_MFMA_PEAK_TF / 6.0, 1) if x6 else FP32_MFMA_PEAK_TF,
            'frac': round(gtf * 6.0 / BF16_MFMA_PEAK_TF if x6 else gtf / FP32_MFMA_PEAK_TF, 4),
            'kernel': ('gemm_x6rd_kernel and the other members of the bf16x6 conv / linear family' if x6 else
                       'gemm_lean_kernel and the other members of the fp32 conv / linear family'),
            'what': 'every conv / linear launch of the propagated frames of one clip, one clip at a time, no encoder look-ahead, host '
                    'launches: HIP events around each call (a call = the GEMM kernel plus, for the split-K forms, its reduce launch)',
            'launches': len(grecs), 'launches_per_frame': round(len(grecs) / max(1, len(frames) - 1), 1),
            'us_per_frame': round(gms * 1e3 / max(1, len(frames) - 1), 1), 'gflop_per_frame': round(gflop / 1e9 / max(1, len(frames) - 1), 2),
            'traffic': None}
    ms = sum(a.elapsed_time(b) for a, b, _, _ in recs)
    flop = sum(f for _, _, f, _ in recs)
    n = len(recs)
    # HBM / fabric bytes per launch come from two separate rocprofv3 --pmc passes over the same launch mix (FETCH_SIZE x2 +
    # WRITE_SIZE, MI355X_MICROARCH.md); a PMC pass cannot run inside this process, so the figure is read from the committed
    # record of that run and `traffic_source` names it -- it is NOT measured in this run
    traffic = src = None
    tp = None
    for rnd in ('r06', 'r04', 'r03z', 'r03'):        # the newest committed record of the kernel as built
        cand = os.path.join('profiles', '%s_%sattn_%straffic.json' % (rnd, 'gated_' if deaot else '', 'x6_' if x6 else ''))
        if os.path.exists(os.path.join(ROOT, cand)):
            tp = cand
            break
    if tp:
        with open(os.path.join(ROOT, tp)) as f:
            traffic = round(json.load(f)['traffic_bytes_per_launch'])
        src = tp + ' (separate rocprofv3 --pmc passes, same launch mix; not measured in this run)'
    tf = flop / (ms * 1e-3) / 1e12
    if x6:
        # the bf16x6 kernels issue SIX bf16 MFMA products per fp32-equivalent product: priced against the dense bf16 peak that is
        # 6 x the algorithmic FLOPs over 2500 TFLOP/s -- the same fraction as the algorithmic rate over 2500 / 6
        return {'bound': 'mfma', 'achieved': round(tf, 2), 'peak': round(BF16_MFMA_PEAK_TF / 6.0, 1),
                'unit': 'TFLOP/s (fp32-equivalent: algorithmic FLOPs; the kernel issues 6 bf16 MFMA products per product)',
                'frac': round(tf * 6.0 / BF16_MFMA_PEAK_TF, 4), 'bf16_mfma_tflops_issued': round(6.0 * tf, 1),
                'peak_bf16_dense': BF16_MFMA_PEAK_TF, 'traffic': traffic, 'traffic_source': src,
                'kernel': 'attn_x6_wide64p_kernel' if deaot else 'attn_x6_d32_kernel', 'launches': n,
                'avg_launch_us': round(ms * 1e3 / n, 2), 'gflop_per_launch': round(flop / n / 1e9, 3),
                'algorithmic_bytes_per_launch': round(sum(b for _, _, _, b in recs) / n), 'gemm': gemm}
    return {'bound': 'mfma', 'achieved': round(tf, 2), 'peak': FP32_MFMA_PEAK_TF,
            'unit': 'TFLOP/s', 'frac': round(tf / FP32_MFMA_PEAK_TF, 4), 'traffic': traffic,
            'traffic_source': src,
            'kernel': 'attn_fwd_wide_coop_kernel<8>' if deaot else 'attn_fwd_d32_pipe_kernel', 'launches': n,
            'avg_launch_us': round(ms * 1e3 / n, 2), 'gflop_per_launch': round(flop / n / 1e9, 3),
            'algorithmic_bytes_per_launch': round(sum(b for _, _, _, b in recs) / n), 'gemm': gemm}


# golden clips of the real reference, whole 70-frame clips (tests/golden/make_golden.py): model -> (fixture, synthetic clip id).
# All three run free-running on the engine's own labels (round 3 had to tie-synchronise SwinB-DeAOTL; round 4 calibrated the
# Swin trunk of the synthetic weights, utils/synth.py::_swin_out_norm_gain, so that the reference itself is stable there).
JF_GOLDEN = {'r50_aotl': ('c2_r50_aotl_70', 0), 'r50_deaotl': ('c3b_r50_deaotl_70', 2),
             'swinb_deaotl': ('c3_swinb_deaotl_480_70', 10)}


def jf_vs_reference(device, graph=False, gemm_table='latency', ahead=1, mfma='f32'):
    """J&F of this engine's FREE-RUNNING masks against the real reference's masks on the committed golden clip of the
    benched model (BASELINE config 2: tests/golden/c2_r50_aotl_70.npz, R50-AOTL, 481x849, 10 objects, 69 propagated
    frames), run in EXACTLY the configuration the timed region used: same GEMM dispatch table, same launch mode (hipGraph
    replay or host launches), labels from aot_hip.fuse_probs.  Every differing pixel is checked against the reference's own
    argmax near-tie map stored with the golden (top-2 logit gap < 2e-4, `gapmask_<t>`): `pixels_outside_near_ties` must be
    0 -- main() exits non-zero otherwise."""
    import numpy as np
    from utils.metric import jf_per_object
    from utils.synth import synth_clip
    if MODEL not in JF_GOLDEN:
        return None
    name, clip_id = JF_GOLDEN[MODEL]
    gp = os.path.join(ROOT, 'tests', 'golden', name + '.npz')
    if not os.path.exists(gp):
        return None
    g = np.load(gp)
    gold = g['masks']
    cfg, model, engine, _ = build_model(device, graph, gemm_table, mfma)
    frames, mask, objs, _ = synth_clip(clip_id, gold.shape[0] + 1, IN_SIZE, OUT_SIZE, NUM_OBJ, device=device)
    run = StreamClip(engine, torch.cuda.current_stream(device), (frames, mask, objs))
    run.ahead = ahead
    run.restart()
    js, fs, diff, outside = [], [], 0, 0
    # the reference's own fp64 run of the same frames (tests/golden/make_fp64_ties.py): where do this run's flips sit on it?
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import fp64_ties
    f64, on64 = fp64_ties.load_fp64_ties(name), {}
    refs = torch.from_numpy(gold.astype('int64')).to(device)
    npix = gold.shape[1] * gold.shape[2]
    on_device = True         # the metric's reductions and dilations run where the masks are; host fallback if the device refuses
    tie_of = lambda t: torch.from_numpy(np.unpackbits(g['gapmask_%d' % t])[:npix].reshape(gold.shape[1:]).astype(bool)).to(device)
    for t in range(1, len(frames)):
        label = run.step(len(frames) - t)[0, 0].long()
        ref = refs[t - 1]
        if on_device:
            try:
                j, f = jf_per_object(label, ref, NUM_OBJ)
            except Exception as exc:
                on_device = False
                print('[bench] J&F metric falls back to the host: %s' % exc, file=sys.stderr, flush=True)
        if not on_device:
            j, f = jf_per_object(label.cpu(), ref.cpu(), NUM_OBJ)
        js.append(j)
        fs.append(f)
        bad = label != ref
        nbad = int(bad.sum())
        if nbad:
            outside += int((bad & ~tie_of(t)).sum())
            if f64 is not None:
                fp64_ties.add_counts(on64, fp64_ties.classify_flips(f64, t, label.cpu().numpy(), gold[t - 1]))
        diff += nbad
    J, Fm = sum(js) / len(js), sum(fs) / len(fs)
    return {'J': round(J, 6), 'F': round(Fm, 6), 'J&F': round((J + Fm) / 2, 6), 'frames': len(js),
            'pixels_differing': diff, 'pixels_outside_near_ties': outside, 'of_pixels': int(gold.size),
            # of the differing pixels: on how many does the REFERENCE's own argmax differ between its fp32 and fp64 runs (and this
            # engine carries the fp64 id), how many of the others have an fp64 top-2 gap below 5e-5 / 1e-4 / 2e-4 (smallest bin)
            'flips_on_the_fp64_reference': None if f64 is None else (on64 or {'flips': 0}),
            'reference_fp32_vs_fp64_flips': None if f64 is None else int(f64['stats'][:, 2].sum()),
            'gemm_table': gemm_table, 'mfma': mfma, 'launch': 'hipGraph replay' if graph else 'host launches',
            'labels': 'engine.decode_current_labels (aot_frame_tail_f32: resize + softmax + argmax + nearest feedback in the decode replay)',
            'encode_ahead_frames': ahead,
            'feedback': 'own labels',
            'clip': 'tests/golden/%s.npz (free-running, masks of the real reference; near-tie = top-2 logit gap < 2e-4 in the '
                    'reference)' % name}


def cpu_baseline(sd, budget_s=20.0, max_frames=12):
    """Oracle (CPU port of the reference algorithm) on the same clip, same FPS definition, on a BOUNDED sample: frames 1..12
    with the long-term gap set to 1, so that the memory bank grows to M = 12 inside the sample and the timed frames see the
    bank sizes of a whole clip (sample mean M = 6.5; the GPU run's timed mean is 7.2-7.4) instead of the M <= 3 of a clip's
    first twelve frames at gap 5.  The per-frame work is the same algorithm at the same shapes; only which frames are
    memorised differs from the GPU run."""
    from oracle.aot_oracle import OracleEngine, OracleModel
    from utils.synth import synth_clip
    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    frames, mask, objs, _ = synth_clip(0, max_frames + 1, IN_SIZE, OUT_SIZE, NUM_OBJ)
    eng = OracleEngine(OracleModel(MODEL, {k: v.cpu() for k, v in sd.items()}), long_term_mem_gap=1)
    done, spent, msum = 0, 0.0, 0
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
            msum += t               # frames in the bank while frame t is matched (reference frame + t - 1 memorised ones)
            if spent > budget_s:
                break
    return {'value': round(done / spent, 3), 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
            'sample': 'frames 1..%d of clip 0 (%dx%d, %d objects) with the long-term gap set to 1: bank M = 1..%d, mean %.1f (the '
                      'timed GPU frames: see config.timed_M_mean); oracle/aot_oracle.py fp32, %d torch threads'
                      % (done, IN_SIZE[0], IN_SIZE[1], NUM_OBJ, done, msum / max(done, 1), threads)}


def other_config_legs(args):
    """BASELINE configs[2] (SwinB-DeAOTL, 480x848) and R50-DeAOTL measured like the headline model, each in a process of its own
    (`bench.py --model ... --leg`: own weights, banks and graphs; a failure there cannot take the headline line down).  Returns
    {model: summary of that run's JSON line}."""
    import subprocess
    out = {}
    for name in ('swinb_deaotl', 'r50_deaotl'):
        cmd = [sys.executable, os.path.abspath(__file__), '--model', name, '--leg', '--gpus', '1', '--steps', str(args.steps),
               '--warmup', str(args.warmup), '--streams', str(args.streams), '--graph', str(args.graph), '--repeats', str(args.repeats),
               '--encode-ahead', str(args.encode_ahead), '--overlap-encode', str(args.overlap_encode), '--mfma', args.mfma]
        for flag in ('no_cpu_baseline', 'no_roofline', 'no_jf', 'no_whole_clip'):
            if getattr(args, flag):
                cmd.append('--' + flag.replace('_', '-'))
        t0 = time.perf_counter()
        try:
            pr = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            rows = [ln for ln in pr.stdout.splitlines() if ln.startswith('{')]
            if not rows:
                raise RuntimeError('no JSON line (rc %d): %s' % (pr.returncode, pr.stderr.strip()[-300:]))
            d = json.loads(rows[-1])
            c = d['config']
            out[name] = {'workload': c['workload'], 'fps': d['value'], 'repeat_fps': c['repeat_fps'], 'ms_per_step': d['ms_per_step'],
                         'dtype': d['dtype'], 'timed_M_mean': c['timed_M_mean'],
                         'single_stream_fps': (c.get('single_stream') or {}).get('fps'),
                         'whole_clip_fps': (c.get('whole_clip') or {}).get('fps'),
                         'jf_vs_reference': c.get('jf_vs_reference'), 'roofline': d.get('roofline'),
                         'cpu_baseline': d.get('cpu_baseline'), 'rc': pr.returncode}
        except Exception as e:           # noqa: BLE001
            out[name] = {'error': '%s: %s' % (type(e).__name__, e)}
        print('[bench] other_configs.%s: %.1f s' % (name, time.perf_counter() - t0), file=sys.stderr, flush=True)
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: re-exec this file under torch.distributed.run, one rank per GPU
    (what tools/eval.py:100-106 does with mp.spawn).  Fails loudly when fewer than N devices are visible."""
    import subprocess
    if not args.dry_run and not args.share_gpu:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit('bench.py --gpus %d: only %d ROCm device(s) visible' % (args.gpus, have))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


class _DryClip:
    """--dry-run stand-in for StreamClip (CPU launcher test, tests/test_sharding_gloo.py): same stepping protocol, no
    device work.  A dry run prints value = null."""

    def __init__(self):
        self.t = None

    def restart(self):
        self.t = 1

    def step(self, remaining=None):
        self.t += 1

    def drop_ahead(self):
        pass

    def advance_to(self, t):
        self.t = max(self.t, t)


def main(argv=None):
    global MODEL, IN_SIZE, OVERLAP_ENCODE
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3 * (CLIP_FRAMES - 1), help='propagated frames timed per GPU (default: one full 70-frame clip per stream)')
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--streams', type=int, default=3, help='clips processed concurrently per GPU (one HIP stream each)')
    ap.add_argument('--graph', type=int, default=1, choices=[0, 1],
                    help='1 (default): the engines replay one captured hipGraph per frame stage and engine state '
                         '(networks/engines/graphs.py); 0: every kernel launched from the host')
    ap.add_argument('--model', default=MODEL, choices=['r50_aotl', 'r50_deaotl', 'swinb_deaotl', 'swinb_aotl', 'r101_aotl'],
                    help='default: R50-AOTL = BASELINE configs[1], the configuration the metric is quoted on; swinb_deaotl = '
                         'configs[2] (480x848 input).  roofline = the long-term attention kernel of the model family; the J&F leg '
                         'runs for the models that have a whole-clip golden (r50_aotl, r50_deaotl, swinb_deaotl).')
    ap.add_argument('--encode-ahead', type=int, default=3,
                    help='K > 1 (default 3): the encoder runs over the next K frames of a clip as one batch on the clip\'s own '
                         'stream (engine.encode_ahead; the encoder does not depend on the mask feedback), never past the end of '
                         'a timed window and never before its start; 1: every frame is encoded when it is matched')
    ap.add_argument('--overlap-encode', type=int, default=-1, choices=[-1, 0, 1],
                    help='1: the look-ahead batch AFTER the one being propagated is encoded meanwhile on a side stream of the engine '
                         '(engine.encode_ahead(..., overlap=True)): the encoder of the coming frames fills the CUs the stride-16 stages '
                         'of the propagated frame leave idle; same kernels, bit-identical results.  -1 (default): only where one clip '
                         'runs at a time (--streams 1 and the single_stream leg; profiles/r06_group_overlap_ab.txt: +9 %% there, -3 %% '
                         'with three clips per GPU, whose other clips already fill those CUs); 0: never')
    ap.add_argument('--mfma', default='bf16x6', choices=['f32', 'bf16x6'],
                    help="matrix-core arithmetic of the run: 'bf16x6' (default since round 4: the fp32-equivalent six-term bf16 split -- "
                         "every fp32 operand as three truncated bf16 numbers, six of the nine partial products, fp32 accumulation; dtype "
                         "'f32 via bf16x6 split') or 'f32' (exact fp32 products on v_mfma_f32_32x32x2_f32).  The OTHER arithmetic is "
                         "measured on the same plan as a leg of the line (config.fp32_exact / config.bf16x6_split)")
    ap.add_argument('--no-x6', '--no-second-arithmetic', dest='no_x6', action='store_true',
                    help='skip the leg that measures the other arithmetic (throughput and J&F of the second kernel family)')
    ap.add_argument('--repeats', type=int, default=3,
                    help='the timed window plan of --steps frames is run this many times; `value` is the MEDIAN run, all runs are '
                         'listed in config.repeat_fps (a --steps 20 window is ~40 ms: one run is not the whole story)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip config.other_configs of a default single-GPU run: the same measurement (throughput, one clip at a time, '
                         'J&F on the whole-clip golden, roofline of the gated kernel, CPU baseline) for BASELINE configs[2] '
                         '(SwinB-DeAOTL, 480x848) and R50-DeAOTL, each in its own process')
    ap.add_argument('--no-whole-clip', action='store_true',
                    help='skip config.whole_clip: the timed plan over WHOLE 70-frame clips (69 propagated frames per stream) next to '
                         'the --steps window')
    ap.add_argument('--leg', action='store_true', help=argparse.SUPPRESS)      # this process is an other_configs leg of another run
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-jf', action='store_true', help='skip the J&F pass on the committed reference clip (tuning runs)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help='nccl = RCCL (default); gloo only with --dry-run')
    ap.add_argument('--dry-run', action='store_true', help='launcher / sharding / gather plumbing only, no device work (CPU tests)')
    ap.add_argument('--share-gpu', action='store_true',
                    help='development only: every rank on GPU 0, collectives over gloo -- walks the N > 1 control flow WITH device '
                         'work on a one-GPU box (tools/dev/r04_call27.sh); the numbers of such a run mean nothing and say so')
    args = ap.parse_args(argv)
    if args.gpus < 1 or args.steps < 1:
        raise SystemExit('bench.py: --gpus and --steps must be >= 1')
    default_model = args.model == MODEL
    MODEL = args.model
    OVERLAP_ENCODE = args.overlap_encode
    if MODEL.startswith('swinb'):
        IN_SIZE = (480, 848)          # align_corners = False models take multiples of 16 (video_transforms.py:640-655)
    if args.backend == 'gloo' and not args.dry_run:
        raise SystemExit('bench.py: the gloo backend is only for --dry-run (the hot path has no CPU fallback)')

    if 'WORLD_SIZE' not in os.environ:
        if args.gpus > 1:
            raise SystemExit(spawn_ranks(args, argv))
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(os.environ['WORLD_SIZE'])
        rank = int(os.environ.get('RANK', '0'))
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        if world != args.gpus:      # never emit a scaling number for a world the caller did not ask for
            raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    dry = args.dry_run
    if not dry and not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU: the AOT hot path has no CPU fallback')
    device = torch.device('cpu') if dry else torch.device('cuda', 0 if args.share_gpu else local_rank)
    if not dry:
        torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if dry or args.share_gpu:
            dist.init_process_group('gloo' if args.share_gpu else args.backend, rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)   # RCCL over xGMI
    joined = dist.get_world_size() if world > 1 else 1
    host = pin_host(rank, world, local_rank, dry)

    def sync():
        if not dry:
            torch.cuda.synchronize(device)

    events = []                # --dry-run: the order of this rank's phases (tests/test_sharding_gloo.py)

    def fence(collective=True):
        sync()
        if world > 1 and collective:
            COLLECTIVES['barrier'] += 1
            dist.barrier()
        sync()

    def reduce_max(t):
        if world > 1:
            COLLECTIVES['all_reduce'] += 1
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t

    S = max(1, args.streams)
    table = 'throughput' if S > 1 else 'latency'      # several clips per GPU keep the chip saturated: cheapest kernel in SIMD time
    passes = plan_windows(args.steps, S)
    nclips_rank = S * len(passes)
    my_ids = shard_clips(nclips_rank * world, rank, world)          # clip i -> rank i mod world
    gap = 5
    sd = None
    if dry:
        lanes = [_DryClip() for _ in range(S)]
        set_clip = lambda lane, j: None
    else:
        from networks.engines import build_engine
        from utils.synth import synth_clip
        cfg, model, engine, sd = build_model(device, bool(args.graph), table, args.mfma)
        gap = cfg.TEST_LONG_TERM_MEM_GAP
        if hasattr(model, 'prepare'):
            model.prepare()        # pack the weights once, before the clips fan out over streams
        clips = []
        for cid in my_ids:
            frames, mask, objs, _ = synth_clip(cid, CLIP_FRAMES, IN_SIZE, OUT_SIZE, NUM_OBJ, device=device)
            clips.append((frames, mask, objs))

        def new_engine(tbl, graph=bool(args.graph), mfma=args.mfma):
            return build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=model, gpu_id=device.index or 0, long_term_mem_gap=gap,
                                graph=graph, gemm_table=tbl, mfma=mfma)
        engines = [engine] + [new_engine(table) for _ in range(S - 1)]
        streams = [torch.cuda.Stream(device) for _ in range(S)]
        lanes = [StreamClip(engines[i], streams[i], clips[i]) for i in range(S)]
        for lane in lanes:
            lane.ahead = max(1, args.encode_ahead)
            lane.overlap = args.overlap_encode == 1 or (args.overlap_encode < 0 and S == 1)

        def set_clip(lane, j):
            lane.clip = clips[j]

    def run_plan(lanes, passes, clip_of, collective=True):
        """Runs the window plan; returns (timed seconds, timed frames, sum of bank sizes over the timed frames).  Only
        the windows are timed: restart + reference frame and the fast-forward between windows are set-up."""
        spent, done, msum = 0.0, 0, 0
        for pi, rounds in enumerate(passes):
            for i, lane in enumerate(lanes):
                set_clip(lane, clip_of(pi, i))
                lane.restart()
            for wins in rounds:
                for lane, (first, n) in zip(lanes, wins):
                    lane.advance_to(first)
                    lane.drop_ahead()
                fence(collective)
                t0 = time.perf_counter()
                for k in range(max(n for _, n in wins)):
                    for lane, (first, n) in zip(lanes, wins):
                        if k < n:
                            msum += bank_frames_at(lane.t, gap)
                            lane.step(n - k)
                            done += 1
                fence(collective)
                spent += time.perf_counter() - t0
        return spent, done, msum

    def median_run(runs):
        """(elapsed, frames, msum) of the median-throughput run (the lower middle one for an even count)."""
        order = sorted(range(len(runs)), key=lambda i: runs[i][0] / runs[i][1])
        return runs[order[(len(runs) - 1) // 2]]

    R = max(1, args.repeats)
    t_phase = [time.perf_counter()]

    def phase(name):
        now = time.perf_counter()
        if rank == 0:
            print('[bench] %s: %.1f s' % (name, now - t_phase[0]), file=sys.stderr, flush=True)
        t_phase[0] = now
    with torch.no_grad():
        # priming (set-up, untimed): one full clip per stream so the caching allocator, the per-stream scratch and the
        # memory banks have reached their steady-state size -- a growing allocator calls hipMalloc, which
        # synchronises the device and serialises the concurrently running clips
        for i, lane in enumerate(lanes):
            lane.restart()
        for t in range(1, CLIP_FRAMES):
            for lane in lanes:
                lane.step()
        phase('model build + priming (one clip per stream)')
        # warmup: W frames spread over the streams (fresh clip state; the plan below restarts every lane anyway)
        for lane in lanes:
            lane.restart()
        for i in range(max(args.warmup, 0)):
            lanes[i % S].step()
        # the timed plan, R times; every run is reduced to the max over ranks, `value` is the median run
        runs = []
        for r in range(R):
            elapsed, frames_done, msum = run_plan(lanes, passes, lambda pi, i: pi * S + i)
            assert frames_done == args.steps
            tm = reduce_max(torch.tensor([elapsed], dtype=torch.float64, device=device))
            runs.append((float(tm.item()), frames_done, msum, elapsed))
            events.append('timed_run_%d' % r)
        tmax, frames_done, msum, own_elapsed = median_run(runs)
        elapsed = tmax
        phase('warm-up + %d timed runs' % R)
        single = None
        if S > 1 and rank == 0 and not dry:      # the same job one clip at a time (the reference's evaluation mode)
            one = StreamClip(new_engine('latency'), streams[0], clips[0])
            one.ahead = lanes[0].ahead
            one.overlap = args.overlap_encode != 0
            one.restart()                        # untimed: one clip under the latency table (graph mode captures its states)
            for t in range(1, CLIP_FRAMES):
                one.step()
            # timed over the WHOLE clip (69 propagated frames, as the reference evaluates a sequence and as the online figure below),
            # whatever --steps is: in the short windows of the driver's 20-frame form the overlapped look-ahead has nothing to run beside
            # for a third of the frames (it never reaches past a window's end); `windows_fps` keeps rounds 2-5's form for continuity
            sruns = [run_plan([one], plan_windows(CLIP_FRAMES - 1, 1), lambda pi, i: 0, collective=False) for _ in range(R)]
            e1, f1, m1 = median_run(sruns)
            wruns = [run_plan([one], plan_windows(args.steps, 1), lambda pi, i: 0, collective=False) for _ in range(R)]
            ew, fw, _ = median_run(wruns)
            single = {'fps': round(f1 / e1, 2), 'repeat_fps': [round(f / e, 2) for e, f, _ in sruns], 'frames': int(f1),
                      'windows_fps': round(fw / ew, 2), 'windows_frames': int(fw),
                      'timed_M_mean': round(m1 / f1, 2), 'gemm_table': 'latency', 'encode_ahead_frames': one.ahead,
                      'encode_overlapped': bool(one.overlap and one.ahead > 1)}
            del one
            # ... and STRICTLY ONLINE, timed the reference's way (evaluator.py:325-330, 444-446, 486-498): one clip, no encoder
            # look-ahead (every frame is encoded when it arrives), a device event just before match_propogate_one_frame and one
            # just after update_memory, All-Frame FPS = propagated frames / sum of the per-frame event times over the whole clip;
            # the reference frame is outside, as there
            onl = StreamClip(new_engine('latency'), streams[0], clips[0])
            onl.ahead = 1
            oruns = []
            for r in range(R + 1):               # the first pass captures the graphs (untimed)
                onl.restart()
                evs = []
                for t in range(1, CLIP_FRAMES):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(streams[0])
                    onl.step()
                    e1.record(streams[0])
                    evs.append((e0, e1))
                torch.cuda.synchronize(device)
                if r:
                    oruns.append(sum(a.elapsed_time(b) for a, b in evs) * 1e-3)
            oruns.sort()
            single['online'] = {'fps': round((CLIP_FRAMES - 1) / oruns[(len(oruns) - 1) // 2], 2),
                                'repeat_fps': [round((CLIP_FRAMES - 1) / e, 2) for e in oruns], 'frames': CLIP_FRAMES - 1,
                                'encode_ahead_frames': 1, 'gemm_table': 'latency',
                                'timing': 'device events around each propagated frame (match -> decode -> memory update), summed '
                                          'over the whole clip: the All-Frame FPS of evaluator.py:325-330,444-446,486-498'}
            del onl

        phase('single-stream leg')
        whole = None
        if not args.no_whole_clip and args.steps != S * (CLIP_FRAMES - 1):
            # the same lanes over whole clips (every propagated frame of a clip timed, M 1 -> 14): what `--steps 207` measures,
            # carried by every line so that a short --steps window is never the only throughput figure (collective: all ranks)
            full = plan_windows(S * (CLIP_FRAMES - 1), S)
            wruns = []
            for r in range(R):
                ew, fw, mw = run_plan(lanes, full, lambda pi, i: i)
                tw = reduce_max(torch.tensor([ew], dtype=torch.float64, device=device))
                wruns.append((float(tw.item()), fw, mw))
                events.append('whole_clip_run_%d' % r)
            ew, fw, mw = median_run(wruns)
            whole = {'fps': None if dry else round(world * fw / ew, 2),
                     'repeat_fps': [None if dry else round(world * f / e, 2) for e, f, _ in wruns],
                     'frames_per_gpu': fw, 'ms_per_frame': round(ew / fw * 1e3, 3), 'timed_M_mean': round(mw / fw, 2)}

        phase('whole-clip leg')
        x6 = None
        other = 'f32' if args.mfma == 'bf16x6' else 'bf16x6'
        other_dtype = 'f32' if other == 'f32' else 'f32 via bf16x6 split'
        if not args.no_x6 and not args.leg and rank == 0 and not dry:
            try:                                  # (an optional leg: its failure must not cost the headline measurement)
                # the other kernel family on the same plan (same clips, streams, table, graphs): a separate number under its
                # own dtype string, next to -- never instead of -- `value`
                xl = [StreamClip(new_engine(table, mfma=other), streams[i], clips[i]) for i in range(S)]
                for lane in xl:
                    lane.ahead = lanes[0].ahead
                    lane.restart()
                for t in range(1, CLIP_FRAMES):          # untimed: packs the split weights, captures the graphs
                    for lane in xl:
                        lane.step()
                xruns = [run_plan(xl, passes, lambda pi, i: pi * S + i, collective=False) for _ in range(R)]
                ex, fx, _ = median_run(xruns)
                x6 = {'dtype': other_dtype, 'value': round(fx / ex, 2), 'repeat_fps': [round(f / e, 2) for e, f, _ in xruns],
                      'n_gpus': 1, 'what': X6_WHAT % aot_hip_x6_min_tiles() if other == 'bf16x6' else
                      'every product exact in fp32: conv / linear on aot_conv2d_nhwc_f32, attention on aot_attn_f32 / aot_gated_attn_f32 '
                      '(v_mfma_f32_32x32x2_f32)'}
                del xl
            except Exception as e:           # noqa: BLE001
                x6 = {'dtype': other_dtype, 'error': '%s: %s' % (type(e).__name__, e)}
                print('[bench] %s leg failed: %s' % (other, x6['error']), file=sys.stderr, flush=True)
        phase('second-arithmetic leg (%s)' % other)
        peak = 0.0 if dry else torch.cuda.max_memory_allocated(device) / 2**30
        stats = gather_stats(torch.tensor([elapsed, float(frames_done), peak, float(msum), own_elapsed], dtype=torch.float64,
                                          device=device), world)

        roof = None
        if rank == 0 and not args.no_roofline and not dry:
            # the instrumented pass launches every kernel from the host (events cannot sit inside a replayed graph)
            probe = new_engine(table, graph=False)
            with torch.cuda.stream(streams[0]):
                roof = attention_roofline(probe, clips[0], device)
                if not args.no_x6 and not args.leg:
                    try:            # the same pass for the other arithmetic's attention kernel (sub-object: never the headline)
                        roof['other_arithmetic'] = attention_roofline(new_engine(table, graph=False, mfma=other), clips[0], device)
                    except Exception as e:       # noqa: BLE001
                        roof['other_arithmetic'] = {'error': '%s: %s' % (type(e).__name__, e)}

    phase('roofline pass')
    base = jf = None
    events.append('rank0_legs')          # everything below runs on rank 0 only, after the last timed barrier of every rank
    if rank == 0 and not dry:
        t_ph = time.perf_counter()
        if not args.no_jf:
            with torch.no_grad():
                jf = jf_vs_reference(device, bool(args.graph), table, max(1, args.encode_ahead), args.mfma)
                if x6 is not None and 'error' not in x6:
                    try:
                        x6['jf_vs_reference'] = jf_vs_reference(device, bool(args.graph), table, max(1, args.encode_ahead), other)
                    except Exception as e:       # noqa: BLE001
                        x6['jf_error'] = '%s: %s' % (type(e).__name__, e)
            print('[bench] J&F pass on the golden clip: %.1f s' % (time.perf_counter() - t_ph), file=sys.stderr, flush=True)
        t_ph = time.perf_counter()
        if not args.no_cpu_baseline and world == 1:
            base = cpu_baseline(sd)          # rank 0 at N = 1 only (it takes every host core), after the timed region
            print('[bench] cpu_baseline: %.1f s' % (time.perf_counter() - t_ph), file=sys.stderr, flush=True)
    others = None
    if rank == 0 and not dry and world == 1 and default_model and not args.leg and not args.no_other_configs:
        others = other_config_legs(args)
    trace = None
    if dry:            # CPU plumbing test only: what every rank did, gathered as objects (not part of a real run)
        mine = {'rank': rank, 'clips': list(my_ids), 'collectives': dict(COLLECTIVES), 'events': list(events)}
        trace = [mine]
        if world > 1:
            trace = [None] * world
            dist.all_gather_object(trace, mine)
    if world > 1:
        dist.barrier()

    if rank == 0:
        total_frames = float(stats[:, 1].sum())
        line = {
            'metric': 'frames/sec, 480p 10-object synthetic clips; J&F vs reference',
            'value': None if dry else round(total_frames / tmax, 2), 'unit': 'frames/s', 'n_gpus': joined,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(tmax / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.mfma == 'f32' else 'f32 via bf16x6 split',
            'data': 'dry-run (no device work)' if dry else ('synthetic -- ALL RANKS ON ONE GPU (--share-gpu smoke run of the N > 1 control '
                                                            'flow): the numbers mean nothing') if args.share_gpu else 'synthetic',
            'config': {'workload': ('R50-AOTL inference, 480p (481x849 in, 480x854 out) 10-object synthetic clips, '
                                    '70 frames/clip, long-term gap 5 (configs[1])') if default_model else
                                   '%s inference, 480p (%dx%d in, 480x854 out) 10-object synthetic clips, 70 frames/clip'
                                   % (MODEL, IN_SIZE[0], IN_SIZE[1]),
                       'frames_per_clip': CLIP_FRAMES, 'clips_per_gpu': nclips_rank, 'streams_per_gpu': S,
                       'timed_M_mean': round(float(stats[:, 3].sum()) / total_frames, 2),
                       'repeats': R, 'repeat_fps': [None if dry else round(world * f / e, 2) for e, f, _, _ in runs],
                       'timed_windows': passes[0] if len(passes) == 1 else '%d passes x %s' % (len(passes), passes[0]),
                       'single_stream': single, 'single_stream_online': (single or {}).get('online'), 'whole_clip': whole,
                       'parallelism': 'clip-sharded dp%d x %d concurrent clips per GPU' % (joined, S),
                       # how `value` is formed, rank by rank (the first N > 1 run on hardware must be readable from its own line):
                       # value = sum(per_rank_frames) / max(per_rank_elapsed_s) of the median run; ranks_joined = the size of the process
                       # group every collective of this run went through (= n_gpus)
                       'ranks_joined': joined, 'per_rank_frames': [int(x) for x in stats[:, 1].tolist()],
                       'per_rank_elapsed_s': [round(float(x), 6) for x in stats[:, 4].tolist()],
                       'elapsed_max_s': round(tmax, 6),
                       'launch': 'hipGraph replay per frame stage' if args.graph else 'host launches',
                       'gemm_table': table,
                       'encode_ahead_frames': max(1, args.encode_ahead), 'encode_overlapped': bool(getattr(lanes[0], 'overlap', False) and lanes[0].ahead > 1),
                       'weights': 'keyed synthetic (utils/synth.py)', 'peak_mem_gib': round(float(stats[:, 2].max()), 2),
                       'timed_region': 'wall clock (barrier + device sync on both sides) over the propagated frames '
                                       'only: windows of consecutive frames spread over the 70-frame clip so that the '
                                       'bank size M is sampled like a whole clip (timed_M_mean; a whole clip is 7.41); '
                                       'restart + reference frame and the fast-forward between windows are untimed '
                                       'set-up, as in the reference FPS (evaluator.py:325-330,444-446)',
                       'host': host, 'jf_vs_reference': jf, ('fp32_exact' if args.mfma == 'bf16x6' else 'bf16x6_split'): x6,
                       'arithmetic': X6_WHAT % aot_hip_x6_min_tiles() if (args.mfma == 'bf16x6' and not dry) else None,
                       'other_configs': others, 'dry_trace': trace},
            'roofline': roof, 'cpu_baseline': base,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and x6 is not None and x6.get('jf_vs_reference') and x6['jf_vs_reference']['pixels_outside_near_ties'] > 0:
        print('[bench] second-arithmetic leg: %d mask pixels outside the reference near-ties' % x6['jf_vs_reference']['pixels_outside_near_ties'],
              file=sys.stderr, flush=True)
    if rank == 0 and jf is not None and jf['pixels_outside_near_ties'] > 0:
        # a timed configuration whose masks leave the reference's near-ties is not a valid measurement: fail loudly
        raise SystemExit('bench.py: %d mask pixels differ from the reference outside its argmax near-ties (%s)'
                         % (jf['pixels_outside_near_ties'], jf['clip']))


if __name__ == '__main__':
    main()

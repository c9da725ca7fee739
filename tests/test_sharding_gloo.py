"""N>1 path of bench.py on CPU: clips are partitioned over ranks with no data-path collective; one
all_gather of a small stats vector at the end (replaces the reference's mp.Queue, evaluator.py:507-531)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    clips = bench.shard_clips(7, rank, world)
    stats = torch.tensor([float(len(clips)), float(sum(clips)), 1.0 + rank], dtype=torch.float64)
    allst = bench.gather_stats(stats, world)
    dist.barrier()
    q.put((rank, clips, allst.tolist()))
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, c0, s0), (r1, c1, s1) = res
    assert sorted(c0 + c1) == list(range(7)) and not set(c0) & set(c1)      # disjoint cover
    assert abs(len(c0) - len(c1)) <= 1
    assert s0 == s1 and len(s0) == 2                                        # every rank sees both stat rows
    assert s0[0][2] == 1.0 and s0[1][2] == 2.0


def test_bench_entry_point_spawns_two_gloo_ranks():
    """`python bench.py --gpus 2` as the driver invokes it (no launcher, WORLD_SIZE unset): bench.py re-execs itself
    under torch.distributed.run with one rank per device and prints ONE line whose n_gpus is the number of ranks that
    joined the process group.  --dry-run/--backend gloo replace the device work, everything else is the real path
    (argument handling, window plan, clip sharding, barrier-bracketed timing, max-over-ranks, stats all_gather)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '5',
                        '--dry-run', '--backend', 'gloo'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 20 and line['warmup'] == 5
    assert line['value'] is None and 'dry-run' in line['data']          # a dry run never reports a throughput
    assert line['config']['parallelism'].startswith('clip-sharded dp2')
    assert 6.5 <= line['config']['timed_M_mean'] <= 8.3                   # a whole 70-frame clip is 7.41


def test_bench_refuses_a_world_it_was_not_asked_for():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '4', '--dry-run', '--backend', 'gloo'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=2' in (r.stderr + r.stdout)


def test_window_plan_samples_the_bank_size_distribution():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    full = sum(bench.bank_frames_at(t, 5) for t in range(1, 70)) / 69.0
    assert abs(full - 7.406) < 1e-3
    for steps, streams in ((20, 3), (20, 1), (207, 3), (500, 3), (69, 1), (40, 2)):
        frames, msum = 0, 0
        for rounds in bench.plan_windows(steps, streams):
            last = [0] * streams
            for wins in rounds:
                assert len(wins) == streams
                for i, (first, n) in enumerate(wins):
                    assert first >= max(1, last[i]) and first + n <= 70       # forward in time, inside the clip
                    last[i] = first + n
                    frames += n
                    msum += sum(bench.bank_frames_at(t, 5) for t in range(first, first + n))
        assert frames == steps
        assert abs(msum / frames - full) < 0.6, (steps, streams, msum / frames)


def test_bench_entry_point_world_8_dry_run():
    """The driver's 8-GPU launch (`python bench.py --gpus 8`), device work replaced by --dry-run: eight ranks join, the clips are
    eight disjoint shards (clip i -> rank i mod 8), every rank issues exactly ONE all_gather on the data path's behalf (the stats
    vector; reference: mp.Queue, evaluator.py:507-531) next to the barriers and the max-over-ranks reductions of the timing
    protocol, and the rank-0-only legs (J&F, x6 leg, CPU baseline) come after the last timed run of EVERY rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '20', '--warmup', '5',
                        '--dry-run', '--backend', 'gloo'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 8 and line['scaling'] == 'weak' and line['value'] is None
    # how the N > 1 figure is formed is readable from the line itself: every rank joined the group the collectives ran in, each
    # contributed --steps frames, and the time `value` divides by is the MAX over the ranks' own elapsed times of the median run
    cfg = line['config']
    assert cfg['ranks_joined'] == 8 and cfg['per_rank_frames'] == [line['steps']] * 8 and len(cfg['per_rank_elapsed_s']) == 8
    assert abs(cfg['elapsed_max_s'] - max(cfg['per_rank_elapsed_s'])) < 1e-5
    assert abs(line['ms_per_step'] - cfg['elapsed_max_s'] / line['steps'] * 1e3) < 2e-3
    assert line['cpu_baseline'] is None                                   # N = 1 only
    tr = line['config']['dry_trace']
    assert sorted(t['rank'] for t in tr) == list(range(8))
    shards = [t['clips'] for t in tr]
    flat = sorted(c for sh in shards for c in sh)
    assert flat == list(range(len(flat))) and len(set(flat)) == len(flat)     # disjoint cover of the clip set
    assert len({len(sh) for sh in shards}) == 1                                # equal work per rank: weak scaling
    for t in tr:
        assert all(c % 8 == t['rank'] for c in t['clips'])
        assert t['collectives']['all_gather'] == 1
        ev = t['events']
        assert ev[-1] == 'rank0_legs' and ev.index('rank0_legs') > max(i for i, e in enumerate(ev) if e.startswith(('timed', 'whole')))
        # same protocol on every rank: identical counts (a rank that skipped a barrier would hang, one that added one would differ)
        assert t['collectives'] == tr[0]['collectives'] and ev == tr[0]['events']
    assert tr[0]['collectives']['all_reduce'] == line['config']['repeats'] * 2      # the --steps plan and the whole-clip plan

"""How long does the host take to ISSUE one frame (async) vs how long the GPU takes to run it?"""
import sys, os, time, importlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch
import bench
dev = torch.device('cuda', 0)
cfg, model, eng, sd = bench.build_model(dev)
from utils.synth import synth_clip
frames, mask, objs, _ = synth_clip(0, 70, bench.IN_SIZE, bench.OUT_SIZE, bench.NUM_OBJ, device=dev)
with torch.no_grad():
    for rep in range(3):
        eng.restart_engine(); eng.add_reference_frame(frames[0], mask, objs, frame_step=0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(1, 70):
            bench.one_frame(eng, frames[t])
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print('rep %d: host issue %.3f ms/frame, total %.3f ms/frame' % (rep, (t1 - t0) / 69 * 1e3, (t2 - t0) / 69 * 1e3))
import cProfile, pstats
eng.restart_engine(); eng.add_reference_frame(frames[0], mask, objs, frame_step=0); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
with torch.no_grad():
    for t in range(1, 30): bench.one_frame(eng, frames[t])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)

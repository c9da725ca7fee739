import sys, os, time, importlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, torch.nn.functional as F
from networks.models import build_vos_model
from networks.engines import build_engine
from utils.synth import synth_state_dict, synth_clip
name = 'r50_aotl'; T = 70
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = importlib.import_module('configs.models.' + name).ModelConfig()
model = build_vos_model(cfg.MODEL_VOS, cfg); model.load_state_dict(synth_state_dict(model.state_dict())); model = model.cuda().eval()
clips = [synth_clip(k, T, in_size=(481, 849), out_size=(480, 854), num_obj=10, device='cuda') for k in range(S)]
engs = [build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=model, gpu_id=0, long_term_mem_gap=cfg.TEST_LONG_TERM_MEM_GAP) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
def frame(e, img, out_size):
    e.match_propogate_one_frame(img); lg = e.decode_current_logits(out_size)
    lab = torch.argmax(torch.softmax(lg, 1), 1, keepdim=True).float()
    e.update_memory(F.interpolate(lab, size=e.input_size_2d, mode='nearest'))
    return lab
for rep in range(2):
    with torch.no_grad():
        for e, st, (fr, mk, ob, osz) in zip(engs, streams, clips):
            with torch.cuda.stream(st):
                e.restart_engine(); e.add_reference_frame(fr[0], mk, ob, frame_step=0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(1, T):
            for e, st, (fr, mk, ob, osz) in zip(engs, streams, clips):
                with torch.cuda.stream(st):
                    last = frame(e, fr[t], osz)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('streams', S, 'rep', rep, 'total %.1f fps (%.2f ms per frame-slot)' % (S * (T - 1) / dt, dt / (T - 1) * 1e3))

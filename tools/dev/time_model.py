import sys, os, time, importlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch, torch.nn.functional as F
from networks.models import build_vos_model
from networks.engines import build_engine
from utils.synth import synth_state_dict, synth_clip
name = sys.argv[1]; T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = importlib.import_module('configs.models.' + name).ModelConfig()
model = build_vos_model(cfg.MODEL_VOS, cfg); model.load_state_dict(synth_state_dict(model.state_dict())); model = model.cuda().eval()
insz = (480, 848) if 'swin' in name else (481, 849)
frames, mask, objs, out_size = synth_clip(0, T, in_size=insz, out_size=(480, 854), num_obj=10, device='cuda')
eng = build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=model, gpu_id=0, long_term_mem_gap=cfg.TEST_LONG_TERM_MEM_GAP)
for rep in range(2):
    eng.restart_engine()
    with torch.no_grad():
        eng.add_reference_frame(frames[0], mask, objs, frame_step=0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(1, T):
            eng.match_propogate_one_frame(frames[t]); lg = eng.decode_current_logits(out_size)
            lab = torch.argmax(torch.softmax(lg, 1), 1, keepdim=True).float()
            eng.update_memory(F.interpolate(lab, size=eng.input_size_2d, mode='nearest'))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, 'rep', rep, '%.2f ms/frame, %.1f fps over %d frames' % (dt / (T - 1) * 1e3, (T - 1) / dt, T - 1))

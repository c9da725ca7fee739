import sys, os, time, importlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'aot-benchmark_amd'))
import torch
import torch.nn.functional as F
from networks.models import build_vos_model
from networks.engines import build_engine
from networks.models.aot import to_tokens
from utils.synth import synth_state_dict, synth_clip
from oracle.aot_oracle import OracleModel, OracleEngine

name = sys.argv[1] if len(sys.argv) > 1 else 'aott'
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
size = tuple(int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else '257x257').split('x'))
nobj = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cfg = importlib.import_module('configs.models.' + name).ModelConfig()
model = build_vos_model(cfg.MODEL_VOS, cfg)
sd = synth_state_dict(model.state_dict())
model.load_state_dict(sd)
model = model.cuda().eval()
out_size = (size[0] - 1, size[1] + 5) if name.startswith('r50') else (size[0] - 1, size[1] - 1)
frames, mask, objs, out_size = synth_clip(0, T, in_size=size, out_size=out_size, num_obj=nobj)
om = OracleModel(name, sd); oe = OracleEngine(om)
eng = build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=model, gpu_id=0, long_term_mem_gap=cfg.TEST_LONG_TERM_MEM_GAP)

def d(a, b, tag):
    a = a.detach().float().cpu(); b = b.detach().float().cpu()
    print('  %-14s max|d| %.3e  (ref absmax %.3e) shape %s' % (tag, (a - b).abs().max().item(), b.abs().max().item(), tuple(a.shape)), flush=True)

with torch.no_grad():
    # stage-level checks on frame 0
    embs_o = om.encode_image(frames[0])
    feats = model.encode_tokens(frames[0].cuda())
    torch.cuda.synchronize()
    for i, ((t, h, w), e) in enumerate(zip(feats, embs_o)):
        d(t.view(h, w, -1).permute(2, 0, 1), e[0], 'enc%d' % i)
    # engines
    oe.add_reference_frame(frames[0], mask, objs)
    eng.add_reference_frame(frames[0].cuda(), mask.cuda(), objs, frame_step=0)
    e0 = eng.aot_engines[0]
    d(e0.pos_emb, oe.pos_emb[:, 0], 'pos_emb')
    d(e0.curr_id_embs, oe.curr_id_emb[:, 0], 'id_emb')
    L = cfg.MODEL_LSTT_NUM; C = 256
    for i in range(L):
        d(e0._cat[:, (i + 1) * C:(i + 2) * C], oe.curr_lstt_output[0][i][:, 0], 'ref lstt%d' % i)
    for t in range(1, T):
        print('frame', t)
        om.trace = {}
        oe.match_propogate_one_frame(frames[t])
        eng.match_propogate_one_frame(frames[t].cuda())
        for i in range(L):
            d(e0._curr[i][0], om.trace['L%d.curr_Q' % i][:, 0], 'curr_Q%d' % i)
            d(e0._cat[:, (i + 1) * C:(i + 2) * C], oe.curr_lstt_output[0][i][:, 0], 'lstt%d' % i)
        lo = oe.decode_current_logits(out_size)
        lg = eng.decode_current_logits(out_size)
        d(e0.pred_id_logits[:, :nobj + 1], oe.pred_id_logits[:, :nobj + 1], 'logits4')
        d(lg[:, :nobj + 1], lo[:, :nobj + 1], 'logits')
        mo = torch.argmax(torch.softmax(lo, 1), 1, keepdim=True).float()
        mg = torch.argmax(torch.softmax(lg, 1), 1, keepdim=True).float()
        print('  mask mismatches', (mo != mg.cpu()).sum().item(), 'of', mo.numel(), flush=True)
        oe.update_memory(F.interpolate(mo, size=oe.input_size_2d, mode='nearest'))
        eng.update_memory(F.interpolate(mg, size=eng.input_size_2d, mode='nearest'))
        d(e0._curr[0][1], oe.curr_lstt_output[1][0][1][:, 0], 'fusedV0')
print('bank_len', e0.bank_len)

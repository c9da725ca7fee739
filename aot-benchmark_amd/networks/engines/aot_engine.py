"""Per-clip inference engines (reference networks/engines/aot_engine.py).

``AOTEngine`` is the state machine of one <=10-object group: it owns the long-term memory bank, the
short-term memory (previous frame) and the current frame's intermediates, and drives the model's fused
HIP stages.  ``AOTInferEngine`` is the caller-facing wrapper (tools/demo.py:187-235,
networks/managers/evaluator.py:265-446 call exactly this surface).

MI355X-first differences from the reference implementation (same results):
  * the bank is a pre-allocated, geometrically grown [cap, C] buffer per layer that frames are
    APPENDED to (the reference re-copies the whole bank with torch.cat every `gap` frames,
    aot_engine.py:291-305); softmax attention is order-invariant;
  * one_hot_mask + the 17x17 identity conv are one gather kernel on the label map;
  * no NCHW<->sequence copies: token-major buffers are viewed as [1,C,h,w] / [N,1,C] at the API.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import aot_hip
from networks.models.aot import as_map, to_tokens


class AOTEngine(nn.Module):
    def __init__(self, aot_model, gpu_id=0, long_term_mem_gap=9999, short_term_mem_skip=1, long_term_mem_max=None):
        super().__init__()
        # long_term_mem_max (repo extension; the reference bank grows without bound): at most that many
        # memorised frames -- the first one (the reference frame) is kept, the others form a ring of the most recent
        if long_term_mem_max is not None and long_term_mem_max < 2:
            raise ValueError('long_term_mem_max must be >= 2 (the reference frame + at least one recent frame)')
        self.long_term_mem_max = long_term_mem_max
        self.cfg = aot_model.cfg
        self.align_corners = aot_model.cfg.MODEL_ALIGN_CORNERS
        self.AOT = aot_model
        self.max_obj_num = aot_model.max_obj_num
        self.gpu_id = gpu_id
        self.long_term_mem_gap = long_term_mem_gap
        self.short_term_mem_skip = short_term_mem_skip
        self.restart_engine()

    def forward(self, *a, **k):
        raise NotImplementedError('training forward (aot_engine.py:33-108) is outside the scoped inference path')

    # ---- state ---------------------------------------------------------------------------------
    def restart_engine(self, batch_size=1, enable_id_shuffle=False):
        if batch_size != 1 or enable_id_shuffle:
            raise NotImplementedError('inference runs batch 1 without id shuffle (aot_engine.py:445-477)')
        self.batch_size = 1
        self.frame_step = 0
        self.last_mem_step = -1
        self.obj_nums = None
        self.pos_emb = None
        self.enc_size_2d = None
        self.enc_hw = None
        self.input_size_2d = None
        # the bank buffers survive a restart (same clip geometry re-uses them: no allocator traffic, which would
        # serialise concurrently running clips); only the fill level is reset
        if not hasattr(self, 'bank_k'):
            self.bank_k, self.bank_v = None, None
        self.bank_len = 0
        self.bank_frames = 0      # frames ever memorised (ring position of a bounded bank)
        self.short_term_memories_list = []
        self.short_term_memories = None
        self._feats = None        # [(f4,h,w), (f8,h,w), (f16,h,w), (proj16,h,w)] token-major
        self._dec_in = None       # decoder input: AOT [N, (L+1)*C] (projected feature | LSTT outs), DeAOT [N, 2C]
        self._curr = None         # per layer current-frame memories (token-major), as returned by LSTT.run
        self.curr_id_embs = None
        self.pred_id_logits = None

    def update_size(self, input_size, enc_size):
        self.input_size_2d = tuple(int(x) for x in input_size)
        self.enc_size_2d = tuple(int(x) for x in enc_size)
        self.enc_hw = self.enc_size_2d[0] * self.enc_size_2d[1]

    @property
    def curr_enc_embs(self):
        """[4x, 8x, 16x, 16x-projected] maps as [1,C,h,w] views (read by multi-group sharing and by callers)."""
        if self._feats is None:
            return None
        return [as_map(t, h, w) for (t, h, w) in self._feats]

    def lstt_last(self):
        """Last LSTT / GPM layer output after its decoder norm, token-major [N, C] (AOT) / [N, 2C] (DeAOT): what the
        reference keeps as curr_lstt_output[0][-1] (aot_engine.py:340-354)."""
        C = self.AOT.encoder_projector.out_channels
        if self._dec_in.shape[1] == 2 * C and not self.AOT.decoder.decode_intermediate_input:
            return self._dec_in
        return self._dec_in[:, -C:]

    @property
    def long_term_memories(self):
        if self.bank_k is None or self.bank_len == 0:
            return None
        return [[k[:self.bank_len].unsqueeze(1), v[:self.bank_len].unsqueeze(1)] for k, v in zip(self.bank_k, self.bank_v)]

    # ---- helpers -------------------------------------------------------------------------------
    def _encode(self, img, img_embs):
        if img_embs is None:
            feats = self.AOT.encode_tokens(img)
        else:   # shared image embedding from another object group (aot_engine.py:606-607,612-616)
            feats = [(to_tokens(e), e.shape[2], e.shape[3]) for e in img_embs]
        self._feats = feats
        return feats

    def assign_identity(self, mask):
        """mask [1,1,H,W] label ids -> id embedding [N, C] (aot_engine.py:168-179 + utils/image.py:69-74)."""
        if mask.dim() == 4 and mask.shape[1] != 1:
            raise NotImplementedError('probability-map identities (MODEL_USE_PREV_PROB) need a dense id conv; not built')
        return self.AOT.id_emb_from_mask(mask, self.enc_size_2d)

    def _append_bank(self, ks, vs):
        N = ks[0].shape[0]
        fits = (self.bank_k is not None and len(self.bank_k) == len(ks) and
                all(b.shape[1] == k.shape[1] and b.device == k.device for b, k in zip(self.bank_k, ks)) and
                all(b.shape[1] == v.shape[1] for b, v in zip(self.bank_v, vs)))
        if not fits:
            cap = 16 * N          # 16 memorised frames up front (a 70-frame clip at gap 5 needs 14); doubles beyond
            self.bank_k = [torch.empty(cap, k.shape[1], dtype=torch.float32, device=k.device) for k in ks]
            self.bank_v = [torch.empty(cap, v.shape[1], dtype=torch.float32, device=v.device) for v in vs]
            self.bank_len = 0
            self.bank_frames = 0
        if self.long_term_mem_max is not None and self.bank_frames >= self.long_term_mem_max:
            # bounded bank: overwrite the oldest non-first frame (attention is order-invariant, so a ring is enough)
            slot = 1 + (self.bank_frames - 1) % (self.long_term_mem_max - 1)
            for i, (k, v) in enumerate(zip(ks, vs)):
                self.bank_k[i][slot * N:(slot + 1) * N].copy_(k)
                self.bank_v[i][slot * N:(slot + 1) * N].copy_(v)
            self.bank_frames += 1
            return
        if self.bank_len + N > self.bank_k[0].shape[0]:
            cap = max(2 * self.bank_k[0].shape[0], self.bank_len + N)
            for lst in (self.bank_k, self.bank_v):
                for i, old in enumerate(lst):
                    new = torch.empty(cap, old.shape[1], dtype=torch.float32, device=old.device)
                    new[:self.bank_len].copy_(old[:self.bank_len])
                    lst[i] = new
        for i, (k, v) in enumerate(zip(ks, vs)):
            self.bank_k[i][self.bank_len:self.bank_len + N].copy_(k)
            self.bank_v[i][self.bank_len:self.bank_len + N].copy_(v)
        self.bank_len += N
        self.bank_frames += 1

    # ---- reference surface ---------------------------------------------------------------------
    def add_reference_frame(self, img=None, mask=None, frame_step=-1, obj_nums=None, img_embs=None):
        if self.obj_nums is None and obj_nums is None:
            print('No objects for reference frame!')
            exit()
        elif obj_nums is not None:
            self.obj_nums = obj_nums
        if frame_step == -1:
            frame_step = self.frame_step
        if img is None and img_embs is None:
            print('No image for reference frame!')
            exit()
        if mask is None:
            print('No mask for reference frame!')
            exit()
        feats = self._encode(img, img_embs)
        if self.input_size_2d is None:
            self.update_size(img.size()[2:] if img is not None else mask.size()[2:], (feats[3][1], feats[3][2]))
        if self.pos_emb is None:
            self.pos_emb = to_tokens(self.AOT.get_pos_emb(as_map(feats[3][0], feats[3][1], feats[3][2])).contiguous(
                memory_format=torch.channels_last)).contiguous()
        id_emb = self.assign_identity(mask)
        self.curr_id_embs = id_emb
        stream = aot_hip.stream_ptr()
        self._dec_in, outs, mems = self.AOT.LSTT.run(feats[3][0], None, None, id_emb, self.pos_emb, self.enc_size_2d,
                                                     self.AOT.ws, stream)
        self._curr = list(mems)
        self._append_bank([m[2][0] for m in mems], [m[2][1] for m in mems])
        self.last_mem_step = self.frame_step
        st = [(m[3][0], m[3][1]) for m in mems]
        self.short_term_memories_list = [st]
        self.short_term_memories = st

    def match_propogate_one_frame(self, img=None, img_embs=None):
        self.frame_step += 1
        feats = self._encode(img, img_embs)
        stream = aot_hip.stream_ptr()
        lm = list(zip(self.bank_k, self.bank_v))
        self._dec_in, outs, mems = self.AOT.LSTT.run(feats[3][0], lm, self.short_term_memories, None, self.pos_emb,
                                                     self.enc_size_2d, self.AOT.ws, stream, t_long=self.bank_len)
        self._curr = list(mems)

    def decode_current_logits(self, output_size=None):
        stream = aot_hip.stream_ptr()
        f4, f8, f16, _ = self._feats
        dec = self.AOT.decoder
        logits, h4, w4 = dec.run(self._dec_in, f16, f8, f4, self.AOT.ws, stream)
        nc = logits.shape[1]
        dev = logits.device
        obj_num = int(self.obj_nums[0])
        out4 = torch.empty(1, nc, h4, w4, dtype=torch.float32, device=dev)
        out = None
        if output_size is not None:
            oh, ow = int(output_size[0]), int(output_size[1])
            out = torch.empty(1, nc, oh, ow, dtype=torch.float32, device=dev)
            aot_hip.logits_finalize(logits, out4, out, h4, w4, nc, oh, ow, obj_num, self.align_corners, stream=stream)
        else:
            aot_hip.logits_finalize(logits, out4, None, h4, w4, nc, 0, 0, obj_num, self.align_corners, stream=stream)
        self.pred_id_logits = out4
        return out if out is not None else out4

    def update_long_term_memory(self, new_long_term_memories):
        """Reference signature (aot_engine.py:291-305): list over layers of [K, V] ([N,1,C]); appended."""
        self._append_bank([to_tokens(m[0]) for m in new_long_term_memories],
                          [to_tokens(m[1]) for m in new_long_term_memories])

    def update_short_term_memory(self, curr_mask, curr_id_emb=None, skip_long_term_update=False):
        if curr_id_emb is None:
            curr_id_emb = self.assign_identity(curr_mask)
        else:
            curr_id_emb = to_tokens(curr_id_emb)
        self.curr_id_embs = curr_id_emb
        stream = aot_hip.stream_ptr()
        fused = [self.AOT.LSTT.layers[i].update_memory_kv(m, curr_id_emb, self.AOT.ws, stream)
                 for i, m in enumerate(self._curr)]
        self.short_term_memories_list.append(fused)
        self.short_term_memories_list = self.short_term_memories_list[-self.short_term_mem_skip:]
        self.short_term_memories = self.short_term_memories_list[0]
        if self.frame_step - self.last_mem_step >= self.long_term_mem_gap:
            if not skip_long_term_update:
                self._append_bank([f[0] for f in fused], [f[1] for f in fused])
            self.last_mem_step = self.frame_step

    def predict_current_mask(self, output_size=None, return_prob=False):
        if output_size is None:
            output_size = self.input_size_2d
        logits = F.interpolate(self.pred_id_logits, size=output_size, mode='bilinear', align_corners=self.align_corners)
        pred_mask = torch.argmax(logits, dim=1)
        if not return_prob:
            return pred_mask
        return pred_mask, torch.softmax(logits, dim=1)


class DeAOTEngine(AOTEngine):
    """reference networks/engines/deaot_engine.py:9-56.  The memory layout differences of DeAOT ([K 128 | V 512 | ID_V 512]
    per token, only ID_V refreshed at update time) live in GatedPropagationModule.update_memory_kv / run; the state
    machine is the same."""


class AOTInferEngine(nn.Module):
    """Caller-facing engine (reference aot_engine.py:485-635): one AOTEngine per group of max_aot_obj_num
    objects, created lazily; the image embedding is computed once per frame and shared."""

    engine_cls = AOTEngine

    def __init__(self, aot_model, gpu_id=0, long_term_mem_gap=9999, short_term_mem_skip=1, max_aot_obj_num=None,
                 long_term_mem_max=None):
        super().__init__()
        self.long_term_mem_max = long_term_mem_max       # bounded bank per object group (repo extension)
        self.cfg = aot_model.cfg
        self.AOT = aot_model
        if max_aot_obj_num is None or max_aot_obj_num > aot_model.max_obj_num:
            self.max_aot_obj_num = aot_model.max_obj_num
        else:
            self.max_aot_obj_num = max_aot_obj_num
        self.gpu_id = gpu_id
        self.long_term_mem_gap = long_term_mem_gap
        self.short_term_mem_skip = short_term_mem_skip
        self.aot_engines = []
        self.restart_engine()

    def restart_engine(self):
        del (self.aot_engines)
        self.aot_engines = []
        self.obj_nums = None

    def separate_mask(self, mask, obj_nums):
        if mask is None:
            return [None] * len(self.aot_engines)
        if len(self.aot_engines) == 1:
            return [mask], [obj_nums]
        separated_obj_nums = [self.max_aot_obj_num for _ in range(len(self.aot_engines))]
        if obj_nums % self.max_aot_obj_num > 0:
            separated_obj_nums[-1] = obj_nums % self.max_aot_obj_num
        if len(mask.size()) == 3 or mask.size()[0] == 1:
            separated_masks = []
            for idx in range(len(self.aot_engines)):
                start_id = idx * self.max_aot_obj_num + 1
                end_id = (idx + 1) * self.max_aot_obj_num
                fg_mask = ((mask >= start_id) & (mask <= end_id)).float()
                separated_masks.append((fg_mask * mask - start_id + 1) * fg_mask)
            return separated_masks, separated_obj_nums
        raise NotImplementedError('probability-map masks (aot_engine.py:536-545) are not on the scoped path')

    def soft_logit_aggregation(self, all_logits):
        """aot_engine.py:565-582.  Identity for <=10 objects (the scoped configs); the multi-group merge is
        host-side torch plumbing for now."""
        if len(all_logits) == 1:
            return all_logits[0]
        fg_probs, bg_probs = [], []
        for logit in all_logits:
            prob = torch.softmax(logit, dim=1)
            bg_probs.append(prob[:, 0:1])
            fg_probs.append(prob[:, 1:1 + self.max_aot_obj_num])
        bg_prob = torch.prod(torch.cat(bg_probs, dim=1), dim=1, keepdim=True)
        merged_prob = torch.cat([bg_prob] + fg_probs, dim=1).clamp(1e-5, 1 - 1e-5)
        return torch.logit(merged_prob)

    def add_reference_frame(self, img, mask, obj_nums, frame_step=-1):
        if isinstance(obj_nums, list):
            obj_nums = obj_nums[0]
        self.obj_nums = obj_nums
        aot_num = max(np.ceil(obj_nums / self.max_aot_obj_num), 1)
        while aot_num > len(self.aot_engines):
            new_engine = self.engine_cls(self.AOT, self.gpu_id, self.long_term_mem_gap, self.short_term_mem_skip,
                                         self.long_term_mem_max)
            new_engine.eval()
            self.aot_engines.append(new_engine)
        separated_masks, separated_obj_nums = self.separate_mask(mask, obj_nums)
        img_embs = None
        for aot_engine, separated_mask, separated_obj_num in zip(self.aot_engines, separated_masks,
                                                                 separated_obj_nums):
            aot_engine.add_reference_frame(img, separated_mask, obj_nums=[separated_obj_num], frame_step=frame_step,
                                           img_embs=img_embs)
            if img_embs is None:
                img_embs = aot_engine.curr_enc_embs
        self.update_size()

    def match_propogate_one_frame(self, img=None):
        img_embs = None
        for aot_engine in self.aot_engines:
            aot_engine.match_propogate_one_frame(img, img_embs=img_embs)
            if img_embs is None:
                img_embs = aot_engine.curr_enc_embs

    def decode_current_logits(self, output_size=None):
        all_logits = [e.decode_current_logits(output_size) for e in self.aot_engines]
        return self.soft_logit_aggregation(all_logits)

    def update_memory(self, curr_mask, skip_long_term_update=False):
        separated_masks, _ = self.separate_mask(curr_mask, self.obj_nums)
        for aot_engine, separated_mask in zip(self.aot_engines, separated_masks):
            aot_engine.update_short_term_memory(separated_mask, skip_long_term_update=skip_long_term_update)

    def update_size(self):
        self.input_size_2d = self.aot_engines[0].input_size_2d
        self.enc_size_2d = self.aot_engines[0].enc_size_2d
        self.enc_hw = self.aot_engines[0].enc_hw


class DeAOTInferEngine(AOTInferEngine):
    """reference networks/engines/deaot_engine.py:59-94."""
    engine_cls = DeAOTEngine

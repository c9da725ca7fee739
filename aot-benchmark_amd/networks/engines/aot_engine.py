"""Per-clip inference engines (reference networks/engines/aot_engine.py, deaot_engine.py).

MI355X-first design -- same results as the reference, different machinery:

  * LANES.  The reference gives every group of <= 10 objects its own ``AOTEngine`` and runs the engines one after the
    other on shared image features (aot_engine.py:584-616).  Here the groups are lanes of ONE batched pass: every LSTT /
    decoder kernel takes the B lanes stacked along the token rows ([B*N, C]), mask separation happens inside the identity
    gather kernel, and the per-group logits are masked, resized and soft-aggregated by one kernel
    (``aot_logits_finalize_f32``).  A frame costs the same number of launches for 44 objects as for 4.
  * ``AOTEngine`` is a COHORT of lanes that share a schedule (frame counter, memorisation steps, bank length).  One
    cohort is the normal case; objects that first appear mid-clip and open a new group start a cohort of their own,
    because the reference starts a fresh engine (own frame counter, empty bank) for them.
  * The long-term bank is a pre-allocated [lanes, capacity, C] buffer per layer that memorised frames are APPENDED to
    (softmax attention is order-invariant; the reference re-copies the whole bank with torch.cat, aot_engine.py:291-305).
    With one lane the frame's K and id-fused V are written by their GEMMs straight into the next bank slot on the frames
    that will be memorised -- no copy at all -- and the short-term memory of the following frame is a view of that slot.
  * one_hot_mask + the 17x17 identity conv + the `V + id_emb` sums of all layers are one gather kernel on the label map.

``AOTInferEngine`` is the caller-facing surface (tools/demo.py:187-235, networks/managers/evaluator.py:265-446).
"""
import functools

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

import aot_hip
from networks.engines.graphs import FrameGraphs, ptr_key
from networks.layers.workspace import Workspace
from networks.models.aot import as_map, to_tokens


def _in_table(fn):
    """Runs an engine stage inside the engine's own GEMM dispatch table (aot_hip.use_gemm_table): the table is an engine
    attribute (build_engine(..., gemm_table=)), not process state."""
    @functools.wraps(fn)
    def stage(self, *args, **kwargs):
        with aot_hip.use_gemm_table(self.gemm_table, self.mfma):
            return fn(self, *args, **kwargs)
    return stage


def _img_key(img):
    """Identity of a caller-owned frame: address, shape and version counter (a buffer re-filled in place is a new frame)."""
    return (img.data_ptr(), tuple(img.shape), img._version)


def _die(msg):
    """The reference reports a missing input by printing and leaving the process (aot_engine.py:194-217)."""
    print(msg)
    exit()


class AOTEngine(nn.Module):
    """A cohort of `lanes` object groups (groups group0 .. group0+lanes-1 of the clip's label map) advancing in lock
    step.  With lanes = 1 and group0 = None this is the reference's AOTEngine for <= 10 objects: masks carry the labels
    0..max_obj_num as they are."""

    def __init__(self, aot_model, gpu_id=0, long_term_mem_gap=9999, short_term_mem_skip=1, long_term_mem_max=None, lanes=1,
                 group0=None, graph=False, gemm_table='latency', mfma='f32'):
        super().__init__()
        # matrix-core arithmetic of the conv / linear layers: 'f32' (exact fp32 products, default) or 'bf16x6' (the
        # fp32-equivalent six-term bf16 split, aot_conv2d_bf16x6_f32: a second, parity-gated kernel family)
        if mfma not in aot_hip.MFMA_MODES:
            raise ValueError('mfma must be one of %s' % (aot_hip.MFMA_MODES,))
        self.mfma = mfma
        # which conv / linear dispatch table this engine's stages run under (include/aot_hip.h, cfg -1 / -2): 'latency' for
        # one clip at a time, 'throughput' when several engines keep the GPU busy on their own streams
        if gemm_table not in aot_hip.GEMM_TABLES:
            raise ValueError('gemm_table must be one of %s' % sorted(aot_hip.GEMM_TABLES))
        self.gemm_table = gemm_table
        # graph=True: the launch sequence of every stage (match / decode / memory update) is captured once per engine
        # state as a hipGraph and replayed (engines/graphs.py).  What decode_current_logits returns is then a buffer that
        # belongs to the graph: valid until the next decode of this engine (the reference returns a fresh tensor).
        self.use_graph = bool(graph)
        self._graphs = None
        self._static = {}            # staged copies of caller-owned inputs (image, label map): stable addresses for replay
        self._arena = Workspace()    # tensors that live from one stage of a frame to the next (curr_V, decoder input)
        self._ahead = {}             # _img_key(image tensor) -> its token-major features: frames encoded by encode_ahead()
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
        self.short_term_mem_skip = max(1, int(short_term_mem_skip))
        self.lanes = int(lanes)
        # Graph mode, "state-free" form: the captured launches must not depend on how many frames the bank holds, or a clip
        # would need one graph per frame.  The bank length reaches the attention kernels through a device int (T_dev of
        # aot_attn_f32 / aot_gated_attn_f32), a memorised frame is appended by a copy whose slot is a device int
        # (aot_copy_rows_f32), and the launch geometry is planned for the bank's capacity -- a graph is keyed on (stage,
        # frame geometry, bank buffer, which scratch set holds the previous frame), a handful per clip however long it is.
        # The long-video knobs that change the ARITHMETIC with the bank length (top_k, max_mem_len_ratio) keep the
        # one-graph-per-state form.
        self._state_free = self.use_graph and not any(l.long_term_attn.top_k > 0 or l.long_term_attn.max_mem_len_ratio > 0
                                                      for l in aot_model.LSTT.layers)
        self._dev_ints = None        # [T, slot] int32 on the device (state-free graph mode)
        self._dev_vals = [None, None]
        self.group0 = group0         # first object group of the clip's label map held by lane 0 (None: labels as they are)
        self.first_group = group0 or 0
        self._bank = None            # per layer (K [lanes, cap*N, Ck], V [lanes, cap*N, Cv]); survives restart_engine
        self._bank_geom = None
        # mfma = 'bf16x6', AOT's 32-wide heads: the bank also exists pre-split into bf16 planes (aot_attn_pack_x6_f32), which is
        # what the long-term attention of that kernel family reads (aot_attn_x6_f32); per layer (planes, rows per lane)
        self._bank_x6 = None
        self._x6_gated = mfma == 'bf16x6' and type(aot_model.LSTT).__name__ == 'DualBranchGPM'
        if not self._x6_gated:
            self._x6_attn = (mfma == 'bf16x6' and all(l.long_term_attn.hidden_dim == 32 and l.long_term_attn.top_k <= 0
                                                      for l in aot_model.LSTT.layers))
        else:           # DeAOT: one 128-wide head, value [V | ID_V] 1024 wide (aot_gated_attn_x6_f32)
            self._x6_attn = all(l.long_term_attn.num_head == 1 and l.d_att == 128 and 2 * l.expand_d_model == 1024
                                and l.long_term_attn.top_k <= 0 for l in aot_model.LSTT.layers)
        self._ring = None            # rotating scratch sets for frames whose K/V do not live in a bank slot; survives too
        self.losses = None           # built by the first forward() (training only)
        self.restart_engine()

    # ---- training-step forward (aot_engine.py:33-108) -------------------------------------------
    def _init_losses(self):
        """aot_engine.py:110-125: 0.5 * hard-mined cross entropy + 0.5 * soft Jaccard; the auxiliary loss of the frames that
        see their own mask fades out linearly over TRAIN_TOTAL_STEPS * TRAIN_AUX_LOSS_RATIO steps."""
        from networks.layers.loss import CrossEntropyLoss, SoftJaccordLoss
        cfg = self.cfg
        self.losses = nn.ModuleList([CrossEntropyLoss(cfg.TRAIN_TOP_K_PERCENT_PIXELS,
                                                      cfg.TRAIN_HARD_MINING_RATIO * cfg.TRAIN_TOTAL_STEPS),
                                     SoftJaccordLoss()])
        self.loss_weights = [0.5, 0.5]
        self.aux_weight = cfg.TRAIN_AUX_LOSS_WEIGHT
        self.aux_step = cfg.TRAIN_TOTAL_STEPS * cfg.TRAIN_AUX_LOSS_RATIO + 1e-5

    @_in_table
    def forward(self, all_frames, all_masks, batch_size, obj_nums, step=0, tf_board=False, use_prev_pred=False,
                enable_prev_frame=False, use_prev_prob=False):
        """The reference's training-step forward: all_frames [T*bs, 3, H, W] / all_masks [T*bs, 1, H, W], time-major
        (reference frame, previous frame, current frames; trainer.py:452-455) -> (loss, per-frame masks [bs, H, W], per-frame
        losses [bs], boards).  Frame 0 memorises its own mask, frames 1.. are propagated and feed their ground-truth mask
        (or, use_prev_pred, their own prediction / probabilities) back; every frame is decoded and scored.

        The samples of a batch are independent clips (every reference op on this path is per-sample), so they run here one
        after the other.  Two forms:
          * autograd enabled (a training step): the differentiable forward of networks/models/train_forward.py -- every graph
            node a C-ABI kernel with a hand-written backward (csrc/train_bwd.hip), drop-path / dropout as the modules'
            `training` flag says; `loss.backward()` fills the parameters' .grad as the reference's does;
          * under torch.no_grad() (validation, the parity tests of the forward): the fused inference kernels, single lane,
            per-frame encoder; no graph, no regularisers."""
        if self.lanes != 1 or self.group0 is not None:
            raise NotImplementedError('the training forward runs <= %d objects per sample on one lane' % self.max_obj_num)
        if self.losses is None:
            self._init_losses()
        bs = int(batch_size)
        if bs != self.batch_size:
            raise ValueError('batch_size %d differs from restart_engine(%d, ...)' % (bs, self.batch_size))
        T = all_frames.shape[0] // bs
        if T < 3 or all_frames.shape[0] != T * bs or all_masks.shape[0] != T * bs:
            raise ValueError('need >= 3 frames per sample, time-major: got %d frames / %d masks for batch %d'
                             % (all_frames.shape[0], all_masks.shape[0], bs))
        if torch.is_grad_enabled():
            from networks.models.train_forward import training_forward
            return training_forward(self, all_frames, all_masks, bs, obj_nums, step=step, use_prev_pred=use_prev_pred,
                                    enable_prev_frame=enable_prev_frame, use_prev_prob=use_prev_prob)
        aux_weight = self.aux_weight * max(self.aux_step - step, 0.) / self.aux_step
        frames = all_frames.view(T, bs, *all_frames.shape[1:])
        masks = all_masks.view(T, bs, *all_masks.shape[1:]).float()
        n_aux = 2 if enable_prev_frame else 1
        losses = [[None] * bs for _ in range(T)]
        preds = [[None] * bs for _ in range(T)]
        for b in range(bs):
            self._restart_clip()
            self._sample = b
            objs = int(obj_nums[b])
            # with the identities shuffled the objects sit on arbitrary channels: nothing is masked by count until the
            # channels are back in place (aot_engine.py:364-372)
            count = [self.max_obj_num if self.enable_id_shuffle else objs]

            def score(t):
                loss, mask, prob = self._loss_and_mask(masks[t, b], objs, step, want_prob=use_prev_prob)
                losses[t][b], preds[t][b] = loss, mask
                return mask.view(1, 1, *mask.shape[-2:]).float() if not use_prev_prob else prob

            def feed_back(t, pred):
                self.update_short_term_memory(self._shuffled(pred if use_prev_pred else masks[t, b:b + 1]))

            self.add_reference_frame(frames[0, b:b + 1], self._shuffled(masks[0, b:b + 1]), frame_step=0, obj_nums=count)
            score(0)
            t = 1
            if enable_prev_frame:
                self.set_prev_frame(frames[1, b:b + 1], self._shuffled(masks[1, b:b + 1]), frame_step=1)
                score(1)
                t = 2
            while t < T:
                self.match_propogate_one_frame(frames[t, b:b + 1])
                pred = score(t)
                if t < T - 1:
                    feed_back(t, pred)
                t += 1
        self._sample = None
        frame_loss = [torch.cat(l, 0) for l in losses]
        frame_mask = [torch.cat(m, 0) for m in preds]
        aux_loss = torch.cat(frame_loss[:n_aux], 0).mean(0)
        pred_loss = torch.cat(frame_loss[n_aux:], 0).mean(0)
        loss = aux_weight * aux_loss + pred_loss
        return loss, frame_mask, frame_loss, {'image': {}, 'scalar': {}}

    def _loss_and_mask(self, gt, objs, step, want_prob=False):
        """generate_loss_mask of one sample (aot_engine.py:398-430): decode at the label size, put shuffled identities back,
        score the first objs+1 channels.  gt [1, H, W].  Returns (loss [1], mask [1, H, W] long, prob [1, L, H, W] | None)."""
        size = tuple(gt.shape[-2:])
        logits = self.decode_current_logits(size)
        if self.enable_id_shuffle:
            perm = self.id_shuffle[self._sample]
            logits = logits.index_select(1, perm)               # channel t <- the channel identity t was moved to
            logits[:, objs + 1:] = -1e10
            self.pred_id_logits = self.pred_id_logits.index_select(1, perm)
            self.pred_id_logits[:, objs + 1:] = -1e10
        scored = [logits[:, :objs + 1].contiguous()]
        label = [gt.view(1, *size)]
        loss = 0
        for fn, wgt in zip(self.losses, self.loss_weights):
            loss = loss + wgt * fn(scored, label, step)
        mask, _, prob = aot_hip.fuse_probs(logits, [False], want_aug_labels=False, want_prob=want_prob)
        return loss, mask.view(1, *size).long(), prob

    @_in_table
    def generate_loss_mask(self, gt_mask, step, return_prob=False):
        """aot_engine.py:421-430 for the clip this engine holds (batch 1): decode, score against gt_mask [1,1,H,W] or
        [1,H,W], predict.  Returns (loss [1], mask [1,H,W]) and, return_prob, the class probabilities [1,L,H,W]."""
        if self.losses is None:
            self._init_losses()
        if self.enable_id_shuffle and self._sample is None:
            self._sample = 0
        loss, mask, prob = self._loss_and_mask(gt_mask.reshape(1, *gt_mask.shape[-2:]).float(), self._group_objects(), step,
                                               want_prob=return_prob)
        return (loss, mask, prob) if return_prob else (loss, mask)

    def _shuffled(self, m):
        """Moves the identities of a label map [1,1,H,W] or a probability map [1,L,H,W] of the current sample to their
        shuffled channels (assign_identity's einsum, aot_engine.py:168-172); unchanged when the shuffle is off."""
        if not self.enable_id_shuffle:
            return m
        perm = self.id_shuffle[self._sample]
        if m.shape[1] != 1:
            out = torch.empty_like(m)
            out[:, perm] = m
            return out
        ids = m.long()
        known = ids <= self.max_obj_num                           # (the ignore label stays what it is)
        return torch.where(known, perm[ids.clamp(max=self.max_obj_num)], ids).float()

    @_in_table
    def set_prev_frame(self, img=None, mask=None, frame_step=1):
        """aot_engine.py:253-289: a second frame that memorises its own mask (appended to the bank, becomes the short-term
        memory) -- the reference-frame stage again at another frame counter."""
        if img is None:
            _die('No image for previous frame!')
        if mask is None:
            _die('No mask for previous frame!')
        self.frame_step = frame_step
        self.add_reference_frame(img, mask)

    # ---- state ---------------------------------------------------------------------------------
    def restart_engine(self, batch_size=1, enable_id_shuffle=False):
        """batch_size > 1 and enable_id_shuffle only matter to forward() (trainer.py:457); the per-frame calls serve one clip.
        The shuffle is one random permutation of the identities 1..max_obj_num per sample, background fixed
        (utils/math.py:3-24)."""
        self.batch_size = int(batch_size)
        self.enable_id_shuffle = bool(enable_id_shuffle)
        self.id_shuffle = None
        self._sample = None
        if self.enable_id_shuffle:
            dev = next(self.AOT.parameters()).device
            self.id_shuffle = [torch.cat([torch.zeros(1, dtype=torch.long, device=dev),
                                          1 + torch.randperm(self.max_obj_num, device=dev)]) for _ in range(self.batch_size)]
        self._restart_clip()

    def _restart_clip(self):
        self.frame_step = 0
        self.last_mem_step = -1
        self.obj_nums = None
        self.pos_emb = None
        self.enc_size_2d = None
        self.enc_hw = None
        self.input_size_2d = None
        self.bank_frames = 0         # frames ever memorised
        self._slots = 0              # bank slots in use (= bank_frames, or the bound of a bounded bank)
        self._short = []             # most recent short-term memories, oldest first; entry = per layer (K, V, rows)
        self._ring_pos = 0           # (the ring buffers themselves are kept: same addresses clip after clip)
        self._feats = None           # [(f4,h,w), (f8,h,w), (f16,h,w), (proj16,h,w)] token-major, shared by the lanes
        self._dec_in = None          # decoder input: AOT [B*N, (L+1)*C] (projected feature | LSTT outs), DeAOT [B*N, 2C]
        self._curr = None            # this frame's per-layer (K, V | Vcat, ...) as returned by LSTT.run
        self._curr_slot = None       # bank slot this frame's K/V were written to directly, if any
        self._dst = None             # buffers this frame's K / V live in
        self.curr_id_embs = None
        self.pred_id_logits = None
        self._ahead = {}
        self._dev_vals = [None, None]

    def update_size(self, input_size, enc_size):
        self.input_size_2d = tuple(int(x) for x in input_size)
        self.enc_size_2d = tuple(int(x) for x in enc_size)
        self.enc_hw = self.enc_size_2d[0] * self.enc_size_2d[1]

    @property
    def bank_len(self):
        """Tokens per lane in the long-term bank."""
        return self._slots * (self.enc_hw or 0)

    @property
    def bank_k(self):
        return None if self._bank is None else [k.view(-1, k.shape[2]) for k, _ in self._bank]

    @property
    def bank_v(self):
        return None if self._bank is None else [v.view(-1, v.shape[2]) for _, v in self._bank]

    @property
    def curr_enc_embs(self):
        """[4x, 8x, 16x, 16x-projected] maps as [1,C,h,w] views (read by callers and shared between cohorts)."""
        if self._feats is None:
            return None
        return [as_map(t, h, w) for (t, h, w) in self._feats]

    def lstt_last(self, lane=0):
        """Last LSTT / GPM layer output after its decoder norm, token-major [N, C] (AOT) / [N, 2C] (DeAOT): what the
        reference keeps as curr_lstt_output[0][-1] (aot_engine.py:340-354)."""
        N = self.enc_hw
        rows = self._dec_in[lane * N:(lane + 1) * N]
        if type(self.AOT.LSTT).__name__ == 'DualBranchGPM':      # DeAOT: the decoder input IS the normed [tgt | tgt_id]
            return rows
        return rows[:, -self.AOT.encoder_projector.out_channels:]   # AOT: last column block of [feature | layer outputs]

    @property
    def long_term_memories(self):
        """Reference-shaped view of lane 0's bank: per layer [K, V] as [T, 1, C]."""
        if self._bank is None or self._slots == 0:
            return None
        T = self.bank_len
        return [[k[0, :T].unsqueeze(1), v[0, :T].unsqueeze(1)] for k, v in self._bank]

    # ---- bank plumbing -------------------------------------------------------------------------
    def _direct(self):
        """K / V GEMMs may write straight into bank slots: one lane, unbounded bank, previous-frame short-term memory."""
        return self.lanes == 1 and self.long_term_mem_max is None and self.short_term_mem_skip == 1 and not self._state_free

    def _ensure_bank(self, frames_needed):
        N, B = self.enc_hw, self.lanes
        widths = self.AOT.mem_widths()
        dev = next(self.AOT.parameters()).device
        geom = (N, B, tuple(widths), dev)
        if self._bank is None or self._bank_geom != geom:
            # memorised frames up front (a 70-frame clip at gap 5 needs 14), x4 beyond: a re-allocation orphans the graphs
            # captured on the old buffers, and the attention launches are planned for the capacity (layers/attention.py)
            cap = 32
            self._bank = [(torch.empty(B, cap * N, ck, dtype=torch.float32, device=dev),
                           torch.empty(B, cap * N, cv, dtype=torch.float32, device=dev)) for ck, cv in widths]
            self._bank_geom = geom
            self._bank_x6 = [self._new_x6(B, cap * N, ck, cv, dev) for ck, cv in widths] if self._x6_attn else None
        cap = self._bank[0][0].shape[1] // N
        if frames_needed > cap:
            new_cap = max(4 * cap, frames_needed)
            used = self._slots * N
            grown = []
            for k, v in self._bank:
                k2 = torch.empty(B, new_cap * N, k.shape[2], dtype=torch.float32, device=dev)
                v2 = torch.empty(B, new_cap * N, v.shape[2], dtype=torch.float32, device=dev)
                k2[:, :used].copy_(k[:, :used])
                v2[:, :used].copy_(v[:, :used])
                grown.append((k2, v2))
            self._bank = grown
            if self._x6_attn:            # the packed copy: re-split what the bank holds
                self._bank_x6 = [self._new_x6(B, new_cap * N, k.shape[2], v.shape[2], dev) for k, v in grown]
                if used:
                    pack = aot_hip.gated_pack_x6 if self._x6_gated else aot_hip.attention_pack_x6
                    for (k, v), xb in zip(grown, self._bank_x6):
                        pack(k.view(-1, k.shape[2]), v.view(-1, v.shape[2]), xb, used, B=B, src_brows=k.shape[1])

    def _next_slot(self):
        """Bank slot the next memorised frame goes to (append; a bounded bank overwrites its oldest non-first frame)."""
        if self.long_term_mem_max is not None and self.bank_frames >= self.long_term_mem_max:
            return 1 + (self.bank_frames - 1) % (self.long_term_mem_max - 1)
        return self._slots

    def _slot_views(self, slot):
        """Per layer (K rows, V rows) [N, C] of bank slot `slot` of lane 0 (the direct-write path is single-lane)."""
        N = self.enc_hw
        return [(k[0, slot * N:(slot + 1) * N], v[0, slot * N:(slot + 1) * N]) for k, v in self._bank]

    def _brows_bank(self):
        return self._bank[0][0].shape[1]

    def _commit(self, slot):
        if slot >= self._slots:
            self._slots = slot + 1
        self.bank_frames += 1

    def _scratch_set(self):
        """A [B*N, C] K / V buffer set for this frame out of a small ring (short_term_mem_skip + 2 sets, so the sets that
        still back a short-term memory are never the one being written)."""
        N, B = self.enc_hw, self.lanes
        dev = next(self.AOT.parameters()).device
        n = self.short_term_mem_skip + 2
        if self._ring is None or self._ring[0][0][0].shape[0] != B * N or self._ring[0][0][0].device != dev:
            self._ring = [[(torch.empty(B * N, ck, dtype=torch.float32, device=dev),
                            torch.empty(B * N, cv, dtype=torch.float32, device=dev)) for ck, cv in self.AOT.mem_widths()]
                          for _ in range(n)]
            self._ring_pos = 0
        s = self._ring[self._ring_pos % len(self._ring)]
        self._ring_pos += 1
        return s

    def _store(self, kv, slot, slot_dev=None):
        """Copies this frame's per-layer (K, V) [B*N, C] into bank slot `slot` of every lane (one launch each; slot_dev: the
        slot as a device int, for a replayed graph)."""
        N, B = self.enc_hw, self.lanes
        stream = aot_hip.stream_ptr()
        for (bk, bv), (k, v) in zip(self._bank, kv):
            aot_hip.copy_rows(k, bk.view(-1, bk.shape[2]), N, B=B, dst_brows=bk.shape[1], slot=slot, slot_dev=slot_dev,
                              stream=stream)
            aot_hip.copy_rows(v, bv.view(-1, bv.shape[2]), N, B=B, dst_brows=bv.shape[1], slot=slot, slot_dev=slot_dev,
                              stream=stream)
        self._pack_x6(kv, slot, slot_dev, N)

    def _pack_x6(self, kv, slot, slot_dev, src_brows):
        """The bf16x6 family's packed copy of bank slot `slot`: per layer (K, V) rows [b * src_brows, + N) of every lane, split
        into planes.  kv may be the frame's own buffers or the bank slot's rows themselves (direct-write path)."""
        if not self._x6_attn:
            return
        stream = aot_hip.stream_ptr()
        pack = aot_hip.gated_pack_x6 if self._x6_gated else aot_hip.attention_pack_x6
        for xb, (k, v) in zip(self._bank_x6, kv):
            pack(k, v, xb, self.enc_hw, B=self.lanes, src_brows=src_brows, slot=slot, slot_dev=slot_dev, stream=stream)

    def _new_x6(self, B, rows, ck, cv, dev):
        return aot_hip.x6_gated_bank(B, rows, ck, cv, dev) if self._x6_gated else aot_hip.x6_bank(B, rows, ck, dev)

    def _dev_int(self, i, value):
        """Device int i (0: bank length in tokens, 1: bank slot of the frame being memorised) set to `value` on the current
        stream -- outside any capture, in front of the replay that reads it; written only when the value changes."""
        if self._dev_ints is None:
            self._dev_ints = torch.zeros(2, dtype=torch.int32, device=next(self.AOT.parameters()).device)
        if self._dev_vals[i] != value:
            self._dev_ints[i:i + 1].fill_(int(value))
            self._dev_vals[i] = value
        return self._dev_ints[i:i + 1]

    def _push_short(self, entry):
        self._short.append(entry)
        self._short = self._short[-self.short_term_mem_skip:]

    # ---- hipGraph replay -----------------------------------------------------------------------
    def _gx(self):
        if self._graphs is None:
            self._graphs = FrameGraphs(next(self.AOT.parameters()).device)
        return self._graphs

    def _stage(self, name, t):
        """Copies a caller-owned input into a buffer with a stable address (one per name and shape), fp32."""
        if getattr(t, '_aot_stable', False) and t.dtype == torch.float32 and t.is_contiguous():
            return t         # the output of one of this engine's own replays (decode_current_labels): its address does not change
        key = (name, tuple(t.shape))
        buf = self._static.get(key)
        if buf is None:
            buf = self._static[key] = torch.empty(tuple(t.shape), dtype=torch.float32, device=t.device)
        buf.copy_(t)
        return buf

    # ---- look-ahead encoding -------------------------------------------------------------------
    def _side(self):
        """Side stream + its own graph cache (own capture stream, own memory pool: its replays run BESIDE the frame graphs) of the
        overlapped look-ahead."""
        if getattr(self, '_side_stream', None) is None:
            dev = next(self.AOT.parameters()).device
            self._side_stream = torch.cuda.Stream(dev)
            self._side_graphs = FrameGraphs(dev)
        return self._side_stream, self._side_graphs

    def _encode_batch(self, src, slot):
        ws = self.AOT.ws
        ws.salt = '' if slot == 0 else '#ahead%d' % slot      # the second feature set lives in buffers of its own
        try:
            return self.AOT.encode_tokens(src)
        finally:
            ws.salt = ''

    @_in_table
    def encode_ahead(self, imgs, overlap=False):
        """Optional: encodes the NEXT frames of the clip (a list of [1,3,H,W] tensors, in the order they will be matched) as
        ONE batch, on the current stream -- the encoder does not depend on the memory state (the reference's offline_encoder,
        aot_engine.py:147-166, likewise encodes every frame it has at once), and a batch of three 480p frames fills the 256
        CUs where one frame cannot (tile counts of the stride-8 / 16 stages triple).  The match_propogate_one_frame calls that
        receive THE SAME tensors pick their features up; any other image is encoded in line as before.  The batch lives in
        per-stream scratch: a new call replaces whatever an earlier call left unused.  Same arithmetic per output element up
        to the split-K summation order of the GEMM dispatch (parity against the reference: the whole-clip golden tests run
        with look-ahead on and off).  No-op for encoders that take one image per call.
        overlap=True (round 6): the batch is encoded on a SIDE stream while the caller goes on propagating the frames of the
        previous batch -- the stride-16 stages of a propagated frame (LSTT, the head's first blocks) leave more than half of the
        CUs idle, and the encoder of the coming frames fills them.  Two feature sets alternate (a call replaces the set encoded
        two calls ago, which the caller has consumed by then); a frame's first use waits for its batch's event.  Same kernels,
        same arithmetic: bit-identical to overlap=False."""
        if not overlap:
            self._ahead = {}
        if not imgs or len(imgs) < 2 or not getattr(self.AOT.encoder, 'batched', False):
            if not imgs:
                self._ahead = {}
            return           # (one frame: nothing to batch -- and a B = 1 encode would share the in-line encoder's scratch)
        slot = 0
        if overlap:
            slot = self._ahead_slot = 1 - getattr(self, '_ahead_slot', 1)
            self._ahead = {k_: v for k_, v in self._ahead.items() if v[2] != slot}
        k = len(imgs)
        key = ('imgs_ahead', k, tuple(imgs[0].shape), slot)
        src = self._static.get(key)          # the batch is gathered into one [k,3,H,W] buffer (stable address: replayable)
        if src is None:
            src = self._static[key] = torch.empty((k,) + tuple(imgs[0].shape[1:]), dtype=torch.float32, device=imgs[0].device)
        for b, img in enumerate(imgs):
            src[b:b + 1].copy_(img)
        done = None
        if overlap:
            cur = torch.cuda.current_stream()
            side, graphs = self._side()
            ev = torch.cuda.Event()
            ev.record(cur)                   # the images are in place, and the frames that read this slot's previous features are issued
            side.wait_event(ev)
            with torch.cuda.stream(side):
                if self.use_graph:
                    feats = graphs.run(ptr_key('encode_ahead', src, aot_hip.gemm_table(), slot), lambda: self._encode_batch(src, slot))
                else:
                    feats = self._encode_batch(src, slot)
                done = torch.cuda.Event()
                done.record(side)
        elif self.use_graph:
            feats = self._gx().run(ptr_key('encode_ahead', src, aot_hip.gemm_table()), lambda: self.AOT.encode_tokens(src))
        else:
            feats = self.AOT.encode_tokens(src)
        for b, img in enumerate(imgs):
            # the entry holds the image: its storage cannot be freed and re-used for another frame of the same shape while the
            # features wait, so (address, shape, version) identifies the frame for as long as the entry lives
            self._ahead[_img_key(img)] = (img, feats.frame(b) if hasattr(feats, 'frame') else
                                          [(f[b * h * w:(b + 1) * h * w], h, w) for (f, h, w) in feats], slot, done)

    def _take_ahead(self, img):
        """Features of a frame encoded by encode_ahead(), if `img` is the same memory, unmodified since."""
        if img is None or not self._ahead:
            return None
        hit = self._ahead.pop(_img_key(img), None)
        if hit is None:
            return None
        if hit[3] is not None:          # encoded on the side stream: this stream continues when the batch is done
            torch.cuda.current_stream().wait_event(hit[3])
        return hit[1]

    # ---- frame stages --------------------------------------------------------------------------
    def _encode(self, img, img_embs):
        if img_embs is None:
            feats = self.AOT.encode_tokens(img)
        else:   # image embedding computed by another cohort / engine (aot_engine.py:606-607,612-616)
            feats = [(to_tokens(e), e.shape[2], e.shape[3]) for e in img_embs]
        self._feats = feats
        return feats

    def _group_objects(self):
        """Objects of this cohort: obj_nums is the cohort's own count (the clip's total minus the groups before it)."""
        return int(self.obj_nums[0]) if isinstance(self.obj_nums, (list, tuple)) else int(self.obj_nums)

    def assign_identity(self, one_hot_mask):
        """aot_engine.py:168-179 (+ utils/image.py:69-74): the reference's argument, a one-hot or probability map
        [1, max_obj_num+1, H, W], goes through the dense identity convolution; a label map [1,1,H,W] (what this engine's own
        stages pass) through the fused gather.  Returns the id embedding token-major, [lanes*N, C]."""
        return self.AOT.id_emb_from_mask(one_hot_mask, self.enc_size_2d, lanes=self.lanes, group0=self.group0)

    @_in_table
    def add_reference_frame(self, img=None, mask=None, frame_step=-1, obj_nums=None, img_embs=None):
        if self.obj_nums is None and obj_nums is None:
            _die('No objects for reference frame!')
        if obj_nums is not None:
            self.obj_nums = obj_nums
        if img is None and img_embs is None:
            _die('No image for reference frame!')
        if mask is None:
            _die('No mask for reference frame!')
        if self.mfma == 'bf16x6':
            self.AOT.pack()
            aot_hip.pack_bf16x6_all()        # (no-op once done; never inside a capture -- this stage is launched from the host)
        feats = self._encode(img, img_embs)
        x16, h, w = feats[3]
        if self.input_size_2d is None:
            self.update_size(img.size()[2:] if img is not None else mask.size()[2:], (h, w))
        if self.pos_emb is None:
            pos = to_tokens(self.AOT.get_pos_emb(as_map(x16, h, w)).contiguous(
                memory_format=torch.channels_last)).contiguous()
            # (kept at ONE address per geometry for the engine's lifetime -- like everything a replayed stage reads: a tensor of its
            #  own per clip would give every clip new graph keys and a new round of captures)
            key = ('pos_emb', tuple(pos.shape))
            buf = self._static.get(key)
            if buf is None:
                buf = self._static[key] = torch.empty_like(pos)
            buf.copy_(pos)
            buf._aot_pos_qkv = None          # (the buffer outlives clips and weight reloads: maps of an earlier clip never leak into this one)
            self.pos_emb = buf
            if hasattr(self.AOT.LSTT, 'prepare_pos') and not os.environ.get('AOT_NO_QKV_MERGE'):     # AOT: the position term of the merged Q|K|V product, once per clip (env: A/B runs)
                key = ('pos_qkv', tuple(pos.shape))
                outs = self._static.get(key)
                if outs is None:
                    outs = self._static[key] = [torch.empty(pos.shape[0], 3 * pos.shape[1], dtype=torch.float32, device=pos.device)
                                                for _ in self.AOT.LSTT.layers]
                self.AOT.LSTT.prepare_pos(self.pos_emb, aot_hip.stream_ptr(), outs)
        id_emb = self.assign_identity(mask)
        self.curr_id_embs = id_emb
        stream = aot_hip.stream_ptr()
        # the reference frame memorises itself (aot_engine.py:243-251): K and the id-fused V go to the next bank slot
        slot = self._next_slot()
        self._ensure_bank(slot + 1)
        direct = self._direct()
        dst = self._slot_views(slot) if direct else self._scratch_set()
        self._dec_in, mems = self.AOT.LSTT.run(x16, None, None, id_emb, self.pos_emb, self.enc_size_2d, self.AOT.ws, stream,
                                               B=self.lanes, dst=dst, keep=self._arena)
        self._curr = mems
        if not direct:
            self._store(dst, slot)
        else:
            self._pack_x6(dst, slot, None, self._brows_bank())
        self._commit(slot)
        self._curr_slot = None
        self._dst = dst
        self.last_mem_step = self.frame_step
        rows = self._brows_bank() if direct else self.enc_hw
        self._short = [[(k, v, rows) for k, v in dst]]
        if self.mfma == 'bf16x6':
            aot_hip.pack_bf16x6_all()        # the weights this first frame packed lazily (encoder blocks)

    @_in_table
    def match_propogate_one_frame(self, img=None, img_embs=None):
        self.frame_step += 1
        T = self.bank_len
        # a frame that update_memory will memorise gets its K / V written straight into its bank slot
        self._curr_slot = None
        dst = None
        if self.frame_step - self.last_mem_step >= self.long_term_mem_gap:
            # the bank gets its room for this frame NOW, whichever way the frame will reach it (the attention launches are
            # planned for the bank's capacity: every engine mode must see the same capacity at the same frame)
            slot = self._next_slot()
            self._ensure_bank(slot + 1)          # (may re-allocate the bank: the views below are taken afterwards)
            if self._direct():
                self._curr_slot = slot
                dst = self._slot_views(slot)
        if dst is None:
            dst = self._scratch_set()
        brows = self._brows_bank()
        sf = self._state_free
        tdev = (self._dev_int(0, T),) if sf else ()
        long_m = [(k.view(-1, k.shape[2]), v.view(-1, v.shape[2]), T, brows) + tdev for k, v in self._bank]
        short = self._short[0]
        self._dst = dst
        x6 = {'x6': self._bank_x6} if self._x6_attn else {}

        ahead = self._take_ahead(img) if img_embs is None else None

        def launch(img_, embs_):
            feats = ahead if ahead is not None else self._encode(img_, embs_)
            dec_in, mems = self.AOT.LSTT.run(feats[3][0], long_m, short, None, self.pos_emb, self.enc_size_2d, self.AOT.ws,
                                             aot_hip.stream_ptr(), B=self.lanes, dst=dst, keep=self._arena, **x6)
            return feats, dec_in, mems

        if self.use_graph:
            src = self._stage('img', img) if (img_embs is None and ahead is None) else None
            # state-free: the bank length is not part of the key (only whether the bank is still the reference frame alone:
            # that launch is planned for its exact length)
            bank_state = T <= self.enc_hw if sf else T
            key = ptr_key('match', src, img_embs, [f[0] for f in ahead] if ahead is not None else None,
                          [m[:2] for m in long_m], brows, bank_state, short, dst, self.pos_emb, self.lanes, self.enc_size_2d,
                          aot_hip.gemm_table(), [b[0] for b in self._bank_x6] if self._x6_attn else None)
            self._feats, self._dec_in, self._curr = self._gx().run(key, lambda: launch(src, img_embs))
        else:
            self._feats, self._dec_in, self._curr = launch(img, img_embs)

    def decode_stride4(self):
        """Runs the decoder for the cohort's lanes: stride-4 logits [lanes*h4*w4, max_obj+1] (a scratch view), h4, w4."""
        f4, f8, f16, _ = self._feats
        dec = self.AOT.decoder
        x_in = self._dec_in
        if not dec.decode_intermediate_input and x_in.shape[1] != dec.in_dim:
            # AOT with MODEL_DECODER_INTERMEDIATE_LSTT = False: the decoder takes the last LSTT output only (aot.py:86-92),
            # the last column block of the concatenated buffer
            x_in = x_in[:, -dec.in_dim:]
        ads = getattr(self._feats, 'ads', None)
        if ads is not None:
            return dec.run(x_in, f16, f8, f4, self.AOT.ws, aot_hip.stream_ptr(), B=self.lanes, ads=ads)
        return dec.run(x_in, f16, f8, f4, self.AOT.ws, aot_hip.stream_ptr(), B=self.lanes)

    @_in_table
    def decode_current_logits(self, output_size=None):
        """Single-cohort form of the reference call (aot_engine.py:356-380)."""
        return _decode(self, [self], output_size)

    @_in_table
    def decode_current_labels(self, output_size):
        """Extension (round 5): the evaluator's next two steps folded into the decode -- returns (label [1,1,OH,OW], label resized to
        the input size [1,1,H,W]): argmax of the softmax of decode_current_logits(output_size) (evaluator.py:332-352 with one
        augmentation) and its nearest resize (:375-381), bit-identical to those calls, without the output-size logits in memory."""
        return _decode(self, [self], output_size, labels=True)

    def update_long_term_memory(self, new_long_term_memories):
        """Reference signature (aot_engine.py:291-305): list over layers of [K, V] ([N,1,C]), lane 0; appended."""
        slot = self._next_slot()
        self._ensure_bank(slot + 1)
        self._store([(to_tokens(m[0]).contiguous(), to_tokens(m[1]).contiguous()) for m in new_long_term_memories], slot)
        self._commit(slot)

    @_in_table
    def update_short_term_memory(self, curr_mask, curr_id_emb=None, skip_long_term_update=False):
        """curr_mask: label map [1,1,H,W], or a probability map [1,max_obj_num+1,H,W] (one lane).  curr_id_emb, when given,
        replaces the mask's identity embedding: [N, C] token-major as assign_identity returns it, or the reference's [N,1,C]."""
        if curr_id_emb is not None:
            curr_id_emb = curr_id_emb.reshape(-1, curr_id_emb.shape[-1]).float().contiguous()
            if curr_id_emb.shape[0] != self.lanes * self.enc_hw:
                raise ValueError('curr_id_emb has %d tokens, the frame has %d' % (curr_id_emb.shape[0], self.lanes * self.enc_hw))
        dst, curr = self._dst, self._curr
        in_bank = self._curr_slot is not None
        memorise = self.frame_step - self.last_mem_step >= self.long_term_mem_gap
        store_slot = None
        if memorise and not skip_long_term_update and not in_bank:
            store_slot = self._next_slot()
            self._ensure_bank(store_slot + 1)

        slot_dev = self._dev_int(1, store_slot) if (self._state_free and store_slot is not None) else None

        def launch(mask_):
            self.AOT.update_memory_values(curr, mask_, self.enc_size_2d, self.lanes, self.group0, [d[1] for d in dst],
                                          aot_hip.stream_ptr(), id_emb=mask_ if curr_id_emb is not None else None)
            if store_slot is not None:
                self._store(dst, store_slot, slot_dev)
            elif in_bank and memorise and not skip_long_term_update:
                self._pack_x6(dst, self._curr_slot, None, self._brows_bank())     # (written in place: the packed copy follows)

        if curr_id_emb is not None:
            curr_mask = curr_id_emb
        if self.use_graph:
            src = self._stage('mask' if curr_id_emb is None else 'id_emb', curr_mask)
            key = ptr_key('update', src, curr, dst, (store_slot is not None) if self._state_free else store_slot,
                          [b[0] for b in self._bank] if store_slot is not None else None,
                          self.lanes, self.group0, self.enc_size_2d, aot_hip.gemm_table(),
                          [b[0] for b in self._bank_x6] if (self._x6_attn and memorise) else None,
                          self._curr_slot if (in_bank and memorise) else None)
            self._gx().run(key, lambda: launch(src))
        else:
            launch(curr_mask)
        rows = self._brows_bank() if in_bank else self.enc_hw
        self._push_short([(k, v, rows) for k, v in dst])
        if memorise:
            if not skip_long_term_update:
                self._commit(self._curr_slot if in_bank else store_slot)
            elif in_bank:
                # the slot is not kept: move this frame's K / V out before the next memorised frame overwrites it
                keep = self._scratch_set()
                for (k2, v2), (k, v) in zip(keep, dst):
                    k2.copy_(k[:self.enc_hw])
                    v2.copy_(v[:self.enc_hw])
                self._short[-1] = [(k, v, self.enc_hw) for k, v in keep]
            self.last_mem_step = self.frame_step
        self._curr_slot = None

    def predict_current_mask(self, output_size=None, return_prob=False):
        if output_size is None:
            output_size = self.input_size_2d
        logits = F.interpolate(self.pred_id_logits, size=output_size, mode='bilinear', align_corners=self.align_corners)
        pred_mask = torch.argmax(logits, dim=1)
        if not return_prob:
            return pred_mask
        return pred_mask, torch.softmax(logits, dim=1)


def _finalize(owner, cohorts, logits, h4, w4, output_size, stream):
    """Masks unused ids and writes the stride-4 planar logits ([G, C, h4, w4]) and the output-size logits of all G lanes:
    plain logits for one group, the reference's soft aggregation (aot_engine.py:565-582) for several -- one kernel either
    way.  The cohorts hold consecutive object groups; every group but the last is full, so the object count of the lanes
    in view is the sum of the cohorts' counts.  Returns (out4, out | None)."""
    nc = logits.shape[1]
    G = sum(c.lanes for c in cohorts)
    dev = logits.device
    objects = sum(c._group_objects() for c in cohorts)
    out4 = torch.empty(G, nc, h4, w4, dtype=torch.float32, device=dev)
    out = None
    oh = ow = 0
    if output_size is not None:
        oh, ow = int(output_size[0]), int(output_size[1])
        out = torch.empty(1, nc if G == 1 else 1 + G * (nc - 1), oh, ow, dtype=torch.float32, device=dev)
    aot_hip.logits_finalize(logits, out4, out, h4, w4, nc, oh, ow, objects, owner.align_corners, G=G, stream=stream)
    return out4, out


def _finalize_labels(owner, cohort, logits, h4, w4, output_size, stream):
    """The frame tail of ONE object group in one launch (aot_frame_tail_f32): planar stride-4 logits, the label map at the output
    size and its nearest resize to the input size (the memory update's mask) -- the output-size logits are never written.
    Returns (out4, (label_out, label_in))."""
    nc = logits.shape[1]
    dev = logits.device
    out4 = torch.empty(1, nc, h4, w4, dtype=torch.float32, device=dev)
    lab = torch.empty(1, 1, output_size[0], output_size[1], dtype=torch.float32, device=dev)
    lin = torch.empty(1, 1, int(cohort.input_size_2d[0]), int(cohort.input_size_2d[1]), dtype=torch.float32, device=dev)
    aot_hip.frame_tail(logits, out4, lab, lin, h4, w4, nc, cohort._group_objects(), owner.align_corners, stream=stream)
    lin._aot_stable = cohort.use_graph         # (a replayed stage rewrites the SAME tensor every frame: update_memory reads it where it lies)
    return out4, (lab, lin)


def _decode(owner, cohorts, output_size, labels=False):
    """decode_current_logits of one or several cohorts (aot_engine.py:356-380, 618-628): decoder, id masking, resize and
    group aggregation.  One cohort in graph mode: a single replay.  labels = True (decode_current_labels): returns
    (label at the output size, label at the input size) instead of the output-size logits -- for one object group the whole tail is
    ONE kernel inside the same replay, otherwise the logits path followed by aot_hip.fuse_probs / label_resize."""
    first = cohorts[0]
    if output_size is not None:
        output_size = (int(output_size[0]), int(output_size[1]))
    # (aot_frame_tail_f32 takes up to 16 classes: larger max_obj_num models go through logits + fuse_probs + label_resize)
    fused = labels and len(cohorts) == 1 and first.lanes == 1 and output_size is not None and first.AOT.max_obj_num + 1 <= 16

    def launch():
        stream = aot_hip.stream_ptr()
        if len(cohorts) == 1:
            logits, h4, w4 = first.decode_stride4()
        else:       # several cohorts: their stride-4 logits are gathered lane after lane into one buffer
            parts = []
            for c in cohorts:
                lg, h4, w4 = c.decode_stride4()
                parts.append(lg.clone())                    # the decoder's output scratch is shared by the cohorts
            # (rows padded to a multiple of four floats, like the decoder's own output: the finalize kernel then takes its
            #  16-byte loads, the path every multi-lane test runs)
            buf = torch.zeros(sum(p.shape[0] for p in parts), (parts[0].shape[1] + 3) // 4 * 4, dtype=torch.float32,
                              device=parts[0].device)
            r = 0
            for lg in parts:
                buf[r:r + lg.shape[0], :lg.shape[1]].copy_(lg)
                r += lg.shape[0]
            logits = buf[:, :parts[0].shape[1]]
        if fused:
            return _finalize_labels(owner, first, logits, h4, w4, output_size, stream)
        osz = output_size
        if osz is None and sum(c.lanes for c in cohorts) > 1:
            # several object groups, no output size: the reference aggregates the groups' stride-4 logits themselves
            # (aot_engine.py:565-582, 618-628) -- the same aggregation kernel with the resize an identity
            osz = (h4, w4)
        return _finalize(owner, cohorts, logits, h4, w4, osz, stream)

    if len(cohorts) == 1 and first.use_graph:
        key = ptr_key('decode_labels' if fused else 'decode', first._dec_in, [f[0] for f in first._feats],
                      list(getattr(first._feats, 'ads', None) or ()), output_size, first.lanes,
                      first._group_objects(), aot_hip.gemm_table(),
                      # the fused tail bakes the INPUT size into its launch (the nearest-resized feedback label): two input sizes can share
                      # every feature-map shape -- a key names every integer argument (graphs.py)
                      tuple(first.input_size_2d) if fused else None)
        out4, out = first._gx().run(key, launch)
    else:
        out4, out = launch()
    g = 0
    for c in cohorts:
        c.pred_id_logits = out4[g:g + c.lanes]
        g += c.lanes
    if labels and not fused:
        if out is None:
            _die('decode_current_labels needs an output size')
        lab = aot_hip.fuse_probs(out, [False], want_aug_labels=False)[0]
        return lab, aot_hip.label_resize(lab, int(first.input_size_2d[0]), int(first.input_size_2d[1]))
    if out is not None:
        return out
    return out4


class DeAOTEngine(AOTEngine):
    """reference networks/engines/deaot_engine.py:9-56.  The memory layout differences of DeAOT ([K 128 | V 512 | ID_V 512]
    per token, only ID_V refreshed at update time) live in the model (DeAOT.update_memory_values / mem_widths) and in
    GatedPropagationModule.run; the state machine is the same."""

    def __init__(self, aot_model, gpu_id=0, long_term_mem_gap=9999, short_term_mem_skip=1, layer_loss_scaling_ratio=2., **kw):
        super().__init__(aot_model, gpu_id, long_term_mem_gap, short_term_mem_skip, **kw)
        self.layer_loss_scaling_ratio = layer_loss_scaling_ratio      # kept (and, as in the reference, read by nothing)


class _GroupView:
    """What callers read from `engine.aot_engines[g]` (reference: one AOTEngine per object group): a window on lane `lane`
    of a cohort."""

    def __init__(self, cohort, lane):
        self._c, self._lane = cohort, lane

    def __getattr__(self, name):
        return getattr(self._c, name)

    @property
    def pred_id_logits(self):
        p = self._c.pred_id_logits
        return None if p is None else p[self._lane:self._lane + 1]

    def lstt_last(self):
        return self._c.lstt_last(self._lane)


class AOTInferEngine(nn.Module):
    """Caller-facing engine (reference aot_engine.py:485-635) for any number of objects."""

    cohort_cls = AOTEngine

    def __init__(self, aot_model, gpu_id=0, long_term_mem_gap=9999, short_term_mem_skip=1, max_aot_obj_num=None,
                 long_term_mem_max=None, graph=False, gemm_table='latency', mfma='f32'):
        super().__init__()
        if mfma not in aot_hip.MFMA_MODES:
            raise ValueError('mfma must be one of %s' % (aot_hip.MFMA_MODES,))
        self.mfma = mfma                 # matrix-core arithmetic of every cohort (see AOTEngine)
        if gemm_table not in aot_hip.GEMM_TABLES:
            raise ValueError('gemm_table must be one of %s' % sorted(aot_hip.GEMM_TABLES))
        self.gemm_table = gemm_table     # conv / linear dispatch table of every cohort of this engine (see AOTEngine)
        self.use_graph = bool(graph)    # replay captured hipGraphs per engine state (see AOTEngine.__init__, engines/graphs.py)
        self.cfg = aot_model.cfg
        self.AOT = aot_model
        if max_aot_obj_num is not None and max_aot_obj_num < aot_model.max_obj_num:
            raise NotImplementedError('groups narrower than the identity bank (max_aot_obj_num < MODEL_MAX_OBJ_NUM)')
        self.max_aot_obj_num = aot_model.max_obj_num
        self.gpu_id = gpu_id
        self.long_term_mem_gap = long_term_mem_gap
        self.short_term_mem_skip = short_term_mem_skip
        self.long_term_mem_max = long_term_mem_max       # bounded bank per object group (repo extension)
        self.align_corners = aot_model.cfg.MODEL_ALIGN_CORNERS
        self._cohorts = []
        self._spare = []             # cohorts of the previous clip: their bank buffers are re-used (no allocator traffic)
        self._last_geom = None       # frame geometry of the previous clip (scratch is released when it changes)
        self.restart_engine()

    # ---- reference attributes ------------------------------------------------------------------
    @property
    def aot_engines(self):
        return [_GroupView(c, i) for c in self._cohorts for i in range(c.lanes)]

    def restart_engine(self):
        self._spare = self._cohorts + self._spare
        self._cohorts = []
        self.obj_nums = None
        self.input_size_2d = self.enc_size_2d = self.enc_hw = None

    def _new_cohort(self, lanes, group0):
        for i, c in enumerate(self._spare):
            if c.lanes == lanes and c.first_group == group0:
                self._spare.pop(i)
                c.restart_engine()
                return c
        c = self.cohort_cls(self.AOT, self.gpu_id, self.long_term_mem_gap, self.short_term_mem_skip,
                            long_term_mem_max=self.long_term_mem_max, lanes=lanes, group0=group0, graph=self.use_graph,
                            gemm_table=self.gemm_table, mfma=self.mfma)
        c.eval()
        return c

    def _groups_needed(self, obj_nums):
        return max(-(-int(obj_nums) // self.max_aot_obj_num), 1)

    def _set_counts(self):
        """Every cohort learns how many of the clip's objects fall into its lanes.  With a single group the label map is
        used as it is (labels beyond max_obj_num contribute nothing, utils/image.py:69-74); with several, every group sees
        the other groups' pixels as background (separate_mask, aot_engine.py:515-534)."""
        k = self.max_aot_obj_num
        single = sum(c.lanes for c in self._cohorts) == 1
        for c in self._cohorts:
            c.obj_nums = [max(0, min(self.obj_nums - c.first_group * k, c.lanes * k))]
            c.group0 = None if single else c.first_group

    def _release_on_new_geometry(self, img):
        """Scratch, banks and arenas are sized for one frame geometry and kept from clip to clip.  A sequence set with mixed
        resolutions (YouTube-VOS, multi-scale testing) would otherwise keep one full set per geometry: when a clip starts
        at a size different from the previous clip's, this engine's sets are dropped first.  (Not in graph mode: captured
        graphs hold the addresses.)"""
        geom = (tuple(img.shape[-2:]), img.device)
        if self._last_geom is not None and geom != self._last_geom and not self.use_graph and not self._cohorts:
            self._spare = []
            self.AOT.ws.clear(stream=aot_hip.stream_ptr())
        self._last_geom = geom

    @_in_table
    def add_reference_frame(self, img, mask, obj_nums, frame_step=-1):
        if isinstance(obj_nums, (list, tuple)):
            obj_nums = obj_nums[0]
        self.obj_nums = int(obj_nums)
        self._release_on_new_geometry(img)
        have = sum(c.lanes for c in self._cohorts)
        need = self._groups_needed(self.obj_nums)
        if need > have:      # first frame: one cohort for every group; later: the groups opened by new objects
            self._cohorts.append(self._new_cohort(need - have, have))
        self._set_counts()
        img_embs = None
        for c in self._cohorts:
            c.add_reference_frame(img, mask, frame_step=frame_step, obj_nums=c.obj_nums, img_embs=img_embs)
            if img_embs is None:
                img_embs = c.curr_enc_embs
        first = self._cohorts[0]
        self.input_size_2d, self.enc_size_2d, self.enc_hw = first.input_size_2d, first.enc_size_2d, first.enc_hw

    def encode_ahead(self, imgs, overlap=False):
        """Optional look-ahead (see AOTEngine.encode_ahead): the next frames of the clip, encoded as one batch; overlap=True: on a side
        stream, beside the propagation of the frames encoded by the previous call."""
        if self._cohorts:
            self._cohorts[0].encode_ahead(imgs, overlap=overlap)

    @_in_table
    def match_propogate_one_frame(self, img=None):
        img_embs = None
        for c in self._cohorts:
            c.match_propogate_one_frame(img, img_embs=img_embs)
            if img_embs is None:
                img_embs = c.curr_enc_embs

    @_in_table
    def decode_current_logits(self, output_size=None):
        return _decode(self, self._cohorts, output_size)

    @_in_table
    def decode_current_labels(self, output_size):
        """(label at the output size, label at the input size): see AOTEngine.decode_current_labels; several object groups take the
        logits path + aot_hip.fuse_probs / label_resize."""
        return _decode(self, self._cohorts, output_size, labels=True)

    @_in_table
    def update_memory(self, curr_mask, skip_long_term_update=False):
        for c in self._cohorts:
            c.update_short_term_memory(curr_mask, skip_long_term_update=skip_long_term_update)

    # reference helpers kept for callers that use them directly (aot_engine.py:515-582) -----------
    def separate_mask(self, mask, obj_nums):
        """Label map -> per-group label maps (the engine itself never materialises them: the identity gather kernel
        separates on the fly)."""
        n = self._groups_needed(obj_nums)
        if n == 1:
            return [mask], [obj_nums]
        k = self.max_aot_obj_num
        counts = [k] * (n - 1) + [obj_nums - k * (n - 1)]
        outs = []
        for g in range(n):
            inside = (mask > g * k) & (mask <= (g + 1) * k)
            outs.append(torch.where(inside, mask - g * k, torch.zeros_like(mask)))
        return outs, counts

    def update_size(self):
        first = self._cohorts[0]
        self.input_size_2d, self.enc_size_2d, self.enc_hw = first.input_size_2d, first.enc_size_2d, first.enc_hw


class DeAOTInferEngine(AOTInferEngine):
    """reference networks/engines/deaot_engine.py:59-94."""
    cohort_cls = DeAOTEngine

"""AOT model (reference networks/models/aot.py:9-115): encoder + 1x1 projector + LSTT + identity bank +
sine positional embedding + FPN decoder, with the reference's method surface and state_dict layout.

Tensors crossing this API keep the reference's shapes ([1,C,h,w] maps, [N,1,C] sequences) but are views
of token-major (= channels-last) device buffers; inside, everything is [h*w, C] and every stage is a
hand-written gfx950 kernel (aot_hip).  There is no CPU path.
"""
import torch
import torch.nn as nn

import aot_hip
from networks.decoders import build_decoder
from networks.encoders import build_encoder
from networks.layers.normalization import fold_conv_bn
from networks.layers.position import PositionEmbeddingSine
from networks.layers.transformer import LongShortTermTransformer
from networks.layers.workspace import Workspace


def to_tokens(t):
    """[1,C,h,w] or [N,1,C] (reference shapes) -> token-major [N, C] with row stride ld >= C.
    Zero-copy when the memory already is channels-last / token-major (every tensor this package hands
    out is); a foreign NCHW-contiguous tensor is transposed once."""
    if t.dim() == 4:
        n, c, h, w = t.shape
        assert n == 1, 'the inference path is batch 1 per object group (aot_engine.py:445-447)'
        if t.stride(1) == 1 and t.stride(2) == w * t.stride(3) and t.stride(3) % 4 == 0:
            return torch.as_strided(t, (h * w, c), (t.stride(3), 1), t.storage_offset())
        return t.permute(0, 2, 3, 1).contiguous().view(h * w, c)
    if t.dim() == 3:
        n, b, c = t.shape
        assert b == 1
        if t.stride(2) == 1 and t.stride(0) % 4 == 0:
            return t[:, 0, :]
        return t.reshape(n, c).contiguous()
    return t


class Feats(list):
    """[(f4,h,w), (f8,h,w), (f16,h,w), (proj16,h,w)] of encode_tokens, plus `.ads` = the decoder's adapter maps (ad16, ad8, ad4) of the
    same frames when the decoder has adapters (None otherwise): a list to every caller of the reference surface."""
    ads = None

    def frame(self, b):
        """The b-th frame of a batch as a Feats of its own (row slices)."""
        out = Feats([(f[b * h * w:(b + 1) * h * w], h, w) for (f, h, w) in self])
        if self.ads is not None:
            out.ads = tuple(a[b * h * w:(b + 1) * h * w] for a, (_, h, w) in zip(self.ads, (self[2], self[1], self[0])))
        return out


def as_map(tok, h, w):
    """token-major [h*w, C] (row stride ld) -> [1,C,h,w] view (channels-last strides)."""
    ld = tok.stride(0)
    return torch.as_strided(tok, (1, tok.shape[1], h, w), (h * w * ld, 1, w * ld, ld), tok.storage_offset())


class AOT(nn.Module):
    def __init__(self, cfg, encoder='mobilenetv2', decoder='fpn'):
        super().__init__()
        self.cfg = cfg
        self.max_obj_num = cfg.MODEL_MAX_OBJ_NUM
        self.epsilon = cfg.MODEL_EPSILON
        emb = cfg.MODEL_ENCODER_EMBEDDING_DIM
        self.encoder = build_encoder(encoder, frozen_bn=cfg.MODEL_FREEZE_BN, freeze_at=cfg.TRAIN_ENCODER_FREEZE_AT)
        self.encoder_projector = nn.Conv2d(cfg.MODEL_ENCODER_DIM[-1], emb, kernel_size=1)
        self.LSTT = LongShortTermTransformer(
            cfg.MODEL_LSTT_NUM, emb, cfg.MODEL_SELF_HEADS, cfg.MODEL_ATT_HEADS,
            emb_dropout=cfg.TRAIN_LSTT_EMB_DROPOUT, droppath=cfg.TRAIN_LSTT_DROPPATH,
            lt_dropout=cfg.TRAIN_LSTT_LT_DROPOUT, st_dropout=cfg.TRAIN_LSTT_ST_DROPOUT,
            droppath_lst=cfg.TRAIN_LSTT_DROPPATH_LST, droppath_scaling=cfg.TRAIN_LSTT_DROPPATH_SCALING,
            intermediate_norm=cfg.MODEL_DECODER_INTERMEDIATE_LSTT, return_intermediate=True)
        # Long-video knobs of the reference's MultiheadAttention (attention.py:37-38,84-89,102-105).  The reference only
        # exposes them as constructor arguments that no caller sets; here two optional config keys reach them.
        for layer in self.LSTT.layers:
            layer.long_term_attn.top_k = int(getattr(cfg, 'MODEL_LT_TOP_K', -1))
            layer.long_term_attn.max_mem_len_ratio = float(getattr(cfg, 'MODEL_LT_MAX_MEM_LEN_RATIO', -1))
        decoder_indim = emb * (cfg.MODEL_LSTT_NUM + 1) if cfg.MODEL_DECODER_INTERMEDIATE_LSTT else emb
        self.decoder = build_decoder(decoder, in_dim=decoder_indim, out_dim=cfg.MODEL_MAX_OBJ_NUM + 1,
                                     decode_intermediate_input=cfg.MODEL_DECODER_INTERMEDIATE_LSTT, hidden_dim=emb,
                                     shortcut_dims=cfg.MODEL_ENCODER_DIM, align_corners=cfg.MODEL_ALIGN_CORNERS)
        if cfg.MODEL_ALIGN_CORNERS:
            self.patch_wise_id_bank = nn.Conv2d(cfg.MODEL_MAX_OBJ_NUM + 1, emb, kernel_size=17, stride=16, padding=8)
        else:
            self.patch_wise_id_bank = nn.Conv2d(cfg.MODEL_MAX_OBJ_NUM + 1, emb, kernel_size=16, stride=16, padding=0)
        self.id_dropout_p = float(getattr(cfg, 'TRAIN_LSTT_ID_DROPOUT', 0.))   # aot.py:65; training-time only
        self.pos_generator = PositionEmbeddingSine(emb // 2, normalize=True)
        self.ws = Workspace()
        self._packed = None
        self._params_touched = False      # set by a training step (models/train_forward.py): the packed copies are stale

    # ---- packing: kernel-layout copies of the parameters, built once ---------------------------
    def pack(self):
        if self._params_touched:          # an optimiser has been at the parameters since they were packed
            self.invalidate()
            self._params_touched = False
        if self._packed is None:
            if not next(self.parameters()).is_cuda:
                raise aot_hip.AotHipError('AOT must be moved to a ROCm device before inference (no CPU fallback)')
            aot_hip.load()
            idw = self.patch_wise_id_bank.weight.detach().float()            # [C, L, K, K]
            self._packed = {
                'proj': fold_conv_bn(self.encoder_projector),
                'id_table': idw.permute(1, 2, 3, 0).contiguous(),            # [L, K, K, C]
                'id_sumtab': idw.double().sum((2, 3)).t().float().contiguous(),  # [L, C]: all-taps sum per label
                'id_bias': self.patch_wise_id_bank.bias.detach().float().contiguous(),
                # the same bank as implicit-GEMM weights (channels padded to 12): identities of probability maps
                'id_dense': fold_conv_bn(self.patch_wise_id_bank, pad_cin=(idw.shape[1] + 3) // 4 * 4),
            }
            for layer in self.LSTT.layers:
                layer.pack()
            self.decoder.pack()
        return self._packed

    def prepare(self):
        """Packs every weight now and waits for the device: call once before inference fans out over several HIP streams
        (lazy packing on whichever stream touches the model first would race with the other streams' first kernels)."""
        self.pack()
        torch.cuda.synchronize(next(self.parameters()).device)
        return self

    def invalidate(self):
        """Drop packed weights (call after changing parameters)."""
        self._packed = None
        for m in self.modules():
            for attr in ('_p', '_stem', '_last'):
                if hasattr(m, attr):
                    setattr(m, attr, None)
            if hasattr(m, '_packed') and m is not self:
                m._packed = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate()
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.invalidate()
        self.ws.clear()
        return r

    # ---- token-major internals -----------------------------------------------------------------
    def encode_tokens(self, img, stream=None, out_cat=None):
        """img [B,3,H,W] -> [(f4,h,w), (f8,h,w), (f16,h,w), (proj16,h,w)] token-major, the B images stacked along the rows
        (B > 1 only for encoders with `batched = True`).  If out_cat is given (B = 1) the projected feature is written
        into its first C columns."""
        p = self.pack()
        stream = stream if stream is not None else aot_hip.stream_ptr()
        B = img.shape[0]
        if B > 1 and not getattr(self.encoder, 'batched', False):
            raise aot_hip.AotHipError('%s encodes one image per call' % type(self.encoder).__name__)
        feats = self.encoder.run(img.float().contiguous(), self.ws, stream)
        if len(feats) == 4:          # mobilenetv2: 4 stages; stage 3 (96 ch) is the 16x shortcut, stage 4 (1280 ch) feeds the projector
            f4, f8, f16, top = feats
        else:                        # resnet: stride-16 map is shortcut and projector input
            f4, f8, f16 = feats
            top = f16
        x, h, w = top
        emb = self.encoder_projector.out_channels
        if out_cat is None:
            out = self.ws.get('enc_proj16', (B * h * w, emb), img.device)     # per-stream scratch, like the encoder maps
        else:
            out = out_cat[:, :emb]
        aot_hip.conv2d(x, *p['proj'], out, h, w, x.shape[1], h, w, emb, B=B, stream=stream)
        feats = Feats([f4, f8, f16, (out, h, w)])
        if hasattr(self.decoder, 'adapters'):
            # the decoder's adapter convolutions read these maps only: formed here, for all B frames in one launch each
            feats.ads = self.decoder.adapters(f16, f8, f4, self.ws, stream, B=B)
        return feats

    def id_emb_from_mask(self, mask, size_2d, stream=None, lanes=1, group0=None, fuse=None, want_out=True):
        """Fused one_hot_mask + patch_wise_id_bank (aot.py:76-79, utils/image.py:69-74): label map [1,1,H,W]
        (float ids) -> id embedding [lanes*h*w, C]; the 18 MB one-hot tensor is never built.  group0 is not None: the map
        holds the labels of ALL objects and lane g is object group group0+g (the mask separation of AOTInferEngine,
        aot_engine.py:515-534, happens inside the gather).  fuse = [(add_i, out_i)]: out_i = id_emb + add_i, same launch.
        A map with max_obj_num+1 channels is a probability map (MODEL_USE_PREV_PROB) and takes the dense convolution."""
        p = self.pack()
        stream = stream if stream is not None else aot_hip.stream_ptr()
        if mask.dim() == 4 and mask.shape[1] != 1:
            if lanes != 1 or group0 is not None:
                # (the reference splits the probability map of several groups along the BATCH axis, aot_engine.py:536-545,
                #  which leaves every group but the first with an empty map: there is no behaviour to match)
                raise NotImplementedError('probability-map identities are defined for one object group (<= %d objects)'
                                          % self.max_obj_num)
            out = self.id_emb_from_prob(mask, size_2d, stream)
            for add_i, out_i in (fuse or []):
                aot_hip.add(add_i, out, out_i, stream=stream)
            return out
        H, W = mask.shape[-2:]
        h, w = size_2d
        conv = self.patch_wise_id_bank
        K, s, pd = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        out = torch.empty(lanes * h * w, conv.out_channels, dtype=torch.float32, device=mask.device) if want_out else None
        aot_hip.idbank(mask.float().contiguous(), p['id_table'], p['id_bias'], out, H, W, h, w, K, s, pd,
                       conv.out_channels, self.max_obj_num + 1, sumtab=p['id_sumtab'], G=lanes,
                       group_size=0 if group0 is None else self.max_obj_num, group0=group0 or 0, fuse=fuse, stream=stream)
        return out

    def id_emb_from_prob(self, prob, size_2d, stream=None):
        """patch_wise_id_bank as the dense convolution it is in the reference (aot.py:76-79), for maps that are not one-hot:
        prob [1, max_obj_num+1, H, W] -> id embedding [h*w, C].  The planar map is repacked token-major with the channels
        padded to a multiple of four and goes through the implicit-GEMM kernel (K = KH*KW*12 taps per output token)."""
        p = self.pack()
        stream = stream if stream is not None else aot_hip.stream_ptr()
        conv = self.patch_wise_id_bank
        L = self.max_obj_num + 1
        if prob.dim() != 4 or prob.shape[0] != 1 or prob.shape[1] != L:
            raise aot_hip.AotHipError('identity map must be [1, %d, H, W], got %s' % (L, tuple(prob.shape)))
        Lp = (L + 3) // 4 * 4
        H, W = prob.shape[-2:]
        h, w = size_2d
        K, s, pd = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        if (H + 2 * pd - K) // s + 1 != h or (W + 2 * pd - K) // s + 1 != w:
            raise aot_hip.AotHipError('identity map %dx%d does not give the %dx%d token grid' % (H, W, h, w))
        x = self.ws.get('id_prob_nhwc', (H * W, Lp), prob.device)
        aot_hip.nchw_to_nhwc(prob.float().contiguous(), x, L, H, W, Lp, stream=stream)
        out = torch.empty(h * w, conv.out_channels, dtype=torch.float32, device=prob.device)
        aot_hip.conv2d(x, *p['id_dense'], out, H, W, Lp, h, w, conv.out_channels, K, K, s, pd, 1, stream=stream)
        return out

    def update_memory_values(self, mems, mask, size_2d, lanes, group0, dst, stream, id_emb=None):
        """After the frame's mask is known (aot_engine.py:307-338): per LSTT layer V <- linear_V(V + id_emb(mask)).  One
        gather launch forms V + id_emb for every layer, one GEMM per layer writes the fused V to dst[i].  id_emb given
        ([lanes*N, C], e.g. assign_identity's): it is used as it is and `mask` is ignored."""
        L = len(mems)
        dev = mems[0][1].device
        sums = [self.ws.get('idsum_%d' % i, tuple(mems[i][1].shape), dev) for i in range(L)]
        if id_emb is not None:
            for i in range(L):
                aot_hip.add(mems[i][1], id_emb, sums[i], stream=stream)
        else:
            self.id_emb_from_mask(mask, size_2d, stream, lanes=lanes, group0=group0,
                                  fuse=[(mems[i][1], sums[i]) for i in range(L)], want_out=False)
        return self.LSTT.update_values(mems, sums, self.ws, stream, dst=dst)

    def mem_widths(self):
        """(key width, value width) of one memorised token per layer."""
        C = self.encoder_projector.out_channels
        return [(C, C) for _ in self.LSTT.layers]

    # ---- reference method surface (aot.py:72-108) ------------------------------------------------
    def get_pos_emb(self, x):
        return self.pos_generator(x)

    def get_id_emb(self, x):
        """x: one-hot or probability map [1, max_obj+1, H, W] (reference signature).  A one-hot map is converted to a label
        map (all-zero columns -> label -1, contributing nothing) and sent through the fused gather kernel; anything else takes
        the dense convolution.  (Reference surface, not on the per-frame path: the one-hot test reads back one flag.)"""
        H, W = x.shape[-2:]
        conv = self.patch_wise_id_bank
        h = (H + 2 * conv.padding[0] - conv.kernel_size[0]) // conv.stride[0] + 1
        w = (W + 2 * conv.padding[0] - conv.kernel_size[0]) // conv.stride[0] + 1
        if not bool(((x == 0) | (x == 1)).all()) or bool((x.sum(1) > 1).any()):
            return as_map(self.id_emb_from_mask(x, (h, w)), h, w)
        lab = x.argmax(1, keepdim=True).float()
        lab = torch.where(x.sum(1, keepdim=True) > 0, lab, torch.full_like(lab, -1.0))
        return as_map(self.id_emb_from_mask(lab, (h, w)), h, w)

    def encode_image(self, img):
        f4, f8, f16, top = self.encode_tokens(img)
        return [as_map(t.clone(), h, w) for (t, h, w) in (f4, f8, f16, top)]      # callers own what this API returns

    def decode_id_logits(self, lstt_emb, shortcuts):
        stream = aot_hip.stream_ptr()
        n, c, h, w = shortcuts[-1].shape
        toks = [to_tokens(shortcuts[-1])] + [to_tokens(e) for e in lstt_emb]
        cat = self._as_cat(toks)
        sc = [(to_tokens(s), s.shape[2], s.shape[3]) for s in shortcuts[:3]]
        x_in = cat if self.decoder.decode_intermediate_input else toks[-1]
        if x_in.shape[1] != self.decoder.in_dim:
            raise aot_hip.AotHipError('decoder expects %d input channels, got %d' % (self.decoder.in_dim, x_in.shape[1]))
        logits, h4, w4 = self.decoder.run(x_in.contiguous(), sc[2], sc[1], sc[0], self.ws, stream)
        out = torch.empty(1, logits.shape[1], h4, w4, dtype=torch.float32, device=logits.device)
        aot_hip.nhwc_to_nchw(logits, out, logits.shape[1], h4, w4, stream=stream)
        return out

    def _as_cat(self, toks):
        """Returns a [N, sum C] buffer holding the column-concatenation of ``toks``; zero-copy if they already
        are adjacent column blocks of one buffer (which is how LSTT_forward lays them out)."""
        base = toks[0]
        ld = base.stride(0)
        total = sum(t.shape[1] for t in toks)
        ok = ld >= total
        off = 0
        for t in toks:
            ok = ok and t.stride(0) == ld and t.data_ptr() == base.data_ptr() + 4 * off
            off += t.shape[1]
        if ok:
            return torch.as_strided(base, (base.shape[0], total), (ld, 1), base.storage_offset())
        return torch.cat(toks, 1)

    def _mems_in(self, mems, with_t):
        """reference-shaped per-layer memories -> the (K, V, [T,] rows-between-lanes) tuples LSTT.run takes (one lane)."""
        if mems is None:
            return None
        out = []
        for m in mems:
            k, v = to_tokens(m[0]).contiguous(), to_tokens(m[1]).contiguous()
            out.append((k, v, k.shape[0], k.shape[0]) if with_t else (k, v, k.shape[0]))
        return out

    def _mems_out(self, mems, long_in, short_in, size_2d):
        h, w = size_2d
        seq = lambda t: t.unsqueeze(1)
        curr = [[seq(m[0]), seq(m[1])] for m in mems]
        if long_in is None:        # reference frame: the frame memorises itself (K, id-fused V)
            long_ = [[seq(m[0]), seq(m[2])] for m in mems]
            short = [[as_map(m[0], h, w), as_map(m[2], h, w)] for m in mems]
        else:
            long_ = [[seq(m[0][:m[2]]), seq(m[1][:m[2]])] for m in long_in]
            short = [[as_map(m[0], h, w), as_map(m[1], h, w)] for m in short_in]
        return curr, long_, short

    def LSTT_forward(self, curr_embs, long_term_memories, short_term_memories, curr_id_emb=None, pos_emb=None,
                     size_2d=(30, 30)):
        """Reference surface (aot.py:94-108), one lane: [N,1,C] / [1,C,h,w] tensors in and out."""
        stream = aot_hip.stream_ptr()
        x = to_tokens(curr_embs[-1]).contiguous()
        pos = to_tokens(pos_emb).contiguous() if pos_emb is not None else None
        idt = to_tokens(curr_id_emb).contiguous() if curr_id_emb is not None else None
        lm, sm = self._mems_in(long_term_memories, True), self._mems_in(short_term_memories, False)
        dec_in, mems = self.LSTT.run(x, lm, sm, idt, pos, size_2d, self.ws, stream)
        curr, long_, short = self._mems_out(mems, lm, sm, size_2d)
        return self._lstt_outs(dec_in), curr, long_, short

    def _lstt_outs(self, dec_in):
        C = self.encoder_projector.out_channels
        L = self.LSTT.num_layers
        return [dec_in[:, (i + 1) * C:(i + 2) * C].unsqueeze(1) for i in range(L)]

"""DeAOT model (reference networks/models/deaot.py:8-55): AOT with the LSTT replaced by the dual-branch gated
propagation stack, a decoder fed by the last GPM output only, and a LayerNorm on the identity embedding."""
import torch
import torch.nn as nn

import aot_hip
from networks.decoders import build_decoder
from networks.layers.transformer import DualBranchGPM
from networks.models.aot import AOT, as_map, to_tokens


class DeAOT(AOT):
    def __init__(self, cfg, encoder='mobilenetv2', decoder='fpn'):
        super().__init__(cfg, encoder, decoder)
        emb = cfg.MODEL_ENCODER_EMBEDDING_DIM
        self.LSTT = DualBranchGPM(
            cfg.MODEL_LSTT_NUM, emb, cfg.MODEL_SELF_HEADS, cfg.MODEL_ATT_HEADS,
            emb_dropout=cfg.TRAIN_LSTT_EMB_DROPOUT, droppath=cfg.TRAIN_LSTT_DROPPATH,
            lt_dropout=cfg.TRAIN_LSTT_LT_DROPOUT, st_dropout=cfg.TRAIN_LSTT_ST_DROPOUT,
            droppath_lst=cfg.TRAIN_LSTT_DROPPATH_LST, droppath_scaling=cfg.TRAIN_LSTT_DROPPATH_SCALING,
            intermediate_norm=cfg.MODEL_DECODER_INTERMEDIATE_LSTT, return_intermediate=True)
        # long-video knobs of GatedPropagation (attention.py:674-679,689-693): constructor arguments nobody sets in the
        # reference; reached here through the same two optional config keys as AOT's
        for layer in self.LSTT.layers:
            layer.long_term_attn.top_k = int(getattr(cfg, 'MODEL_LT_TOP_K', -1))
            layer.long_term_attn.max_mem_len_ratio = float(getattr(cfg, 'MODEL_LT_MAX_MEM_LEN_RATIO', -1))
        decoder_indim = emb * (cfg.MODEL_LSTT_NUM * 2 + 1) if cfg.MODEL_DECODER_INTERMEDIATE_LSTT else emb * 2
        self.decoder = build_decoder(decoder, in_dim=decoder_indim, out_dim=cfg.MODEL_MAX_OBJ_NUM + 1,
                                     decode_intermediate_input=cfg.MODEL_DECODER_INTERMEDIATE_LSTT, hidden_dim=emb,
                                     shortcut_dims=cfg.MODEL_ENCODER_DIM, align_corners=cfg.MODEL_ALIGN_CORNERS)
        self.id_norm = nn.LayerNorm(emb)

    def id_emb_from_mask(self, mask, size_2d, stream=None, lanes=1, group0=None, fuse=None, want_out=True):
        """fused one-hot + id bank (models/aot.py:76-79) followed by LayerNorm over channels (deaot.py:51-55)."""
        stream = stream if stream is not None else aot_hip.stream_ptr()
        raw = super().id_emb_from_mask(mask, size_2d, stream, lanes=lanes, group0=group0)
        out = torch.empty_like(raw)
        aot_hip.layernorm(raw, self.id_norm.weight, self.id_norm.bias, out, stream=stream)
        return out

    def update_memory_values(self, mems, mask, size_2d, lanes, group0, dst, stream, id_emb=None):
        """deaot_engine.py:20-56: only ID_V is refreshed with the new identity embedding (in place in the frame's
        [V | ID_V] buffers, which already are dst); K and V stay as produced.  id_emb given: used instead of the mask's."""
        if id_emb is None:
            id_emb = self.id_emb_from_mask(mask, size_2d, stream, lanes=lanes, group0=group0)
        return self.LSTT.update_values(mems, id_emb, self.ws, stream, dst=dst)

    def mem_widths(self):
        return [(l.d_att, 2 * l.expand_d_model) for l in self.LSTT.layers]

    # reference-shaped memories: [K, V, None, ID_V] per layer (transformer.py:655-657) <-> token-major (K, [V | ID_V])
    def _mems_in(self, mems, with_t):
        if mems is None:
            return None
        out = []
        for m in mems:
            k = to_tokens(m[0]).contiguous()
            vcat = torch.cat([to_tokens(m[1]), to_tokens(m[3])], 1)
            out.append((k, vcat, k.shape[0], k.shape[0]) if with_t else (k, vcat, k.shape[0]))
        return out

    def _mems_out(self, mems, long_in, short_in, size_2d):
        h, w = size_2d
        E = self.LSTT.layers[0].expand_d_model
        seq = lambda t: t.unsqueeze(1)
        curr = [[seq(m[0]), seq(m[1][:, :E]), None, None if m[2] is None else seq(m[2])] for m in mems]
        if long_in is None:
            long_ = [[seq(m[0]), seq(m[1][:, :E]), None, seq(m[1][:, E:])] for m in mems]
            short = [[as_map(m[0], h, w), as_map(m[1][:, :E], h, w), None, as_map(m[1][:, E:], h, w)] for m in mems]
        else:
            long_ = [[seq(m[0][:m[2]]), seq(m[1][:m[2], :E]), None, seq(m[1][:m[2], E:])] for m in long_in]
            short = [[as_map(m[0], h, w), as_map(m[1][:, :E], h, w), None, as_map(m[1][:, E:], h, w)] for m in short_in]
        return curr, long_, short

    def _lstt_outs(self, dec_in):
        return [dec_in.unsqueeze(1)]

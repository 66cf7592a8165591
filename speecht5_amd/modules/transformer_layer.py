"""Encoder / decoder layer mirrors of SpeechT5/speecht5/models/modules/transformer_layer.py:23-411."""
import contextlib

import torch
import torch.nn as nn

from .. import functional as Fn
from .common import LayerNorm
from .multihead_attention import MultiheadAttention, RelPosKeys

_ACTS = {"gelu": Fn.ACT_GELU, "relu": Fn.ACT_RELU, "tanh": Fn.ACT_TANH, "linear": Fn.ACT_NONE}


def _relays(x, n):
    """n gradient relays for the pre-LN blocks of a layer whose input is x (None each when no gradient will flow).
    OFF by default (ST5_PRELN_RELAY=1 switches it on): measured on Large B = 32, same box, alternating -- 126.8 / 127.3 ms with the
    relays against 126.3 / 127.0 without; the 240 add kernels it removes were not on the critical path and the LayerNorm backward, a
    latency-bound kernel, pays for the third operand stream (profiles/r6b_knob_ab.txt)."""
    import os
    on = torch.is_grad_enabled() and x.requires_grad and os.environ.get("ST5_PRELN_RELAY", "0") == "1"
    return [Fn.GradRelay() if on else None for _ in range(n)]


class TransformerSentenceEncoderLayer(nn.Module):
    def __init__(self, embedding_dim=768, ffn_embedding_dim=3072, num_attention_heads=8, dropout=0.1,
                 attention_dropout=0.1, activation_dropout=0.1, activation_fn="relu", layer_norm_first=False,
                 has_relative_attention_bias=False):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.dropout = dropout
        self.activation_dropout = activation_dropout
        self.act = _ACTS[activation_fn]
        self.self_attn = MultiheadAttention(embedding_dim, num_attention_heads, dropout=attention_dropout,
                                            self_attention=True, has_relative_attention_bias=has_relative_attention_bias)
        self.layer_norm_first = layer_norm_first
        self.self_attn_layer_norm = LayerNorm(embedding_dim)
        self.fc1 = nn.Linear(embedding_dim, ffn_embedding_dim)
        self.fc2 = nn.Linear(ffn_embedding_dim, embedding_dim)
        self.final_layer_norm = LayerNorm(embedding_dim)
        if has_relative_attention_bias:
            self.norm_k = LayerNorm(embedding_dim // num_attention_heads)

    def gate_ok(self):
        """A LayerDrop gate (functional.LayerDropGate) can ride on this layer: its last operation is a LayerNorm."""
        return not self.layer_norm_first

    def forward_rows(self, x, B, T, padding_mask=None, pos_bias=None, gate=None):
        """x rows [B*T, C] -> rows.  Post-LN (:112-132) or pre-LN (:90-111).  gate: LayerDrop select folded into the last LayerNorm."""
        assert gate is None or self.gate_ok()
        tr = self.training
        p, pa = (self.dropout if tr else 0.0), (self.activation_dropout if tr else 0.0)
        x = Fn.layer_boundary(x, self)
        if self.layer_norm_first:
            # y = x + f(LN(x)): the residual's gradient CAN be relayed from the Linear / FFN that adds it into the LayerNorm's backward
            # kernel (round 6, an A/B mode -- see _relays: autograd's add kernel per block, 240 launches per Large update, is faster)
            rel = _relays(x, 2)
            h = self.self_attn_layer_norm(x, q8=True, relay=rel[0])   # (consumer: the QKV projection; fp8 mode takes its fp8 image from this pass)
            pb = pos_bias
            if pos_bias is not None:
                pb = RelPosKeys(self.norm_k(pos_bias.table), pos_bias.maxlen)
            x, _ = self.self_attn.forward_rows(h, B, T, key_padding_mask=padding_mask, position_bias=pb, residual=x, out_dropout=p,
                                               res_relay=rel[0])
            h = self.final_layer_norm(x, q8=True, relay=rel[1])       # (consumer: fc1)
            x = Fn.ffn(h, x, self.fc1, self.fc2, self.act, pa, p, relay_out=rel[1])
        else:
            x, _ = self.self_attn.forward_rows(x, B, T, key_padding_mask=padding_mask, position_bias=pos_bias, residual=x,
                                               out_dropout=p)
            x = self.self_attn_layer_norm(x)
            x = Fn.ffn(x, x, self.fc1, self.fc2, self.act, pa, p)
            x = self.final_layer_norm(x, gate=gate)
        return x

    def forward(self, x, self_attn_mask=None, self_attn_padding_mask=None, need_weights=False, att_args=None, pos_bias=None):
        T, B, C = x.shape
        rows = Fn.as_compute(x.transpose(0, 1).contiguous()).view(B * T, C)
        y = self.forward_rows(rows, B, T, self_attn_padding_mask, pos_bias)
        return y.view(B, T, C).transpose(0, 1), None


class TransformerDecoderLayer(nn.Module):
    def __init__(self, args, no_encoder_attn=False, add_bias_kv=False, add_zero_attn=False, has_relative_attention_bias=False):
        super().__init__()
        self.embed_dim = args.decoder_embed_dim
        self.num_updates = 0
        self.dropout = args.dropout
        self.freeze_decoder_updates = getattr(args, "freeze_decoder_updates", 0)
        self.self_attn = MultiheadAttention(self.embed_dim, args.decoder_attention_heads, dropout=args.attention_dropout,
                                            self_attention=True)
        self.act = _ACTS[str(getattr(args, "activation_fn", None) or "relu")]
        pa = getattr(args, "activation_dropout", 0) or 0
        if pa == 0:
            pa = getattr(args, "relu_dropout", 0) or 0
        self.activation_dropout = float(pa)
        self.normalize_before = args.decoder_normalize_before
        self.self_attn_layer_norm = LayerNorm(self.embed_dim)
        if no_encoder_attn:
            self.encoder_attn = None
            self.encoder_attn_layer_norm = None
        else:
            self.encoder_attn = MultiheadAttention(self.embed_dim, args.decoder_attention_heads,
                                                   kdim=getattr(args, "encoder_embed_dim", None),
                                                   vdim=getattr(args, "encoder_embed_dim", None),
                                                   dropout=args.attention_dropout, encoder_decoder_attention=True)
            self.encoder_attn_layer_norm = LayerNorm(self.embed_dim)
        self.fc1 = nn.Linear(self.embed_dim, args.decoder_ffn_embed_dim)
        self.fc2 = nn.Linear(args.decoder_ffn_embed_dim, self.embed_dim)
        self.final_layer_norm = LayerNorm(self.embed_dim)
        self.need_attn = True
        self.has_relative_attention_bias = has_relative_attention_bias
        if has_relative_attention_bias:
            self.norm_k = LayerNorm(self.embed_dim // args.decoder_attention_heads)  # unused (:241), kept for checkpoints

    def gate_ok(self):
        """A LayerDrop gate can ride on this layer: post-LN (the last operation is final_layer_norm) and nothing frozen."""
        return (not self.normalize_before) and self.freeze_decoder_updates <= self.num_updates

    def forward_rows(self, x, B, T, enc_rows, S, enc_padding_mask, self_padding_mask, causal, need_attn, kv_all=None, gate=None):
        """transformer_layer.py:262-404 without incremental state.  Returns (rows, cross-attn probs [B,H,T,S] or None).
        gate: LayerDrop select folded into the last LayerNorm."""
        assert gate is None or self.gate_ok()
        ft = self.freeze_decoder_updates <= self.num_updates
        tr = self.training
        p, pa = (self.dropout if tr else 0.0), (self.activation_dropout if tr else 0.0)
        nb = self.normalize_before
        x = Fn.layer_boundary(x, self)
        with torch.no_grad() if not ft else contextlib.ExitStack():
            r0 = _relays(x, 1)[0] if nb else None       # (pre-LN: the residual's gradient goes into the LayerNorm's backward kernel)
            h = self.self_attn_layer_norm(x, q8=True, relay=r0) if nb else x
            x, _ = self.self_attn.forward_rows(h, B, T, key_padding_mask=self_padding_mask, causal=causal, residual=x,
                                               out_dropout=p, res_relay=r0)
            if not nb:
                x = self.self_attn_layer_norm(x)
        attn = None
        if self.encoder_attn is not None and enc_rows is not None:
            r1 = _relays(x, 1)[0] if nb else None
            h = self.encoder_attn_layer_norm(x, q8=True, relay=r1) if nb else x
            x, attn = self.encoder_attn.forward_rows(h, B, T, kv=enc_rows, S=S, key_padding_mask=enc_padding_mask, residual=x,
                                                     out_dropout=p, need_weights=need_attn or (not tr and self.need_attn), kv_all=kv_all,
                                                     res_relay=r1)
            if not nb:
                x = self.encoder_attn_layer_norm(x)
        with torch.no_grad() if not ft else contextlib.ExitStack():
            r2 = _relays(x, 1)[0] if nb else None
            h = self.final_layer_norm(x, q8=True, relay=r2) if nb else x
            x = Fn.ffn(h, x, self.fc1, self.fc2, self.act, pa, p, relay_out=r2)
            if not nb:
                x = self.final_layer_norm(x, gate=gate)
        return x, attn

    @torch.no_grad()
    def forward_rows_cached(self, x, B, cache, enc_rows, S, enc_padding_mask, self_padding_mask, need_attn):
        """Incremental step (transformer_layer.py:262-404 with incremental_state): x [B, C] = the newest position only;
        `cache` = {"self": {...}, "cross": {...}} of this layer.  Returns (rows [B, C], cross-attn probs [B,H,1,S] or None)."""
        nb = self.normalize_before
        h = self.self_attn_layer_norm(x) if nb else x
        x, _ = self.self_attn.forward_rows_cached(h, B, cache.setdefault("self", {}), key_padding_mask=self_padding_mask, residual=x)
        if not nb:
            x = self.self_attn_layer_norm(x)
        attn = None
        if self.encoder_attn is not None and enc_rows is not None:
            h = self.encoder_attn_layer_norm(x) if nb else x
            x, attn = self.encoder_attn.forward_rows_cached(h, B, cache.setdefault("cross", {}), kv=enc_rows, S=S,
                                                            key_padding_mask=enc_padding_mask, residual=x,
                                                            need_weights=need_attn or self.need_attn)
            if not nb:
                x = self.encoder_attn_layer_norm(x)
        h = self.final_layer_norm(x) if nb else x
        x = Fn.ffn(h, x, self.fc1, self.fc2, self.act, 0.0, 0.0)
        if not nb:
            x = self.final_layer_norm(x)
        return x, attn

    def forward(self, x, encoder_out=None, encoder_padding_mask=None, incremental_state=None, prev_self_attn_state=None,
                prev_attn_state=None, self_attn_mask=None, self_attn_padding_mask=None, need_attn=False,
                need_head_weights=False, pos_bias=None):
        if need_head_weights:
            need_attn = True
        T, B, C = x.shape
        rows = Fn.as_compute(x.transpose(0, 1).contiguous()).view(B * T, C)
        enc_rows, S = None, None
        if encoder_out is not None:
            S = encoder_out.shape[0]
            enc_rows = Fn.as_compute(encoder_out.transpose(0, 1).contiguous()).view(B * S, -1)
        y, attn = self.forward_rows(rows, B, T, enc_rows, S, encoder_padding_mask, self_attn_padding_mask,
                                    self_attn_mask is not None, need_attn)
        if attn is not None:
            attn = attn.transpose(0, 1)  # [H,B,T,S]
            if not need_head_weights:
                attn = attn.mean(dim=0)
        return y.view(B, T, C).transpose(0, 1), attn, None

    def make_generation_fast_(self, need_attn=False, **kwargs):
        self.need_attn = need_attn

    def set_num_updates(self, num_updates):
        self.num_updates = num_updates

"""MultiheadAttention mirror of SpeechT5/speecht5/models/modules/multihead_attention.py:24-522 on the HIP kernels.

Same constructor / forward signature and parameter names (q_proj, k_proj, v_proj, out_proj).  The
relative-position bias is passed as a `RelPosKeys` handle (table + clip length) instead of the
materialised [T,T,hd] tensor the reference builds (encoder.py:240-244): the kernels gather the
bias from q.pe^T.  Not supported (unused by every SpeechT5 recipe): add_bias_kv, add_zero_attn,
quant-noise, arbitrary additive attn_mask (causal masks are recognised and handled in-kernel)."""
import torch
import torch.nn as nn

from .. import functional as Fn
from ..fairseq_compat import IncrementalState


class RelPosKeys:
    """Handle for Shaw-style relative keys: bias[i,j] = q_i . table[clip(i-j,-maxlen,maxlen-1)+maxlen]."""

    def __init__(self, table, maxlen):
        self.table = table      # [2*maxlen, head_dim] in the compute dtype (differentiable)
        self.maxlen = maxlen
        self._t = None

    def transposed(self):
        """[head_dim, 2*maxlen] constant copy, made once per forward and shared by every layer that uses this table: the
        attention backward's dQ += dQP . PE then runs as an NT-form GEMM on the LDS-DMA kernel (K-major B operand)."""
        if self._t is None:
            self._t = self.table.detach().t().contiguous()
        return self._t


def _mask_bytes(mask):
    """Key-padding mask as the uint8 [B, S] the kernels read: a bool tensor is reinterpreted in place (same 0/1 bytes), no
    conversion kernel per attention call."""
    if mask.dtype == torch.bool and mask.is_contiguous():
        return mask.view(torch.uint8)
    return mask.to(torch.uint8).contiguous()


def _is_causal_mask(attn_mask):
    return attn_mask is not None


class MultiheadAttention(nn.Module, IncrementalState):
    def __init__(self, embed_dim, num_heads, kdim=None, vdim=None, dropout=0.0, bias=True, add_bias_kv=False,
                 add_zero_attn=False, self_attention=False, encoder_decoder_attention=False, q_noise=0.0,
                 qn_block_size=8, has_relative_attention_bias=False):
        super().__init__()
        self.init_incremental_state()
        assert not add_bias_kv and not add_zero_attn and q_noise == 0.0
        self.embed_dim = embed_dim
        self.kdim = kdim if kdim is not None else embed_dim
        self.vdim = vdim if vdim is not None else embed_dim
        self.num_heads = num_heads
        self.dropout_p = dropout
        self.has_relative_attention_bias = has_relative_attention_bias
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim
        self.scaling = self.head_dim ** -0.5
        self.self_attention = self_attention
        self.encoder_decoder_attention = encoder_decoder_attention
        self.k_proj = nn.Linear(self.kdim, embed_dim, bias=bias)
        self.v_proj = nn.Linear(self.vdim, embed_dim, bias=bias)
        self.q_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.reset_parameters()

    def reset_parameters(self):
        g = 1 / 2 ** 0.5 if self.kdim == self.embed_dim and self.vdim == self.embed_dim else 1.0
        nn.init.xavier_uniform_(self.k_proj.weight, gain=g)
        nn.init.xavier_uniform_(self.v_proj.weight, gain=g)
        nn.init.xavier_uniform_(self.q_proj.weight, gain=g)
        nn.init.xavier_uniform_(self.out_proj.weight)
        if self.out_proj.bias is not None:
            nn.init.constant_(self.out_proj.bias, 0.0)

    # ---- batch-major row interface used by the layer mirrors (no layout copies) ----
    def forward_rows(self, x, B, T, *, kv=None, S=None, key_padding_mask=None, causal=False, position_bias=None,
                     residual=None, out_dropout=0.0, need_weights=False, kv_all=None, res_relay=None):
        """x [B*T, C] rows (query source); kv [B*S, C] rows (cross-attention source) or None for self-attention.
        Returns (out rows [B*T, C] = dropout(out_proj(attn)) + residual, probs [B,H,T,S] fp32 or None)."""
        H, hd = self.num_heads, self.head_dim
        p = self.dropout_p if self.training else 0.0
        kpm = None
        if key_padding_mask is not None:
            kpm = _mask_bytes(key_padding_mask)
        probs = None
        # post-LN blocks add the block input itself as the residual: relay its gradient into the first projection's dX GEMM
        relay = Fn.GradRelay() if (residual is x and x.requires_grad and torch.is_grad_enabled()) else None
        if kv is None and kv_all is None:
            qkv = Fn.linear(x, [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight],
                            [self.q_proj.bias, self.k_proj.bias, self.v_proj.bias], relay_in=relay)
            pe, maxrel, pe_t = None, 0, None
            if position_bias is not None and self.has_relative_attention_bias:
                pe, maxrel = position_bias.table, position_bias.maxlen
                if torch.is_grad_enabled() and qkv.requires_grad:
                    pe_t = position_bias.transposed()
            ctx = Fn.SelfAttentionFunction.apply(qkv, pe, kpm, (B, H, T, hd, maxrel, causal, p, pe_t))
        else:
            q = Fn.linear(x, self.q_proj.weight, self.q_proj.bias, relay_in=relay)
            if kv_all is not None:   # (tensor [B*S, ld], column offset of this layer's [K | V], functional.KVShare)
                kvt, koff, share = kv_all
                ctx, probs = Fn.CrossAttentionFunction.apply(q, kvt, kpm, (B, H, T, S, hd, p, need_weights, kvt.shape[1], koff, share))
            else:
                kvp = Fn.linear(kv, [self.k_proj.weight, self.v_proj.weight], [self.k_proj.bias, self.v_proj.bias])
                ctx, probs = Fn.CrossAttentionFunction.apply(q, kvp, kpm, (B, H, T, S, hd, p, need_weights))
        out = Fn.linear(ctx, self.out_proj.weight, self.out_proj.bias, residual=residual,
                        dropout_p=out_dropout if self.training else 0.0, relay_out=relay if relay is not None else res_relay)
        return out, probs

    # ---- incremental decoding (multihead_attention.py:269-307 of the reference: saved_state prev_key / prev_value) ----
    @torch.no_grad()
    def forward_rows_cached(self, x, B, cache, *, kv=None, S=None, key_padding_mask=None, residual=None, need_weights=False):
        """One new position per sequence: x [B, C] rows.  Self-attention appends this step's (k, v) to cache["kv"]
        ([B, t, 2C], projected) and attends over all t cached keys; cross-attention projects the encoder rows once
        (cache["kvp"], [B*S, 2C]) and re-uses them every step.  Returns (out rows [B, C], probs [B,H,1,S] or None)."""
        H, hd, C = self.num_heads, self.head_dim, self.embed_dim
        kpm = _mask_bytes(key_padding_mask) if key_padding_mask is not None else None
        if kv is None:
            qkv = Fn.linear(x, [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight],
                            [self.q_proj.bias, self.k_proj.bias, self.v_proj.bias])          # [B, 3C]
            q = qkv[:, :C].contiguous()
            step = qkv[:, C:].reshape(B, 1, 2 * C)
            cache["kv"] = step.contiguous() if "kv" not in cache else torch.cat([cache["kv"], step], dim=1)
            t = cache["kv"].shape[1]
            ctx, probs = Fn.CrossAttentionFunction.apply(q, cache["kv"].view(B * t, 2 * C), kpm, (B, H, 1, t, hd, 0.0, need_weights))
        else:
            if "kvp" not in cache:
                cache["kvp"] = Fn.linear(kv, [self.k_proj.weight, self.v_proj.weight], [self.k_proj.bias, self.v_proj.bias])
            q = Fn.linear(x, self.q_proj.weight, self.q_proj.bias)
            ctx, probs = Fn.CrossAttentionFunction.apply(q, cache["kvp"], kpm, (B, H, 1, S, hd, 0.0, need_weights))
        out = Fn.linear(ctx, self.out_proj.weight, self.out_proj.bias, residual=residual)
        return out, probs

    # ---- reference-compatible Time x Batch x Channel interface ----
    def forward(self, query, key, value, key_padding_mask=None, incremental_state=None, need_weights=True,
                static_kv=False, attn_mask=None, before_softmax=False, need_head_weights=False, position_bias=None):
        assert not before_softmax
        if need_head_weights:
            need_weights = True
        T, B, C = query.shape
        xq = Fn.as_compute(query.transpose(0, 1).contiguous()).view(B * T, C)
        if self.self_attention or key is query:
            out, _ = self.forward_rows(xq, B, T, key_padding_mask=key_padding_mask, causal=_is_causal_mask(attn_mask),
                                       position_bias=position_bias)
            probs = None
        else:
            S = key.shape[0]
            xk = Fn.as_compute(key.transpose(0, 1).contiguous()).view(B * S, C)
            out, probs = self.forward_rows(xq, B, T, kv=xk, S=S, key_padding_mask=key_padding_mask, need_weights=need_weights)
        attn = out.view(B, T, C).transpose(0, 1)
        weights = None
        if probs is not None and need_weights:
            weights = probs.transpose(0, 1)  # [H,B,T,S]
            if not need_head_weights:
                weights = weights.mean(dim=0)
        return attn, weights

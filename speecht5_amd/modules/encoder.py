"""TransformerEncoder mirror of SpeechT5/speecht5/models/modules/encoder.py:61-380."""
import contextlib

import numpy as np
import torch
import torch.nn as nn

from .. import functional as Fn
from ..fairseq_compat import FairseqEncoder
from .common import LayerNorm
from .multihead_attention import RelPosKeys
from .transformer_layer import TransformerSentenceEncoderLayer


def Linear(in_features, out_features, bias=True):
    m = nn.Linear(in_features, out_features, bias)
    nn.init.xavier_uniform_(m.weight)
    if bias:
        nn.init.constant_(m.bias, 0.0)
    return m


class RelativePositionalEncoding(nn.Module):
    """encoder.py:40-59.  The reference gathers pe_k[clip(i-j)+maxlen] into a [T,T,hd] tensor; here the
    module hands out a RelPosKeys handle and the gather happens inside the attention kernels."""

    def __init__(self, d_model, maxlen=1000, embed_v=False):
        super().__init__()
        assert not embed_v
        self.d_model = d_model
        self.maxlen = maxlen
        self.pe_k = nn.Embedding(2 * maxlen, d_model)
        self.embed_v = embed_v
        self.spans_cuts = False

    def forward(self, pos_seq=None):
        return RelPosKeys(Fn.as_compute(self.pe_k.weight), self.maxlen), None

    def for_layer(self, keys):
        """The keys as the NEXT layer of the stack reads them (call after that layer's own Fn.layer_boundary): the same table,
        behind the gradient-exchange wrapper's "shared" boundary (ddp._boundary: one leaf per region of a phased backward)."""
        if keys is None or not self.spans_cuts:
            return keys
        t = Fn.layer_boundary(keys.table, self, "shared")
        if t is keys.table:            # no cut in force: the same handle (and its one transposed copy) serves every layer
            return keys
        k = RelPosKeys(t, keys.maxlen)
        k._t = keys.transposed()       # (a constant of the forward: the same values behind every region's leaf)
        return k


class TransformerEncoder(FairseqEncoder):
    def __init__(self, args, tgt_dict=None, embed_tokens=None):
        self.args = args
        super().__init__(None)
        self.register_buffer("version", torch.Tensor([3]))
        self.dropout = args.dropout
        self.encoder_layerdrop = args.encoder_layerdrop
        self.freeze_encoder_updates = args.freeze_encoder_updates
        self.no_freeze_encoder_layer = eval(args.no_freeze_encoder_layer) if args.no_freeze_encoder_layer is not None else None
        self.num_updates = 0
        assert args.use_sent_enc_layer, "only the TransformerSentenceEncoderLayer path exists in SpeechT5 recipes"
        self.layers = nn.ModuleList([self.build_encoder_layer(args) for _ in range(args.encoder_layers)])
        self.num_layers = len(self.layers)
        self.use_sent_enc_layer = args.use_sent_enc_layer
        self.unb_enc_layer = getattr(args, "unb_enc_layer", -1)
        self.layer_norm_first = args.layer_norm_first
        self.layer_norm = LayerNorm(args.encoder_embed_dim, eps=args.layer_norm_eps)
        if args.share_ctc_embed and embed_tokens is not None:
            self.proj = nn.Linear(embed_tokens.weight.shape[1], embed_tokens.weight.shape[0], bias=False)
            self.proj.weight = embed_tokens.weight
        elif tgt_dict is not None:
            self.proj = Linear(args.encoder_embed_dim, len(tgt_dict))
        else:
            self.proj = None
        if args.relative_position_embedding:
            self.pos_emb = RelativePositionalEncoding(args.encoder_embed_dim // args.encoder_attention_heads,
                                                      args.encoder_max_relative_position)
            self.pos_emb.spans_cuts = True

    def build_encoder_layer(self, args):
        return TransformerSentenceEncoderLayer(
            embedding_dim=args.encoder_embed_dim, ffn_embedding_dim=args.encoder_ffn_embed_dim,
            num_attention_heads=args.encoder_attention_heads, dropout=args.dropout,
            attention_dropout=args.attention_dropout, activation_dropout=args.activation_dropout,
            activation_fn=args.activation_fn, layer_norm_first=args.layer_norm_first,
            has_relative_attention_bias=args.relative_position_embedding)

    def forward(self, encoder_in, encoder_padding_mask, return_all_hiddens=False, tgt_layer=None):
        ft = True if self.no_freeze_encoder_layer is not None else self.freeze_encoder_updates <= self.num_updates
        with torch.no_grad() if not ft else contextlib.ExitStack():
            encoder_out = self.forward_scriptable(encoder_in, encoder_padding_mask, return_all_hiddens, tgt_layer=tgt_layer)
        x_for_ctc = None
        if self.proj is not None:  # CTC head on dropout(encoder_out) (:173-179)
            x = encoder_out["encoder_out"][0]  # T x B x C view of batch-major storage
            T, B, C = x.shape
            rows = Fn.dropout(x.transpose(0, 1).contiguous(), self.dropout, self.training)
            logits = Fn.linear(rows.view(B * T, C), self.proj.weight, self.proj.bias)
            x_for_ctc = Fn.as_float(logits.contiguous()).view(B, T, -1).transpose(0, 1)
        encoder_out["encoder_out_for_ctc"] = [x_for_ctc]
        return encoder_out

    def forward_scriptable(self, encoder_in, encoder_padding_mask, return_all_hiddens=False, tgt_layer=None):
        ft = self.freeze_encoder_updates <= self.num_updates if self.no_freeze_encoder_layer is not None else True
        B, T, C = encoder_in.shape
        with torch.no_grad() if not ft else contextlib.ExitStack():
            x = Fn.as_compute(encoder_in.contiguous())
            if not self.layer_norm_first:
                x = self.layer_norm(x)
            x = Fn.dropout(x, self.dropout, self.training)
            x = x.view(B * T, C)
            encoder_states = []
            if return_all_hiddens:
                encoder_states.append(x.view(B, T, C).transpose(0, 1))
            pos_k = self.pos_emb()[0] if self.args.relative_position_embedding else None
        r, d = None, None
        # LayerDrop (encoder.py:251-257 of the reference: one numpy draw per layer, the layer runs when the draw exceeds
        # --encoder-layerdrop).  In a step recorded for graph replay the decision cannot be host control flow: the draws of the
        # whole stack are made at once (same positions of the numpy stream as the per-layer draws), staged to the device as
        # keep flags, every layer runs and its output is selected on the device (Fn.layerdrop_select).
        keep_dev = None
        if self.training and self.encoder_layerdrop > 0 and Fn.layerdrop_on_device(x):
            n_draw = len(self.layers) if tgt_layer is None else min(len(self.layers), tgt_layer + 1)
            p_drop = float(self.encoder_layerdrop)
            keep_dev = Fn.stage_host(lambda: torch.from_numpy((np.random.random(n_draw) > p_drop).astype(np.float32)), x.device)
        for i, layer in enumerate(self.layers):
            if keep_dev is None:
                dropout_probability = Fn.host_draw(np.random.random)   # (drawn per layer even at LayerDrop 0, as the reference does)
            frozen = (not ft) and i not in self.no_freeze_encoder_layer
            with torch.no_grad() if frozen else contextlib.ExitStack():
                if keep_dev is not None:
                    x = Fn.layer_boundary(x, layer)      # (the select's skip operand is the tensor BEHIND the layer's boundary)
                    pb = self.pos_emb.for_layer(pos_k) if pos_k is not None else None
                    if i != self.unb_enc_layer and not frozen and torch.is_grad_enabled() and Fn.LAYERDROP_GATE and layer.gate_ok():
                        # post-LN layer: the select rides on its last LayerNorm and its gradient on the layer's input (Fn.LayerDropGate)
                        gate, xg = Fn.layerdrop_gate(x, keep_dev[i:i + 1])
                        x = layer.forward_rows(xg, B, T, padding_mask=encoder_padding_mask, pos_bias=pb, gate=gate)
                        assert gate.used
                    else:
                        y = layer.forward_rows(x, B, T, padding_mask=encoder_padding_mask, pos_bias=pb)
                        x = y if i == self.unb_enc_layer else Fn.layerdrop_select(x, y, keep_dev[i:i + 1])
                elif not self.training or (dropout_probability > self.encoder_layerdrop) or i == self.unb_enc_layer:
                    x = Fn.layer_boundary(x, layer)
                    x = layer.forward_rows(x, B, T, padding_mask=encoder_padding_mask, pos_bias=self.pos_emb.for_layer(pos_k) if pos_k is not None else None)
                else:   # LayerDrop: the layer's (zero) gradient bucket still reports ready at this point of backward
                    x = Fn.layer_boundary(x, layer)
                if i == self.unb_enc_layer:
                    d = x.view(B, T, C).transpose(0, 1)
                if i == tgt_layer:
                    r = x
                    break
                if return_all_hiddens:
                    encoder_states.append(x.view(B, T, C).transpose(0, 1))
        with torch.no_grad() if not ft else contextlib.ExitStack():
            if self.layer_norm_first:
                x = self.layer_norm(x)
            if r is not None:
                x = r
        x = Fn.layer_boundary(x, self, "out")   # everything that consumes the encoder output is behind this point
        return {
            "encoder_out": [x.view(B, T, C).transpose(0, 1)],  # T x B x C (view of batch-major rows)
            "encoder_padding_mask": [encoder_padding_mask],
            "encoder_states": encoder_states,
            "src_tokens": [],
            "decoder_input": [d],
        }

    def reorder_encoder_out(self, encoder_out, new_order):
        out = {}
        out["encoder_out"] = [x.index_select(1, new_order) for x in encoder_out["encoder_out"]]
        out["encoder_padding_mask"] = [x.index_select(0, new_order) for x in encoder_out["encoder_padding_mask"]]
        out["encoder_out_for_ctc"] = [x.index_select(1, new_order) if x is not None else None
                                      for x in encoder_out.get("encoder_out_for_ctc", [])]
        out["encoder_states"] = [s.index_select(1, new_order) for s in encoder_out.get("encoder_states", [])]
        out["src_tokens"] = []
        out["decoder_input"] = [x.index_select(1, new_order) if x is not None else None
                                for x in encoder_out.get("decoder_input", [])]
        return out

    def set_num_updates(self, num_updates):
        self.num_updates = num_updates

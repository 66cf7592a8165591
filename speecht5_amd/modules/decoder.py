"""TransformerDecoder mirror of SpeechT5/speecht5/models/modules/decoder.py:33-324."""
import torch
import torch.nn as nn

from .. import functional as Fn
from ..fairseq_compat import FairseqIncrementalDecoder
from .common import LayerNorm
from .encoder import RelativePositionalEncoding
from .transformer_layer import TransformerDecoderLayer


def state_B(state, rows):
    """Batch size of a cached [B*S, C] row tensor (S is recorded when the encoder rows are first cached)."""
    return rows.shape[0] // state["S"]


class TransformerDecoder(FairseqIncrementalDecoder):
    def __init__(self, args, no_encoder_attn=False):
        self.args = args
        super().__init__(None)
        self.register_buffer("version", torch.Tensor([3]))
        self.dropout = args.dropout
        self.decoder_layerdrop = args.decoder_layerdrop
        self.cross_self_attention = getattr(args, "cross_self_attention", False)
        assert not self.cross_self_attention
        self.layers = nn.ModuleList([self.build_decoder_layer(args, no_encoder_attn) for _ in range(args.decoder_layers)])
        self.num_layers = len(self.layers)
        self.materialise_alignment = True   # criterions.alignment_weights(): whether a training forward returns `attn`
        if args.decoder_normalize_before and not getattr(args, "no_decoder_final_norm", False):
            self.layer_norm = LayerNorm(args.decoder_embed_dim, eps=args.layer_norm_eps)
        else:
            self.layer_norm = None
        if args.relative_position_embedding:  # built but unused by the layers (decoder.py:83-84, transformer_layer.py:241)
            self.pos_emb = RelativePositionalEncoding(args.encoder_embed_dim // args.encoder_attention_heads,
                                                      args.decoder_max_relative_position)

    def build_decoder_layer(self, args, no_encoder_attn=False):
        return TransformerDecoderLayer(args, no_encoder_attn=no_encoder_attn,
                                       has_relative_attention_bias=args.relative_position_embedding)

    def forward(self, prev_output_tokens, tgt_mask, encoder_out=None, incremental_state=None, full_context_alignment=False,
                alignment_layer=None, alignment_heads=None, src_lengths=None, return_all_hiddens=False):
        return self.extract_features(prev_output_tokens, tgt_mask, encoder_out=encoder_out, incremental_state=incremental_state,
                                     full_context_alignment=full_context_alignment, alignment_layer=alignment_layer,
                                     alignment_heads=alignment_heads)

    def extract_features(self, prev_output_tokens, tgt_mask, encoder_out, incremental_state=None, full_context_alignment=False,
                         alignment_layer=None, alignment_heads=None):
        return self.extract_features_scriptable(prev_output_tokens, tgt_mask, encoder_out, incremental_state,
                                                full_context_alignment, alignment_layer, alignment_heads)

    def extract_features_scriptable(self, prev_output_tokens, tgt_mask, encoder_out, incremental_state=None,
                                    full_context_alignment=False, alignment_layer=None, alignment_heads=None):
        """prev_output_tokens: decoder pre-net output [B,T,C].  With `incremental_state` (inference) the pre-nets hand over
        the NEWEST position only (T = 1, as in the reference) and every layer keeps its projected keys/values in the state
        dict (multihead_attention.py:269-307 of the reference), so a step costs O(prefix) instead of O(prefix^2)."""
        B, T, C = prev_output_tokens.shape
        if alignment_layer is None:
            alignment_layer = self.num_layers - 1
        if incremental_state is not None and not self.training:
            return self._extract_features_incremental(prev_output_tokens, tgt_mask, encoder_out, incremental_state,
                                                      alignment_layer, alignment_heads)
        enc_rows, S, enc_pad = None, None, None
        if encoder_out is not None and len(encoder_out["encoder_out"]) > 0:
            enc = encoder_out["encoder_out"][0]  # T x B x C
            assert enc.size(1) == B, f"Expected enc.shape == (t, {B}, c) got {enc.shape}"
            S = enc.size(0)
            enc_rows = Fn.as_compute(enc.transpose(0, 1).contiguous()).view(B * S, -1)
        if encoder_out is not None and len(encoder_out["encoder_padding_mask"]) > 0:
            enc_pad = encoder_out["encoder_padding_mask"][0]
        x = Fn.as_compute(prev_output_tokens.contiguous()).view(B * T, C)
        causal = not full_context_alignment
        # keys / values of every layer's cross-attention in ONE projection of the encoder output (same input for all layers;
        # the reference runs k_proj / v_proj per layer, transformer_layer.py:352-366): [B*S, L*2C], layer l = columns l*2C...
        kv_all, share = None, None
        cross = [l.encoder_attn for l in self.layers]
        if enc_rows is not None and all(a is not None for a in cross) and len(cross) > 1:
            ws = [w for a in cross for w in (a.k_proj.weight, a.v_proj.weight)]
            bs = [b for a in cross for b in (a.k_proj.bias, a.v_proj.bias)]
            if all(b is not None for b in bs) and all(w.shape == ws[0].shape for w in ws):
                kv_all = Fn.linear(enc_rows, ws, bs)
                # (a layer can only be SKIPPED -- and leave its slice of the shared gradient buffer unwritten -- when LayerDrop is
                # host control flow; in the recorded / replayed form every layer runs and writes its slice, no zero fill needed)
                share = Fn.KVShare(len(cross), 2 * C, self.training and self.decoder_layerdrop > 0 and not Fn.layerdrop_on_device(x))
        attn_list, attn = [], None
        inner_states = [x.view(B, T, C).transpose(0, 1)]
        # LayerDrop (decoder.py:64-67 of the reference: LayerDropModuleList draws torch.empty(L).uniform_() once per pass over
        # the layers, a layer runs when its draw exceeds --decoder-layerdrop).  Recorded for graph replay: keep flags staged to
        # the device, every layer runs, outputs selected on the device (see the encoder).
        drops, keep_dev = None, None
        if self.decoder_layerdrop > 0:
            L_, p_drop = len(self.layers), float(self.decoder_layerdrop)
            if self.training and Fn.layerdrop_on_device(x):
                keep_dev = Fn.stage_host(lambda: (torch.empty(L_).uniform_() > p_drop).float(), x.device)
            else:
                drops = Fn.host_draw(lambda: torch.empty(L_).uniform_()) <= p_drop
        for idx, layer in enumerate(self.layers):
            if self.training and drops is not None and bool(drops[idx]):
                x = Fn.layer_boundary(x, layer)   # skipped layer: its (zero) gradient bucket still reports ready here
                continue
            want = bool(idx == alignment_layer or alignment_layer == -1) and (self.materialise_alignment or not self.training)
            if keep_dev is not None:
                x = Fn.layer_boundary(x, layer)      # (the select's skip operand is the tensor BEHIND the layer's boundary)
            kva = (kv_all, idx * 2 * C, share) if kv_all is not None else None
            if keep_dev is not None and torch.is_grad_enabled() and Fn.LAYERDROP_GATE and layer.gate_ok():
                # post-LN layer: the select rides on its last LayerNorm and its gradient on the layer's input (Fn.LayerDropGate)
                gate, xg = Fn.layerdrop_gate(x, keep_dev[idx:idx + 1])
                x, layer_attn = layer.forward_rows(xg, B, T, enc_rows, S, enc_pad, tgt_mask, causal, want, kv_all=kva, gate=gate)
                assert gate.used
            else:
                y, layer_attn = layer.forward_rows(x, B, T, enc_rows, S, enc_pad, tgt_mask, causal, want, kv_all=kva)
                x = y if keep_dev is None else Fn.layerdrop_select(x, y, keep_dev[idx:idx + 1])
            inner_states.append(x.view(B, T, C).transpose(0, 1))
            if layer_attn is not None and want:
                attn = layer_attn.transpose(0, 1)       # [H,B,T,S] as the reference's per-head weights
                attn_list.append(layer_attn)             # [B,H,T,S] (= attn.transpose(0,1))
        if attn is not None and len(attn_list) == 1:
            if alignment_heads is not None:
                attn = attn[:alignment_heads]
            attn = attn.mean(dim=0)
        if self.layer_norm is not None:
            x = self.layer_norm(x)
        x = Fn.layer_boundary(x, self, "out")   # the post-nets' gradients are complete when this node's backward runs
        x = x.view(B, T, C)
        return x, {"attn": [attn if len(attn_list) <= 1 else attn_list], "inner_states": inner_states}

    _CACHE_KEY = "st5_decoder_kv_cache"

    @torch.no_grad()
    def _extract_features_incremental(self, prev_output_tokens, tgt_mask, encoder_out, incremental_state, alignment_layer,
                                      alignment_heads):
        B, T, C = prev_output_tokens.shape
        state = incremental_state.setdefault(self._CACHE_KEY, {"layers": [dict() for _ in self.layers]})
        enc_rows, S, enc_pad = None, None, None
        if encoder_out is not None and len(encoder_out["encoder_out"]) > 0:
            enc = encoder_out["encoder_out"][0]  # T x B x C
            assert enc.size(1) == B, f"Expected enc.shape == (t, {B}, c) got {enc.shape}"
            S = enc.size(0)
            if "enc_rows" not in state:
                state["enc_rows"] = Fn.as_compute(enc.transpose(0, 1).contiguous()).view(B * S, -1)
                state["S"] = S
            enc_rows = state["enc_rows"]
        if encoder_out is not None and len(encoder_out["encoder_padding_mask"]) > 0:
            enc_pad = encoder_out["encoder_padding_mask"][0]
        x_all = Fn.as_compute(prev_output_tokens.contiguous())
        attn_list, attn, inner_states = [], None, []
        x = None
        for t in range(T):   # T == 1 in the generator loop; a longer first chunk is fed position by position
            x = x_all[:, t].contiguous()
            inner_states = [x.view(B, 1, C).transpose(0, 1)]
            attn_list, attn = [], None
            for idx, layer in enumerate(self.layers):
                want = bool(idx == alignment_layer or alignment_layer == -1)
                x, layer_attn = layer.forward_rows_cached(x, B, state["layers"][idx], enc_rows, S, enc_pad, tgt_mask, want)
                inner_states.append(x.view(B, 1, C).transpose(0, 1))
                if layer_attn is not None and want:
                    attn = layer_attn.transpose(0, 1)       # [H,B,1,S]
                    attn_list.append(layer_attn)             # [B,H,1,S]
        if attn is not None and len(attn_list) == 1:
            if alignment_heads is not None:
                attn = attn[:alignment_heads]
            attn = attn.mean(dim=0)
        if self.layer_norm is not None:
            x = self.layer_norm(x)
        return x.view(B, 1, C), {"attn": [attn if len(attn_list) <= 1 else attn_list], "inner_states": inner_states}

    def reorder_incremental_state_scripting(self, incremental_state, new_order):
        """Beam search re-ordering (decoder.py:301-314 of the reference): every cached tensor is batch-major."""
        state = incremental_state.get(self._CACHE_KEY) if incremental_state is not None else None
        if state is None:
            return incremental_state
        for lc in state["layers"]:
            if "kv" in lc.get("self", {}):
                lc["self"]["kv"] = lc["self"]["kv"].index_select(0, new_order)
            if "kvp" in lc.get("cross", {}):
                kvp = lc["cross"]["kvp"]
                lc["cross"]["kvp"] = kvp.view(state_B(state, kvp), -1, kvp.shape[1]).index_select(0, new_order).reshape(-1, kvp.shape[1])
        if "enc_rows" in state:
            er = state["enc_rows"]
            state["enc_rows"] = er.view(state_B(state, er), -1, er.shape[1]).index_select(0, new_order).reshape(-1, er.shape[1])
        return incremental_state

    def set_num_updates(self, num_updates):
        def _apply(m):
            if hasattr(m, "set_num_updates") and m != self:
                m.set_num_updates(num_updates)
        self.apply(_apply)

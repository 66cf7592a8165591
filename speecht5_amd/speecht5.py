"""T5TransformerModel mirror of SpeechT5/speecht5/models/speecht5.py:47-1447: same registration names
(`t5_transformer` + archs), constructor/builders, forward routing, inference entry points and state-dict
keys, with every module running on the gfx950 HIP kernels (speecht5_amd/csrc)."""
import logging
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import functional as Fn
from .fairseq_compat import FairseqEncoderDecoderModel, register_model, register_model_architecture
from .modules import (GumbelVectorQuantizer, SpeechDecoderPostnet, SpeechDecoderPrenet, SpeechEncoderPostnet,
                      SpeechEncoderPrenet, TextDecoderPostnet, TextDecoderPrenet, TextEncoderPrenet, TransformerDecoder,
                      TransformerEncoder)

logger = logging.getLogger(__name__)

DEFAULT_MAX_TEXT_POSITIONS = 450
DEFAULT_MAX_SPEECH_POSITIONS = 4000


def Embedding(num_embeddings, embedding_dim, padding_idx):
    m = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)
    nn.init.normal_(m.weight, mean=0, std=embedding_dim ** -0.5)
    nn.init.constant_(m.weight[padding_idx], 0)
    return m


def init_bert_params(module):
    """fairseq init_bert_params: N(0, 0.02) for Linear / Embedding / MHA q,k,v weights; zero biases and pad row."""
    from .modules import MultiheadAttention

    def normal_(data):
        data.copy_(data.cpu().normal_(mean=0.0, std=0.02).to(data.device))

    if isinstance(module, nn.Linear):
        normal_(module.weight.data)
        if module.bias is not None:
            module.bias.data.zero_()
    if isinstance(module, nn.Embedding):
        normal_(module.weight.data)
        if module.padding_idx is not None:
            module.weight.data[module.padding_idx].zero_()
    if isinstance(module, MultiheadAttention):
        normal_(module.q_proj.weight.data)
        normal_(module.k_proj.weight.data)
        normal_(module.v_proj.weight.data)


@register_model("t5_transformer")
class T5TransformerModel(FairseqEncoderDecoderModel):
    def __init__(self, args, encoder, decoder, text_encoder_prenet, speech_encoder_prenet, text_decoder_prenet,
                 speech_decoder_prenet, text_decoder_postnet, speech_decoder_postnet, speaker_decoder_postnet,
                 speech_encoder_postnet):
        super().__init__(encoder, decoder)
        self.encoder = encoder
        self.decoder = decoder
        self.text_encoder_prenet = text_encoder_prenet
        self.speech_encoder_prenet = speech_encoder_prenet
        self.text_decoder_prenet = text_decoder_prenet
        self.speech_decoder_prenet = speech_decoder_prenet
        self.text_decoder_postnet = text_decoder_postnet
        self.speech_decoder_postnet = speech_decoder_postnet
        self.speaker_decoder_postnet = speaker_decoder_postnet
        self.hubert_layer = speech_encoder_postnet
        self.reduction_factor = args.reduction_factor
        self.spk_embed_dim = args.spk_embed_dim
        self.spk_embed_integration_type = args.spk_embed_integration_type
        if self.spk_embed_dim is not None and self.spk_embed_integration_type != "pre":
            raise NotImplementedError("speaker-embedding integration other than 'pre' is not used by the recipes")
        self.use_codebook = args.use_codebook
        self.codebook_prob = getattr(args, "codebook_prob", 0.5)
        if self.use_codebook:
            vq_dim = args.latent_dim if args.latent_dim > 0 else args.encoder_embed_dim
            self.quantizer = GumbelVectorQuantizer(dim=args.encoder_embed_dim, num_vars=args.latent_vars, temp=args.latent_temp,
                                                   groups=args.latent_groups, combine_groups=False, vq_dim=vq_dim, time_first=True,
                                                   weight_proj_depth=args.quantizer_depth, weight_proj_factor=args.quantizer_factor)
        self.num_updates = 0
        if args.bert_init:
            self.apply(init_bert_params)
        self.args = args

    @staticmethod
    def add_args(parser):
        """Model options, speecht5.py:117-614 of the reference (same names / dests / defaults: speecht5_amd/cli.py)."""
        from . import cli
        cli.declare(parser, cli.MODEL_OPTIONS)

    # ---- builders (speecht5.py:616-729) ----
    @classmethod
    def build_encoder(cls, args, dictionary=None, embed_tokens=None):
        return TransformerEncoder(args, dictionary, embed_tokens)

    @classmethod
    def build_decoder(cls, args):
        return TransformerDecoder(args)

    @classmethod
    def build_text_encoder_prenet(cls, embed_tokens, args):
        return TextEncoderPrenet(embed_tokens, args)

    @classmethod
    def build_speech_encoder_prenet(cls, args):
        return SpeechEncoderPrenet(args)

    @classmethod
    def build_text_decoder_prenet(cls, embed_tokens, args):
        return TextDecoderPrenet(embed_tokens, args)

    @classmethod
    def build_speech_decoder_prenet(cls, odim, args):
        return SpeechDecoderPrenet(odim, args)

    @classmethod
    def build_text_decoder_postnet(cls, embed_tokens, dictionary, args):
        return TextDecoderPostnet(embed_tokens, dictionary, args)

    @classmethod
    def build_speech_decoder_postnet(cls, odim, args):
        return SpeechDecoderPostnet(odim, args)

    @classmethod
    def build_speech_encoder_postnet(cls, dictionaries, args):
        return SpeechEncoderPostnet(dictionaries, args)

    @classmethod
    def build_model(cls, args, task):
        base_architecture(args)

        def build_embedding(dictionary, embed_dim):
            return Embedding(len(dictionary), embed_dim, dictionary.pad())

        text_decoder_embed_tokens = build_embedding(task.dicts["text"], args.decoder_embed_dim)
        if args.share_input_output_embed:
            text_encoder_embed_tokens = text_decoder_embed_tokens
        else:
            text_encoder_embed_tokens = build_embedding(task.dicts["text"], args.encoder_embed_dim)
        speech_odim = args.speech_odim
        if "text" in task.dicts:
            encoder = cls.build_encoder(args, task.dicts["text"], text_encoder_embed_tokens)
        else:
            encoder = cls.build_encoder(args)
        decoder = cls.build_decoder(args)
        text_encoder_prenet = cls.build_text_encoder_prenet(text_encoder_embed_tokens, args)
        speech_encoder_prenet = cls.build_speech_encoder_prenet(args)
        text_decoder_prenet = cls.build_text_decoder_prenet(text_decoder_embed_tokens, args)
        speech_decoder_prenet = cls.build_speech_decoder_prenet(speech_odim, args)
        text_decoder_postnet = cls.build_text_decoder_postnet(text_decoder_embed_tokens, task.dicts["text"], args)
        speech_decoder_postnet = cls.build_speech_decoder_postnet(speech_odim, args)
        if getattr(task, "t5_task", None) == "s2c":
            raise NotImplementedError("speaker identification (s2c) is outside the SpeechT5 hot-path scope (SURVEY.md 2.2)")
        speech_encoder_postnet = cls.build_speech_encoder_postnet(task.dicts["hubert"], args) if "hubert" in task.dicts else None
        return cls(args, encoder, decoder, text_encoder_prenet, speech_encoder_prenet, text_decoder_prenet,
                   speech_decoder_prenet, text_decoder_postnet, speech_decoder_postnet, None, speech_encoder_postnet)

    # ---- criterion-facing helpers (speecht5.py:731-784) ----
    def get_normalized_probs(self, net_output, log_probs, sample=None):
        logits = net_output[0]
        lprobs = F.log_softmax(logits.float(), dim=-1) if log_probs else F.softmax(logits.float(), dim=-1)
        lprobs.batch_first = True
        return lprobs

    def get_normalized_probs_for_ctc(self, net_output, log_probs):
        logits = net_output["encoder_out_for_ctc"][0]
        return F.log_softmax(logits.float(), dim=-1) if log_probs else F.softmax(logits.float(), dim=-1)

    def get_logits(self, net_output, is_masked=True):
        logits_list = net_output["logit_m_list"] if is_masked else net_output["logit_u_list"]
        return [x.float() for x in logits_list if x is not None]

    def get_targets(self, sample, net_output, is_masked=True):
        if "logit_m_list" in net_output:
            sel = net_output.get("sel_m" if is_masked else "sel_u")
            if sel is not None:
                # fixed-shape form (recorded / replayed steps): logits of ALL frames; the frames outside the (un)masked set
                # get the ignored target -1, the others class 0 as in the reference
                return [(sel.long() - 1) for _ in self.get_logits(net_output, is_masked)]
            return [x.new_zeros(x.size(0), dtype=torch.long) for x in self.get_logits(net_output, is_masked)]
        return sample["target"]

    def get_target_count(self, net_output, is_masked=True):
        """Number of frames the (un)masked prediction loss runs over: the reference's `targ_list[0].numel()`
        (speech_pretrain_criterion.py:117,127) -- a Python int, or a device scalar in the fixed-shape form."""
        sel = net_output.get("sel_m" if is_masked else "sel_u")
        if sel is not None:
            return sel.sum().float()
        lg = self.get_logits(net_output, is_masked)
        return lg[0].size(0) if lg else 0

    def get_extra_losses(self, net_output):
        extra_losses, names = [], []
        if "features_pen" in net_output:
            extra_losses.append(net_output["features_pen"])
            names.append("features_pen")
        if "prob_perplexity" in net_output:
            extra_losses.append((net_output["num_vars"] - net_output["prob_perplexity"]) / net_output["num_vars"])
            names.append("prob_perplexity")
        return extra_losses, names

    def max_positions(self):
        return None

    def max_decoder_positions(self):
        return self.args.max_text_positions

    # ---- forward (speecht5.py:786-963) ----
    def forward(self, source=None, src_tokens=None, src_lengths=None, prev_output_tokens=None, tgt_lengths=None, spkembs=None,
                target_list=None, task_name=None, padding_mask=None, only_hubert=False, only_ctc=False, feature_only=False,
                tgt_enc_layer=None, mask=True):
        assert source is not None or src_tokens is not None
        input_type = "text" if (source is None and padding_mask is None and not feature_only) else "speech"
        if prev_output_tokens is not None and len(prev_output_tokens.size()) == 2:
            output_type = "text"
            codebook_out = {}
        else:
            output_type = "speech"
        if task_name is not None and task_name == "s2c":
            raise NotImplementedError("s2c")

        # Encoder pre-net
        if input_type == "text":
            encoder_input, encoder_padding_mask = self.text_encoder_prenet(src_tokens)
        elif target_list is not None:
            encoder_input, encoder_padding_mask = self.speech_encoder_prenet(
                source, require_feat_pen=True, target_list=target_list, padding_mask=padding_mask, mask=mask)
            encoder_input, features_pen, mask_indices, target_list = encoder_input
        else:
            encoder_input, encoder_padding_mask = self.speech_encoder_prenet(source, padding_mask=padding_mask, mask=self.training)

        encoder_output = self.encoder(encoder_input, encoder_padding_mask, tgt_layer=tgt_enc_layer)

        if task_name is not None and task_name == "speech_pretrain" and feature_only:
            return encoder_output["encoder_out"][0].transpose(0, 1)

        if target_list is not None:
            hubert_results = self.hubert_layer(encoder_output["encoder_out"][0].transpose(0, 1), encoder_padding_mask,
                                               mask_indices, target_list)
            hubert_results["features_pen"] = features_pen

        if "decoder_input" in encoder_output and encoder_output["decoder_input"][0] is not None:
            encoder_output["encoder_out"] = encoder_output["decoder_input"]

        if self.use_codebook:
            enc_btc = encoder_output["encoder_out"][0].transpose(0, 1)
            tlen = enc_btc.size(1)
            n_mix = int(tlen * self.codebook_prob)

            def draw_mix():   # built on the host: no device-side index_put / sync
                w = torch.zeros(tlen)
                w[torch.randperm(tlen)[:n_mix]] = 1.0
                return w
            q_w = Fn.stage_host(draw_mix, enc_btc.device)
            # quantised codes, mixed time-wise with the encoder states (:870-877) inside the quantizer's kernel
            q = self.quantizer(enc_btc.contiguous(), mix_w=q_w)
            encoder_output["encoder_out"][0] = q["x"].transpose(0, 1)
            tgt = hubert_results if output_type == "speech" else codebook_out
            tgt["prob_perplexity"] = q["prob_perplexity"]
            tgt["code_perplexity"] = q["code_perplexity"]
            tgt["num_vars"] = q["num_vars"]
            tgt["temp"] = q["temp"]

        if only_hubert and target_list is not None:
            return hubert_results, None
        if only_ctc and task_name is not None and task_name == "s2t":
            return None, encoder_output
        elif not self.training and prev_output_tokens is None and task_name == "s2t" and task_name is not None:
            return encoder_output

        # Decoder pre-net
        if output_type == "text":
            prev_output_tokens, tgt_mask, _ = self.text_decoder_prenet(prev_output_tokens)
        else:
            prev_output_tokens, tgt_mask = self.speech_decoder_prenet(prev_output_tokens, tgt_lengths, spkembs)
        if task_name is not None and task_name == "s2s" and getattr(self.args, "se_decoder_input", "previous_target") == "source":
            prev_output_tokens, tgt_mask = self.speech_decoder_prenet(src_tokens, src_lengths)

        decoder_output, extra = self.decoder(
            prev_output_tokens, tgt_mask, encoder_output,
            full_context_alignment=getattr(self.args, "decoder_full_context_alignment", False),
            alignment_layer=(-1 if target_list is None and output_type == "speech" else None))

        if task_name is not None and task_name == "s2t":
            return (self.text_decoder_postnet(decoder_output), None), encoder_output
        if output_type == "text":
            return (self.text_decoder_postnet(decoder_output), None), codebook_out, encoder_output
        if target_list is not None:
            return hubert_results, (self.speech_decoder_postnet(decoder_output) + (extra["attn"][0],))
        return self.speech_decoder_postnet(decoder_output) + (extra["attn"][0],)

    # ---- inference entry points (speecht5.py:1112-1249) ----
    def forward_encoder_torchscript(self, net_input: Dict[str, Tensor]):
        enc_in = {k: v for k, v in net_input.items() if k != "prev_output_tokens" and k != "task_name"}
        return self.forward_encoder(**enc_in)

    def forward_encoder(self, source, padding_mask=None):
        encoder_input, encoder_padding_mask = self.speech_encoder_prenet(source, padding_mask=padding_mask, mask=False)
        return self.encoder(encoder_input, encoder_padding_mask)

    def forward_text_encoder(self, src_tokens):
        encoder_input, encoder_padding_mask = self.text_encoder_prenet(src_tokens)
        return self.encoder(encoder_input, encoder_padding_mask)

    def forward_decoder(self, tokens, encoder_out, incremental_state):
        """`tokens` is the whole prefix (as fairseq's SequenceGenerator passes it).  With a state dict (inference) only
        the newest token is embedded and run through the decoder, whose layers keep their keys/values in the state."""
        prev_output_tokens, tgt_mask, incremental_state = self.text_decoder_prenet(tokens, incremental_state)
        decoder_output, extra = self.decoder(prev_output_tokens, tgt_mask, encoder_out=encoder_out,
                                             incremental_state=incremental_state)
        out = self.text_decoder_postnet(decoder_output)
        if incremental_state is not None:
            out = out[:, -1:]
        return out, extra

    def set_num_updates(self, num_updates):
        for m in self.modules():
            if m is not self and hasattr(m, "set_num_updates"):
                m.set_num_updates(num_updates)
        self.num_updates = num_updates

    @torch.no_grad()
    def generate_speech(self, source=None, src_tokens=None, spkembs=None, **kwargs):
        assert source is not None or src_tokens is not None
        threshold = kwargs.get("threshold", 0.5)
        minlenratio = kwargs.get("threshold", 0.0)  # (sic) the reference reads "threshold" for all three (:1191-1201)
        if source is None:
            assert src_tokens.size(0) == 1
            encoder_out = self.forward_text_encoder(src_tokens)
            maxlenratio = kwargs.get("threshold", 20.0)
        else:
            assert source.size(0) == 1
            encoder_out = self.forward_encoder(source, padding_mask=kwargs["padding_mask"])
            maxlenratio = kwargs.get("threshold", 10.0)
        T_in = encoder_out["encoder_out"][0].size(0)
        maxlen = int(T_in * maxlenratio / self.reduction_factor)
        minlen = int(T_in * minlenratio / self.reduction_factor)
        odim = self.speech_decoder_postnet.odim
        idx = 0
        ys = encoder_out["encoder_out"][0].new_zeros(1, 1, odim, dtype=torch.float32)
        outs, probs, attns = [], [], []
        post = self.speech_decoder_postnet
        incremental_states = {}
        while True:
            idx += 1
            decoder_in, _ = self.speech_decoder_prenet(ys, spkembs=spkembs)
            z, extra = self.decoder(decoder_in[:, -1:], None, encoder_out, incremental_states, alignment_layer=-1)
            zl = Fn.as_compute(z[:, -1:].contiguous())
            both = Fn.as_float(Fn.linear(zl, [post.feat_out.weight, post.prob_out.weight], [post.feat_out.bias, post.prob_out.bias]))
            nf = post.feat_out.weight.shape[0]
            outs += [both[0, 0, :nf].view(self.reduction_factor, odim)]
            probs += [torch.sigmoid(both[0, 0, nf:])]
            ys = torch.cat((ys, outs[-1][-1].view(1, 1, odim)), dim=1)
            attns.append(torch.stack([att_l[0, :, -1:] for att_l in extra["attn"][0]], dim=0) if isinstance(extra["attn"][0], list)
                         else extra["attn"][0][:, -1:].unsqueeze(0))
            if int(sum(probs[-1] >= threshold)) > 0 or idx >= maxlen:
                if idx < minlen:
                    continue
                mel = torch.cat(outs, dim=0).unsqueeze(0)  # (1, L, odim)
                if post.postnet is not None:
                    mel = post.postnet(Fn.as_compute(mel.contiguous()))   # fp32: mel + post-net residual (fused in the last BatchNorm kernel)
                mel = mel.squeeze(0)
                probs = torch.cat(probs, dim=0)
                attn = torch.cat(attns, dim=2)
                break
        if mel.size(0) == maxlen:
            logging.warning("output length reaches maximum length")
        return mel, probs, attn

    # ---- checkpoint compatibility (speecht5.py:1022-1058): non-strict per-submodule load ----
    def load_state_dict(self, state_dict, strict=True, model_cfg=None, args=None):
        """Never strict (as the reference): a checkpoint trained with another dictionary loses its dictionary-sized tensors
        (text pre/post-nets, CTC projection: speecht5.py:1036-1051), every other tensor whose shape differs is dropped with
        a warning, and each top-level module then takes what the checkpoint has for it."""
        state_dict = dict(state_dict)
        self.upgrade_state_dict_named(state_dict, "")
        post = getattr(self, "text_decoder_postnet", None)
        key = "text_decoder_postnet.output_projection.weight"
        if post is not None and key in state_dict and state_dict[key].size(0) != post.output_projection.weight.size(0):
            logger.warning(f"dictionary size differs (model {post.output_projection.weight.size(0)} vs checkpoint "
                           f"{state_dict[key].size(0)}): dictionary-sized modules keep their initialisation")
            for k in [k for k in state_dict if k.startswith(("encoder.proj", "text_encoder_prenet", "text_decoder_prenet",
                                                             "text_decoder_postnet"))]:
                state_dict.pop(k)
        own = self.state_dict()
        for k in list(state_dict.keys()):
            if k in own and own[k].shape != state_dict[k].shape:
                logger.warning(f"dropping {k}: checkpoint shape {tuple(state_dict[k].shape)} != model {tuple(own[k].shape)}")
                state_dict.pop(k)
        return super().load_state_dict(state_dict, strict=False)

    def upgrade_state_dict_named(self, state_dict, name):
        """Old-checkpoint fixes of the fairseq lineage, applied to the keys in place: fused `in_proj_weight / in_proj_bias`
        of attention blocks are split into q/k/v projections (multihead_attention.py:493-522); the numbered decoder layer
        norms `layer_norms.{0,1,2}` get their names (decoder.py:290-300)."""
        for k in list(state_dict.keys()):
            if k.endswith("in_proj_weight") or k.endswith("in_proj_bias"):
                base, kind = k.rsplit("in_proj_", 1)
                w = state_dict.pop(k)
                n = w.shape[0] // 3
                for i, pn in enumerate(("q_proj", "k_proj", "v_proj")):
                    state_dict[f"{base}{pn}.{kind}"] = w[i * n:(i + 1) * n]
        names = {"0": "self_attn_layer_norm", "1": "encoder_attn_layer_norm", "2": "final_layer_norm"}
        for k in list(state_dict.keys()):
            if ".layer_norms." in k and k.startswith((name + "decoder.") if name else "decoder."):
                head, tail = k.split(".layer_norms.", 1)
                idx, rest = tail.split(".", 1)
                if idx in names:
                    state_dict[f"{head}.{names[idx]}.{rest}"] = state_dict.pop(k)
        return state_dict

    def prune_modules(self, modules_filter=None):
        """Drop the sub-modules a fine-tuning task never uses (speecht5.py:1060-1110): `s2s`, `t2s`, `s2c`, `s3prl`."""
        if modules_filter is None:
            return

        def drop(*names):
            for n in names:
                if n in self._modules:
                    del self._modules[n]
                    if n == "quantizer":
                        self.use_codebook = False

        def drop_decoder_stack():
            for n in ("dropout_module", "layers", "layer_norm"):
                if n in self.decoder._modules:
                    del self.decoder._modules[n]

        # (the reference deletes `speech_encoder_postnet`, an attribute that does not exist -- the HuBERT head is registered as
        # `hubert_layer` -- so the head SURVIVES pruning there and its keys stay in fine-tuned checkpoints: kept here too, for
        # byte-compatible state dicts)
        common = ("speech_encoder_postnet", "projection", "quantizer")
        if modules_filter == "s2c":
            pooling = getattr(self.args, "sid_pooling_layer", "decoder")
            drop("text_encoder_prenet", "speech_decoder_postnet", "text_decoder_postnet", *common)
            if pooling != "decoder-las":
                drop("speech_decoder_prenet")
            if pooling.startswith("encoder") or getattr(self.args, "sid_decoder_speaker", False):
                drop_decoder_stack()
                drop("text_decoder_prenet")
        elif modules_filter == "s2s":
            drop("speaker_decoder_postnet", "text_encoder_prenet", "text_decoder_prenet", "text_decoder_postnet", *common)
        elif modules_filter == "t2s":
            drop("speaker_decoder_postnet", "speech_encoder_prenet", "text_decoder_prenet", "text_decoder_postnet", *common)
        elif modules_filter == "s3prl":
            drop_decoder_stack()
            drop("speaker_decoder_postnet", "text_decoder_prenet", "text_decoder_postnet", "speech_decoder_prenet",
                 "speech_decoder_postnet", *common)
        else:
            raise ValueError(f"unknown modules filter {modules_filter!r}")
        if getattr(self.encoder, "proj", None) is not None:
            self.encoder.proj = None


# ---- architectures (speecht5.py:1252-1447) ----
@register_model_architecture(model_name="t5_transformer", arch_name="t5_transformer")
def base_architecture(args):
    def d(name, value):
        setattr(args, name, getattr(args, name, value))

    d("bert_init", False)
    d("encoder_embed_dim", 768)
    d("encoder_ffn_embed_dim", 768 * 4)
    d("encoder_layers", 12)
    d("encoder_attention_heads", 12)
    d("encoder_normalize_before", False)
    d("decoder_embed_dim", args.encoder_embed_dim)
    d("decoder_ffn_embed_dim", args.encoder_ffn_embed_dim)
    d("decoder_layers", 6)
    d("decoder_attention_heads", 12)
    d("decoder_normalize_before", False)
    d("dropout", 0.1)
    d("attention_dropout", args.dropout)
    d("activation_dropout", args.dropout)
    d("activation_fn", "gelu")
    d("decoder_layerdrop", 0.0)
    d("decoder_output_dim", args.decoder_embed_dim)
    d("decoder_input_dim", args.decoder_embed_dim)
    d("encoder_layerdrop", 0)
    d("max_text_positions", DEFAULT_MAX_TEXT_POSITIONS)
    d("max_speech_positions", DEFAULT_MAX_SPEECH_POSITIONS)
    d("use_batch_norm", True)
    d("enc_use_scaled_pos_enc", True)
    d("dec_use_scaled_pos_enc", True)
    d("postnet_layers", 5)
    d("postnet_chans", 256)
    d("postnet_filts", 5)
    d("postnet_dropout_rate", 0.5)
    d("dprenet_dropout_rate", 0.5)
    d("dprenet_layers", 2)
    d("dprenet_units", 256)
    d("initial_encoder_alpha", 1.0)
    d("initial_decoder_alpha", 1.0)
    d("spk_embed_integration_type", "pre")
    d("spk_embed_dim", 512)
    d("encoder_reduction_factor", 1)
    d("reduction_factor", 2)
    d("transformer_enc_positional_dropout_rate", 0.1)
    d("transformer_dec_positional_dropout_rate", 0.1)
    d("layer_norm_eps", 1e-5)
    d("no_scale_embedding", True)
    d("encoder_speech_prenet", "conv")
    d("quant_noise_pq", 0)
    d("adaptive_softmax_cutoff", None)
    d("adaptive_softmax_dropout", 0)
    d("no_token_positional_embeddings", False)
    d("adaptive_input", False)
    d("decoder_learned_pos", False)
    d("share_input_output_embed", False)
    d("share_ctc_embed", False)
    d("freeze_encoder_updates", 0)
    d("freeze_decoder_updates", 0)
    d("no_freeze_encoder_layer", None)
    d("modules_filter", None)
    d("conv_pos", 128)
    d("conv_pos_groups", 16)
    d("target_glu", False)
    d("logit_temp", 0.1)
    d("final_dim", 256)
    d("untie_final_proj", True)
    d("feature_grad_mult", 0.1)
    d("use_sent_enc_layer", True)
    d("extractor_mode", "default")
    d("conv_feature_layers", "[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2")
    d("conv_bias", False)
    d("hubert_mask_length", 10)
    d("mask_prob", 0.0)
    d("mask_selection", "static")
    d("mask_other", 0)
    d("no_mask_overlap", False)
    d("mask_min_space", 1)
    d("mask_channel_length", 10)
    d("mask_channel_prob", 0.0)
    d("mask_channel_selection", "static")
    d("mask_channel_other", 0)
    d("no_mask_channel_overlap", False)
    d("mask_channel_min_space", 1)
    d("skip_masked", False)
    d("skip_nomask", False)
    d("use_conv_pos", False)
    d("use_sinc_pos", False)
    d("use_codebook", False)
    d("latent_vars", 100)
    d("latent_groups", 2)
    d("latent_dim", 0)
    d("latent_temp", (2, 0.5, 0.999995))
    d("quantizer_depth", 1)
    d("quantizer_factor", 3)
    d("codebook_prob", 0.5)
    d("relative_position_embedding", False)
    d("num_buckets", 320)
    d("max_distance", 1280)
    d("encoder_max_relative_position", 160)
    d("decoder_max_relative_position", 160)


@register_model_architecture("t5_transformer", "t5_transformer_base")
def t5_transformer_base(args):
    def d(name, value):
        setattr(args, name, getattr(args, name, value))
    d("use_conv_pos", True)
    d("use_sinc_pos", True)
    d("layernorm_embedding", False)
    d("encoder_normalize_before", False)
    d("decoder_normalize_before", False)
    d("layer_norm_first", False)
    d("relative_position_embedding", True)
    d("dropout", 0.1)
    d("activation_dropout", 0.0)
    d("attention_dropout", 0.1)
    d("encoder_layerdrop", 0.05)
    d("decoder_layerdrop", 0.05)
    d("mask_prob", 0.80)
    base_architecture(args)


@register_model_architecture("t5_transformer", "t5_transformer_large")
def t5_transformer_large(args):
    def d(name, value):
        setattr(args, name, getattr(args, name, value))
    d("use_conv_pos", True)
    d("use_sinc_pos", True)
    d("encoder_normalize_before", False)
    d("decoder_normalize_before", True)
    d("layer_norm_first", True)
    d("relative_position_embedding", True)
    d("dropout", 0.0)
    d("activation_dropout", 0.0)
    d("attention_dropout", 0.0)
    d("encoder_layerdrop", 0.0)
    d("decoder_layerdrop", 0.0)
    d("encoder_embed_dim", 1024)
    d("encoder_layers", 24)
    d("decoder_layers", 6)
    d("encoder_ffn_embed_dim", 4096)
    d("encoder_attention_heads", 16)
    d("decoder_attention_heads", 16)
    d("feature_grad_mult", 1.0)
    d("extractor_mode", "layer_norm")
    d("final_dim", 768)
    d("mask_prob", 0.80)
    base_architecture(args)


@register_model_architecture("t5_transformer", "t5_transformer_base_asr")
def t5_transformer_base_asr(args):
    def d(name, value):
        setattr(args, name, getattr(args, name, value))
    d("use_conv_pos", True)
    d("use_sinc_pos", True)
    d("encoder_normalize_before", False)
    d("decoder_normalize_before", False)
    d("layer_norm_first", False)
    d("relative_position_embedding", True)
    d("dropout", 0.1)
    d("activation_dropout", 0.1)
    d("attention_dropout", 0.1)
    d("feature_grad_mult", 0.0)
    d("encoder_layerdrop", 0.1)
    d("decoder_layerdrop", 0.1)
    d("mask_prob", 0.75)
    d("mask_selection", "static")
    d("mask_channel_length", 64)
    d("mask_channel_prob", 0.5)
    d("mask_channel_selection", "static")
    d("max_text_positions", 600)
    base_architecture(args)

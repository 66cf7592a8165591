"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the product path `speecht5_amd/`).

A functional, plain-PyTorch fp32 restatement of the SpeechT5 forward path (microsoft/SpeechT5,
SpeechT5/speecht5/models/) that runs without fairseq/espnet and therefore travels to the GPU box,
where /root/reference does not exist.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import it, and only as the checker.  Backward = torch autograd of this
forward, exactly as in the reference.

Pinning: the reference ships no tests or golden vectors, so this file is
pinned against outputs of the reference itself, run here: `oracle/make_golden.py` imports the
*verbatim* reference modules (through `oracle/ref_stubs.py`), and `tests/test_oracle_golden.py`
checks every function below against the committed fixtures in `tests/golden/`.

All weights are read from a flat state dict `sd` with the reference's parameter names
(SURVEY.md Appendix C); `cfg` is an argparse-like namespace with the reference's arg names
(speecht5.py:1252-1383).  Stochastic pieces (dropout, HuBERT span mask, Gumbel noise, codebook
time-mix, LayerDrop) are inputs or disabled -- parity runs use p = 0 / given masks, as the
reference would under the same seeds.  Citations are file:line under
/root/reference/SpeechT5/speecht5/.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------
def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(sd, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def conv_out_lengths(lengths, conv_layers):
    """models/modules/speech_encoder_prenet.py:370-374."""
    out = lengths.clone()
    for _, k, s in conv_layers:
        out = ((out.float() - (k - 1) - 1) / s + 1).floor().long()
    return out


def fairseq_sinusoidal_table(num_embeddings, dim, padding_idx):
    """fairseq SinusoidalPositionalEmbedding.get_embedding (vendored copy SpeechLM/modules.py:1318-1341):
    [sin | cos] halves (NOT interleaved), frequency exp(-i*ln(1e4)/(dim/2-1)), zero pad row."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    ang = torch.arange(num_embeddings, dtype=torch.float).unsqueeze(1) * freq.unsqueeze(0)
    emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    if dim % 2 == 1:
        emb = torch.cat([emb, torch.zeros(num_embeddings, 1)], dim=1)
    if padding_idx is not None:
        emb[padding_idx, :] = 0
    return emb


def fairseq_positions(tokens_ne_pad, padding_idx):
    """fairseq utils.make_positions (SpeechLM/modules.py:32-41): cumsum over non-pad, offset by pad."""
    m = tokens_ne_pad.int()
    return (torch.cumsum(m, dim=1).type_as(m) * m).long() + padding_idx


def espnet_pe_table(length, dim):
    """espnet PositionalEncoding.extend_pe: interleaved sin/cos, frequency exp(-2i*ln(1e4)/dim)."""
    pe = torch.zeros(length, dim)
    pos = torch.arange(0, length, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(10000.0) / dim))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


# --------------------------------------------------------------------------------------------
# speech encoder pre-net   (models/modules/speech_encoder_prenet.py)
# --------------------------------------------------------------------------------------------
def conv_feature_extractor(sd, prefix, wav, cfg):
    """ConvFeatureExtractionModel.forward (:349-354); blocks (:290-347). wav [B,S] -> [B,C,T]."""
    layers = eval(cfg.conv_feature_layers)
    x = wav.unsqueeze(1)
    for i, (dim, k, s) in enumerate(layers):
        p = f"{prefix}conv_layers.{i}."
        x = F.conv1d(x, sd[p + "0.weight"], sd.get(p + "0.bias"), stride=s)
        if cfg.extractor_mode == "layer_norm":
            x = F.layer_norm(x.transpose(-2, -1).float(), (dim,), sd[p + "2.1.weight"], sd[p + "2.1.bias"]).transpose(-2, -1)
        elif i == 0:
            x = F.group_norm(x.float(), dim, sd[p + "2.weight"], sd[p + "2.bias"], 1e-5)
        x = F.gelu(x)
    return x


def speech_encoder_prenet(sd, cfg, wav, padding_mask, *, target_list=None, mask_indices=None,
                          feature_grad_scale=True):
    """SpeechEncoderPrenet._forward (:155-204).  `mask_indices` (bool [B,T]) replaces the numpy-RNG
    span mask of apply_hubert_mask (:234-249); None = no masking.
    Returns dict(x [B,T,d], padding_mask [B,T], features_pen, target_list)."""
    pre = "speech_encoder_prenet."
    feats = conv_feature_extractor(sd, pre + "feature_extractor.", wav, cfg)  # [B,C,T]
    if feature_grad_scale and cfg.feature_grad_mult != 1.0 and cfg.feature_grad_mult > 0 and feats.requires_grad:
        g = cfg.feature_grad_mult  # GradMultiply (:158-160): identity forward, grad * g
        feats = feats * g + (feats * (1 - g)).detach()
    elif cfg.feature_grad_mult <= 0:
        feats = feats.detach()
    if target_list is not None:  # forward_targets (:206-217)
        ratio = cfg.label_rates * math.prod(s for _, _, s in eval(cfg.conv_feature_layers)) / cfg.sample_rate
        feat_tsz = feats.size(2)
        targ_tsz = min(t.size(1) for t in target_list)
        if ratio * feat_tsz > targ_tsz:
            feat_tsz = int(targ_tsz / ratio)
            feats = feats[..., :feat_tsz]
        inds = (torch.arange(feat_tsz).float() * ratio).long()
        target_list = [t[:, inds] for t in target_list]
    features_pen = feats.float().pow(2).mean()  # (:172)
    x = feats.transpose(1, 2)
    x = _ln(sd, pre + "layer_norm", x)  # (:174)
    # forward_padding_mask (:219-229)
    B, T = x.shape[:2]
    extra = padding_mask.size(1) % T
    pm = padding_mask[:, :-extra] if extra > 0 else padding_mask
    pm = pm.view(B, T, -1).all(-1)
    if (pre + "post_extract_proj.weight") in sd:
        x = _lin(sd, pre + "post_extract_proj", x)  # (:176-177)
    if mask_indices is not None:  # (:249)
        x = x.clone()
        x[mask_indices] = sd[pre + "mask_emb"]
    if cfg.use_conv_pos:  # (:187-192) weight-normed grouped conv, SamePad, GELU
        v, g = sd[pre + "pos_conv.0.weight_v"], sd[pre + "pos_conv.0.weight_g"]
        w = v * (g / v.norm(dim=(0, 1), keepdim=True))  # nn.utils.weight_norm(dim=2)
        k = cfg.conv_pos
        pos = F.conv1d(x.transpose(1, 2), w, sd[pre + "pos_conv.0.bias"], padding=k // 2, groups=cfg.conv_pos_groups)
        if k % 2 == 0:
            pos = pos[:, :, :-1]  # SamePad
        x = x + F.gelu(pos).transpose(1, 2)
    if cfg.use_sinc_pos:  # (:194-196): fairseq sinusoid indexed by the bool padding mask, pad idx 1
        positions = fairseq_positions(pm.long().ne(1), 1)
        table = fairseq_sinusoidal_table(int(positions.max()) + 1 + 1, x.shape[-1], 1)
        x = x + table[positions]
    return dict(x=x, padding_mask=pm, features_pen=features_pen, target_list=target_list)


# --------------------------------------------------------------------------------------------
# attention   (models/modules/multihead_attention.py:202-407)
# --------------------------------------------------------------------------------------------
def relative_position_keys(pe_k, T, maxlen):
    """encoder.py:52-59,240-244: pe_k[clip(i-j, -maxlen, maxlen-1) + maxlen] -> [T,T,hd]."""
    pos = torch.arange(T)
    d = (pos[:, None] - pos[None, :]).clamp(-maxlen, maxlen - 1) + maxlen
    return pe_k[d]


def multihead_attention(sd, prefix, query, key, H, *, key_padding_mask=None, attn_mask=None, position_bias=None,
                        need_head_weights=False):
    """Inputs T x B x C (query) / S x B x C (key = value source).  Returns (out [T,B,C], probs [B,H,T,S])."""
    T, B, C = query.shape
    S = key.shape[0]
    hd = C // H
    q = _lin(sd, prefix + "q_proj", query) * hd ** -0.5  # (:213,232)
    k = _lin(sd, prefix + "k_proj", key)
    v = _lin(sd, prefix + "v_proj", key)
    q = q.contiguous().view(T, B * H, hd).transpose(0, 1)
    k = k.contiguous().view(S, B * H, hd).transpose(0, 1)
    v = v.contiguous().view(S, B * H, hd).transpose(0, 1)
    w = torch.bmm(q, k.transpose(1, 2))  # (:340)
    if position_bias is not None:  # (:343-353): B[bh,i,j] = q[bh,i] . pos_k[i,j]
        w = w + torch.matmul(q.transpose(0, 1), position_bias.transpose(-2, -1)).transpose(0, 1)
    if attn_mask is not None:
        w = w + attn_mask.unsqueeze(0)
    if key_padding_mask is not None:  # (:365-377)
        w = w.view(B, H, T, S).masked_fill(key_padding_mask[:, None, None, :].to(torch.bool), float("-inf")).view(B * H, T, S)
    p = F.softmax(w, dim=-1, dtype=torch.float32)  # (:382)
    out = torch.bmm(p, v).transpose(0, 1).contiguous().view(T, B, C)  # (:389-395)
    out = _lin(sd, prefix + "out_proj", out)
    return out, p.view(B, H, T, S)


# --------------------------------------------------------------------------------------------
# encoder   (models/modules/encoder.py:134-291, transformer_layer.py:75-134)
# --------------------------------------------------------------------------------------------
def encoder_layer(sd, p, x, cfg, pad_mask, pos_k):
    H = cfg.encoder_attention_heads
    if cfg.layer_norm_first:  # (:90-111)
        res = x
        h = _ln(sd, p + "self_attn_layer_norm", x)
        pb = _ln(sd, p + "norm_k", pos_k) if pos_k is not None else None
        h, _ = multihead_attention(sd, p + "self_attn.", h, h, H, key_padding_mask=pad_mask, position_bias=pb)
        x = res + h
        res = x
        h = _ln(sd, p + "final_layer_norm", x)
        h = _lin(sd, p + "fc2", F.gelu(_lin(sd, p + "fc1", h)))
        x = res + h
    else:  # post-LN (:112-132)
        h, _ = multihead_attention(sd, p + "self_attn.", x, x, H, key_padding_mask=pad_mask, position_bias=pos_k)
        x = _ln(sd, p + "self_attn_layer_norm", x + h)
        h = _lin(sd, p + "fc2", F.gelu(_lin(sd, p + "fc1", x)))
        x = _ln(sd, p + "final_layer_norm", x + h)
    return x


def encoder(sd, cfg, x, pad_mask):
    """TransformerEncoder.forward / forward_scriptable.  x [B,T,C] -> dict like encoder.py:285-291."""
    assert cfg.activation_fn == "gelu"
    if not cfg.layer_norm_first:
        x = _ln(sd, "encoder.layer_norm", x, cfg.layer_norm_eps)  # (:226-227)
    x = x.transpose(0, 1)
    pos_k = None
    if cfg.relative_position_embedding:
        pos_k = relative_position_keys(sd["encoder.pos_emb.pe_k.weight"], x.shape[0], cfg.encoder_max_relative_position)
    for i in range(cfg.encoder_layers):
        x = encoder_layer(sd, f"encoder.layers.{i}.", x, cfg, pad_mask, pos_k)
    if cfg.layer_norm_first:
        x = _ln(sd, "encoder.layer_norm", x, cfg.layer_norm_eps)  # (:275-276)
    ctc = _lin(sd, "encoder.proj", x) if "encoder.proj.weight" in sd else None  # (:173-179)
    return {"encoder_out": [x], "encoder_padding_mask": [pad_mask], "encoder_out_for_ctc": [ctc]}


# --------------------------------------------------------------------------------------------
# decoder   (models/modules/decoder.py:171-269, transformer_layer.py:262-404)
# --------------------------------------------------------------------------------------------
def decoder_layer(sd, p, x, cfg, enc, enc_pad, self_mask, self_pad):
    H = cfg.decoder_attention_heads
    nb = cfg.decoder_normalize_before
    res = x
    h = _ln(sd, p + "self_attn_layer_norm", x) if nb else x
    h, _ = multihead_attention(sd, p + "self_attn.", h, h, H, key_padding_mask=self_pad, attn_mask=self_mask)  # no rel-pos (:241)
    x = res + h
    if not nb:
        x = _ln(sd, p + "self_attn_layer_norm", x)
    res = x
    h = _ln(sd, p + "encoder_attn_layer_norm", x) if nb else x
    h, attn = multihead_attention(sd, p + "encoder_attn.", h, enc, H, key_padding_mask=enc_pad)
    x = res + h
    if not nb:
        x = _ln(sd, p + "encoder_attn_layer_norm", x)
    res = x
    h = _ln(sd, p + "final_layer_norm", x) if nb else x
    h = _lin(sd, p + "fc2", F.gelu(_lin(sd, p + "fc1", h)))
    x = res + h
    if not nb:
        x = _ln(sd, p + "final_layer_norm", x)
    return x, attn


def decoder(sd, cfg, x, tgt_mask, enc_out, *, alignment_layer=None, causal=True):
    """x [B,T,C].  Returns (x [B,T,C], attn) with attn as decoder.py:256-269: head-mean [B,T,S] of
    the last layer by default; list of per-layer [B,H,T,S] when alignment_layer == -1."""
    enc = enc_out["encoder_out"][0]
    enc_pad = enc_out["encoder_padding_mask"][0]
    x = x.transpose(0, 1)
    T = x.shape[0]
    self_mask = torch.triu(torch.full((T, T), float("-inf")), 1) if causal else None
    L = cfg.decoder_layers
    if alignment_layer is None:
        alignment_layer = L - 1
    attns = []
    for i in range(L):
        x, a = decoder_layer(sd, f"decoder.layers.{i}.", x, cfg, enc, enc_pad, self_mask, tgt_mask)
        if i == alignment_layer or alignment_layer == -1:
            attns.append(a)  # [B,H,T,S]  (reference: [H,B,T,S].transpose(0,1))
    if "decoder.layer_norm.weight" in sd:
        x = _ln(sd, "decoder.layer_norm", x, cfg.layer_norm_eps)
    attn = attns[0].mean(dim=1) if len(attns) == 1 else attns
    return x.transpose(0, 1), attn


# --------------------------------------------------------------------------------------------
# text / speech pre- and post-nets
# --------------------------------------------------------------------------------------------
def text_encoder_prenet(sd, cfg, tokens, pad_idx=1):
    """text_encoder_prenet.py:44-45: Embedding + espnet ScaledPositionalEncoding (x + alpha*pe)."""
    x = F.embedding(tokens, sd["text_encoder_prenet.encoder_prenet.0.weight"], pad_idx)
    pe = espnet_pe_table(x.shape[1], x.shape[2])
    if cfg.enc_use_scaled_pos_enc:
        x = x + sd["text_encoder_prenet.encoder_prenet.1.alpha"] * pe
    else:
        x = x * math.sqrt(x.shape[2]) + pe
    return x, tokens.eq(pad_idx)


def text_decoder_prenet(sd, cfg, tokens, pad_idx=1):
    """text_decoder_prenet.py:89-124 (no incremental state): embed_scale*emb + fairseq sinusoid."""
    mask = tokens.eq(pad_idx) if tokens.eq(pad_idx).any() else None
    d = cfg.decoder_embed_dim
    scale = 1.0 if cfg.no_scale_embedding else math.sqrt(d)
    x = scale * F.embedding(tokens, sd["text_decoder_prenet.embed_tokens.weight"], pad_idx)
    positions = fairseq_positions(tokens.ne(pad_idx), pad_idx)
    x = x + fairseq_sinusoidal_table(int(positions.max()) + 1, d, pad_idx)[positions]
    return x, mask


def text_decoder_postnet(sd, x):
    return F.linear(x, sd["text_decoder_postnet.output_projection.weight"])


def speech_decoder_prenet(sd, cfg, prev, tgt_lengths=None, spkembs=None, prenet_dropout_masks=None):
    """speech_decoder_prenet.py:76-89.  The Tacotron pre-net dropout (always on in the reference) is
    supplied as explicit masks (list of [B,T,units] already scaled by 1/(1-p)) or skipped (p=0)."""
    p = "speech_decoder_prenet.decoder_prenet.0.0.prenet."
    x = prev
    for i in range(cfg.dprenet_layers):
        x = F.relu(_lin(sd, f"{p}{i}.0", x))
        if prenet_dropout_masks is not None:
            x = x * prenet_dropout_masks[i]
    x = _lin(sd, "speech_decoder_prenet.decoder_prenet.0.1", x)
    pe = espnet_pe_table(x.shape[1], x.shape[2])
    x = x + sd["speech_decoder_prenet.decoder_prenet.1.alpha"] * pe
    if spkembs is not None:
        s = F.normalize(spkembs).unsqueeze(1).expand(-1, x.size(1), -1)
        x = F.relu(_lin(sd, "speech_decoder_prenet.spkembs_layer.0", torch.cat([x, s], dim=-1)))
    mask = None
    if tgt_lengths is not None:
        mask = torch.arange(int(max(tgt_lengths)))[None, :] >= torch.as_tensor(tgt_lengths)[:, None]
    return x, mask


def speech_decoder_postnet(sd, cfg, zs, training=True, bn_stats=None):
    """speech_decoder_postnet.py:57-72 + espnet Postnet (5x Conv1d k5 + BatchNorm1d + tanh)."""
    odim = cfg.speech_odim
    before = _lin(sd, "speech_decoder_postnet.feat_out", zs).view(zs.size(0), -1, odim)
    logits = _lin(sd, "speech_decoder_postnet.prob_out", zs).view(zs.size(0), -1)
    x = before.transpose(1, 2)
    n = cfg.postnet_layers
    for i in range(n):
        p = f"speech_decoder_postnet.postnet.postnet.{i}."
        x = F.conv1d(x, sd[p + "0.weight"], padding=(cfg.postnet_filts - 1) // 2)
        x = F.batch_norm(x, None if training else sd[p + "1.running_mean"], None if training else sd[p + "1.running_var"],
                         sd[p + "1.weight"], sd[p + "1.bias"], training=training, eps=1e-5)
        if i < n - 1:
            x = torch.tanh(x)
    after = before + x.transpose(1, 2)
    return before, after, logits


def hubert_logits(sd, cfg, x, pad_mask, mask_indices, target_list):
    """speech_encoder_postnet.py:56-124 (single label set, untie_final_proj with one dictionary)."""
    emb = sd["hubert_layer.label_embs_concat"]

    def nce(proj_x, target):
        y = emb[target.long()]
        negs = emb.unsqueeze(1).expand(-1, proj_x.size(0), -1)
        neg_is_pos = (y == negs).all(-1)
        targets = torch.cat([y.unsqueeze(0), negs], dim=0)
        logits = torch.cosine_similarity(proj_x.float(), targets.float(), dim=-1) / cfg.logit_temp
        if neg_is_pos.any():
            logits[1:][neg_is_pos] = float("-inf")
        return logits.transpose(0, 1)

    t = target_list[0]
    m = torch.logical_and(~pad_mask, mask_indices)
    u = torch.logical_and(~pad_mask, ~mask_indices)
    logit_m = nce(_lin(sd, "hubert_layer.final_proj", x[m]), t[m])
    logit_u = nce(_lin(sd, "hubert_layer.final_proj", x[u]), t[u])
    return logit_m, logit_u


def gumbel_quantizer(sd, cfg, x, gumbel_noise=None, tau=2.0):
    """fairseq GumbelVectorQuantizer.forward (restated in SURVEY.md App. A; call site speecht5.py:858).
    `gumbel_noise` ([B*T*G, V], -log(Exp(1))) turns on the training path; None = eval (hard arg-max)."""
    G, V = cfg.latent_groups, cfg.latent_vars
    B, T, C = x.shape
    logits = _lin(sd, "quantizer.weight_proj", x.reshape(-1, C)).view(B * T * G, V)
    k = logits.argmax(-1)
    hard = torch.zeros_like(logits).scatter_(-1, k.view(-1, 1), 1.0).view(B * T, G, V)
    hp = hard.float().mean(0)
    code_ppl = torch.exp(-torch.sum(hp * torch.log(hp + 1e-7), dim=-1)).sum()
    ap = torch.softmax(logits.view(B * T, G, V).float(), dim=-1).mean(0)
    prob_ppl = torch.exp(-torch.sum(ap * torch.log(ap + 1e-7), dim=-1)).sum()
    if gumbel_noise is not None:  # F.gumbel_softmax(hard=True) with the given noise
        y = ((logits.float() + gumbel_noise) / tau).softmax(-1)
        idx = y.argmax(-1, keepdim=True)
        y_hard = torch.zeros_like(y).scatter_(-1, idx, 1.0)
        sel = (y_hard - y.detach() + y).view(B * T, G * V)
    else:
        sel = hard.view(B * T, G * V)
    vars_ = sd["quantizer.vars"]  # [1, G*V, C/G]
    q = (sel.unsqueeze(-1) * vars_).view(B * T, G, V, -1).sum(-2).view(B, T, -1)
    return dict(x=q, code_perplexity=code_ppl, prob_perplexity=prob_ppl, num_vars=G * V)


# --------------------------------------------------------------------------------------------
# T5TransformerModel.forward paths   (models/speecht5.py:786-963)
# --------------------------------------------------------------------------------------------
def codebook_mix(sd, cfg, enc_out, mix_idx=None, gumbel_noise=None, tau=2.0):
    """speecht5.py:858-877: replace the time steps in `mix_idx` by their quantised vectors."""
    x = enc_out["encoder_out"][0].transpose(0, 1)
    q = gumbel_quantizer(sd, cfg, x, gumbel_noise, tau)
    w = x.new_zeros(x.size(1))
    if mix_idx is not None:
        w[mix_idx] = 1.0
    enc_out["encoder_out"][0] = (w.view(-1, 1) * q["x"] + (1 - w).view(-1, 1) * x).transpose(0, 1)
    return q


def forward_speech_pretrain(sd, cfg, sample, *, mask_indices, mix_idx=None, gumbel_noise=None, training=True):
    """task speech_pretrain: returns dict of everything the criterion consumes."""
    ni = sample["net_input"]
    pre = speech_encoder_prenet(sd, cfg, ni["source"], ni["padding_mask"], target_list=sample["target_list"],
                                mask_indices=mask_indices)
    enc = encoder(sd, cfg, pre["x"], pre["padding_mask"])
    logit_m, logit_u = hubert_logits(sd, cfg, enc["encoder_out"][0].transpose(0, 1), pre["padding_mask"], mask_indices,
                                     pre["target_list"])
    out = dict(logit_m=logit_m, logit_u=logit_u, features_pen=pre["features_pen"], encoder_out=enc["encoder_out"][0])
    if cfg.use_codebook:
        q = codebook_mix(sd, cfg, enc, mix_idx, gumbel_noise)
        out.update(prob_perplexity=q["prob_perplexity"], code_perplexity=q["code_perplexity"], num_vars=q["num_vars"])
    x, tmask = speech_decoder_prenet(sd, cfg, ni["prev_output_tokens"], ni["tgt_lengths"], ni["spkembs"])
    y, attn = decoder(sd, cfg, x, tmask, enc)
    before, after, logits = speech_decoder_postnet(sd, cfg, y, training=training)
    out.update(before=before, after=after, stop_logits=logits, attn=attn, decoder_out=y)
    return out


def forward_text_pretrain(sd, cfg, sample, *, mix_idx=None, gumbel_noise=None):
    ni = sample["net_input"]
    x, pad = text_encoder_prenet(sd, cfg, ni["src_tokens"])
    enc = encoder(sd, cfg, x, pad)
    out = {}
    if cfg.use_codebook:
        q = codebook_mix(sd, cfg, enc, mix_idx, gumbel_noise)
        out.update(prob_perplexity=q["prob_perplexity"], code_perplexity=q["code_perplexity"], num_vars=q["num_vars"])
    y, tmask = text_decoder_prenet(sd, cfg, ni["prev_output_tokens"])
    y, _ = decoder(sd, cfg, y, tmask, enc)
    out["logits"] = text_decoder_postnet(sd, y)
    return out


def forward_s2t(sd, cfg, sample, *, mask_indices=None):
    ni = sample["net_input"]
    pre = speech_encoder_prenet(sd, cfg, ni["source"], ni["padding_mask"], mask_indices=mask_indices)
    enc = encoder(sd, cfg, pre["x"], pre["padding_mask"])
    y, tmask = text_decoder_prenet(sd, cfg, ni["prev_output_tokens"])
    y, _ = decoder(sd, cfg, y, tmask, enc)
    return dict(logits=text_decoder_postnet(sd, y), encoder_out=enc)


def forward_t2s(sd, cfg, sample, training=True):
    ni = sample["net_input"]
    x, pad = text_encoder_prenet(sd, cfg, ni["src_tokens"])
    enc = encoder(sd, cfg, x, pad)
    y, tmask = speech_decoder_prenet(sd, cfg, ni["prev_output_tokens"], ni["tgt_lengths"], ni["spkembs"])
    y, attn = decoder(sd, cfg, y, tmask, enc, alignment_layer=-1)
    before, after, logits = speech_decoder_postnet(sd, cfg, y, training=training)
    return dict(before=before, after=after, stop_logits=logits, attn=attn)


# --------------------------------------------------------------------------------------------
# criteria   (criterions/*.py)
# --------------------------------------------------------------------------------------------
def tacotron_loss(before, after, logits, sample, reduction_factor, bce_pos_weight=5.0):
    """text_to_speech_loss.py:154-214 (+ Tacotron2Loss :296-345, use_masking=True)."""
    ys, labels, olens = sample["dec_target"], sample["labels"], sample["dec_target_lengths"]
    r = reduction_factor
    dev = ys.device
    olens = torch.as_tensor(olens).to(dev)
    if r > 1:
        olens = torch.as_tensor([int(o) - int(o) % r for o in olens], device=dev)
        mx = int(olens.max())
        ys, labels = ys[:, :mx], labels[:, :mx]
        labels = torch.scatter(labels, 1, (olens - 1).unsqueeze(1), 1.0)
    m = (torch.arange(ys.shape[1], device=dev)[None, :] < olens[:, None]).unsqueeze(-1)
    ysm, am, bm = ys.masked_select(m), after.masked_select(m), before.masked_select(m)
    lm, lg = labels.masked_select(m[:, :, 0]), logits.masked_select(m[:, :, 0])
    l1 = F.l1_loss(am, ysm) + F.l1_loss(bm, ysm)
    l2 = F.mse_loss(am, ysm) + F.mse_loss(bm, ysm)
    bce = F.binary_cross_entropy_with_logits(lg, lm, pos_weight=torch.tensor(bce_pos_weight, device=dev))
    return l1 + bce, l1, l2, bce


def guided_attention_loss(att_ws, ilens, olens, sigma=0.4, alpha=10.0):
    """text_to_speech_loss.py:379-427.  att_ws [B, heads*, T_out, T_in]."""
    B, _, To, Ti = att_ws.shape
    gm = torch.zeros(B, To, Ti)
    mk = torch.zeros(B, To, Ti, dtype=torch.bool)
    for b in range(B):
        il, ol = int(ilens[b]), int(olens[b])
        gx, gy = torch.meshgrid(torch.arange(ol).float(), torch.arange(il).float(), indexing="ij")
        gm[b, :ol, :il] = 1.0 - torch.exp(-((gy / il - gx / ol) ** 2) / (2 * sigma ** 2))
        mk[b, :ol, :il] = True
    return alpha * torch.mean((gm.unsqueeze(1) * att_ws).masked_select(mk.unsqueeze(1)))


def speech_pretrain_loss(out, sample, cfg, loss_weights=(10.0, 0.1), pred_masked_weight=1.0, pred_nomask_weight=0.0,
                         hubert_weight=1.0, dec_weight=1.0):
    """speech_pretrain_criterion.py:83-198.  Returns (loss, sample_size, parts)."""
    tm = out["logit_m"].new_zeros(out["logit_m"].size(0), dtype=torch.long)
    loss_m = F.cross_entropy(out["logit_m"].float(), tm, reduction="sum")
    loss, sample_size = 0.0, 0
    if pred_masked_weight > 0:
        loss = loss + pred_masked_weight * loss_m
        sample_size += tm.numel()
    loss_u = None
    if out["logit_u"] is not None and out["logit_u"].numel() > 0:
        tu = out["logit_u"].new_zeros(out["logit_u"].size(0), dtype=torch.long)
        loss_u = F.cross_entropy(out["logit_u"].float(), tu, reduction="sum")
        if pred_nomask_weight > 0:
            loss = loss + pred_nomask_weight * loss_u
            sample_size += tu.numel()
    extra = [out["features_pen"]]
    if "prob_perplexity" in out:
        extra.append((out["num_vars"] - out["prob_perplexity"]) / out["num_vars"])
    for p, coef in zip(extra, list(loss_weights)[:len(extra)]):
        if coef != 0:
            loss = loss + coef * p.float() * sample_size
    dec_loss, l1, l2, bce = tacotron_loss(out["before"], out["after"], out["stop_logits"], sample, cfg.reduction_factor)
    total = hubert_weight * loss + dec_weight * sample_size * dec_loss
    return total, sample_size, dict(loss_m=loss_m, loss_u=loss_u, dec_loss=dec_loss, l1=l1, l2=l2, bce=bce)


def text_pretrain_loss(out, sample, loss_weights=(0.1,), pad_idx=1):
    """text_pretrain_criterion.py:42-101."""
    lp = F.log_softmax(out["logits"].float(), dim=-1)
    tgt = sample["target"].view(-1)
    loss = F.nll_loss(lp.view(-1, lp.size(-1)), tgt, ignore_index=pad_idx, reduction="sum")
    sample_size = sample["ntokens"]
    total = loss
    if "prob_perplexity" in out:
        total = total + loss_weights[-1] * ((out["num_vars"] - out["prob_perplexity"]) / out["num_vars"]).float() * sample_size
    return total, sample_size, dict(bart_loss=loss)


def s2t_loss(out, sample, cfg, ce_weight=0.5, ctc_weight=0.5, label_smoothing=0.1, pad_idx=1, blank_idx=None):
    """speech_to_text_loss.py:186-337."""
    lp = F.log_softmax(out["logits"].float(), dim=-1)
    tgt = sample["target"]
    nll = -lp.gather(-1, tgt.unsqueeze(-1))
    smooth = -lp.sum(-1, keepdim=True)
    pm = tgt.unsqueeze(-1).eq(pad_idx)
    nll = nll.masked_fill(pm, 0.0).sum()
    smooth = smooth.masked_fill(pm, 0.0).sum()
    eps_i = label_smoothing / (lp.size(-1) - 1)
    ce = (1.0 - label_smoothing - eps_i) * nll + eps_i * smooth
    total = ce_weight * ce
    parts = dict(ce=ce, nll=nll)
    if ctc_weight > 0:
        logits = out["encoder_out"]["encoder_out_for_ctc"][0]  # [T,B,V]
        lprobs = F.log_softmax(logits.float(), dim=-1)
        epm = out["encoder_out"]["encoder_padding_mask"][0]
        in_len = (~epm).long().sum(-1)
        tmask = (tgt != pad_idx) & (tgt != 2)  # strip pad and eos
        flat = tgt.masked_select(tmask)
        tlen = tmask.sum(-1)
        ctc = F.ctc_loss(lprobs, flat, in_len, tlen, blank=blank_idx, reduction="sum", zero_infinity=True)
        total = total + ctc_weight * ctc
        parts["ctc"] = ctc
    return total, sample["ntokens"], parts


# --------------------------------------------------------------------------------------------
# inference   (sequence_generator.py greedy path; speecht5.py:1188-1249)
# --------------------------------------------------------------------------------------------
@torch.no_grad()
def greedy_decode_asr(sd, cfg, wav, padding_mask, *, bos=2, eos=2, pad=1, max_len=200, forbid=()):
    """beam=1, ctc_weight=0 path of SequenceGenerator._generate (sequence_generator.py:207-520):
    the decoder is re-run on the whole prefix each step (mathematically identical to the reference's
    incremental KV-cache decoding)."""
    pre = speech_encoder_prenet(sd, cfg, wav, padding_mask, mask_indices=None)
    enc = encoder(sd, cfg, pre["x"], pre["padding_mask"])
    B = wav.shape[0]
    tokens = torch.full((B, 1), bos, dtype=torch.long)
    done = torch.zeros(B, dtype=torch.bool)
    for step in range(max_len):
        y, tmask = text_decoder_prenet(sd, cfg, tokens)
        y, _ = decoder(sd, cfg, y, tmask, enc)
        lp = F.log_softmax(text_decoder_postnet(sd, y[:, -1]).float(), dim=-1)
        lp[:, pad] = -math.inf
        for f in forbid:
            lp[:, f] = -math.inf
        if step == max_len - 1:
            lp[:, :eos] = -math.inf
            lp[:, eos + 1:] = -math.inf
        nxt = lp.argmax(-1)
        nxt = torch.where(done, torch.full_like(nxt, pad), nxt)
        tokens = torch.cat([tokens, nxt[:, None]], dim=1)
        done = done | nxt.eq(eos)
        if bool(done.all()):
            break
    return tokens[:, 1:]


@torch.no_grad()
def generate_speech(sd, cfg, src_tokens, spkembs, threshold=None):
    """T5TransformerModel.generate_speech (speecht5.py:1188-1249), text input, batch 1, eval mode with
    the Tacotron pre-net dropout disabled (p=0) so that the output is deterministic.  The reference
    reads kwargs["threshold"] for min/max length ratio as well (speecht5.py:1191-1201): defaults
    0.5 / 0.0 / 20.0 apply only when `threshold` is not passed."""
    minlenratio = threshold if threshold is not None else 0.0
    maxlenratio = threshold if threshold is not None else 20.0
    threshold = threshold if threshold is not None else 0.5
    x, pad = text_encoder_prenet(sd, cfg, src_tokens)
    enc = encoder(sd, cfg, x, pad)
    r, odim = cfg.reduction_factor, cfg.speech_odim
    maxlen = int(enc["encoder_out"][0].size(0) * maxlenratio / r)
    minlen = int(enc["encoder_out"][0].size(0) * minlenratio / r)
    ys = enc["encoder_out"][0].new_zeros(1, 1, odim)
    outs, probs, idx = [], [], 0
    while True:
        idx += 1
        d, _ = speech_decoder_prenet(sd, cfg, ys, None, spkembs)
        z, _ = decoder(sd, cfg, d, None, enc, alignment_layer=-1)
        z = z[0, -1]
        outs.append(_lin(sd, "speech_decoder_postnet.feat_out", z).view(r, odim))
        probs.append(torch.sigmoid(_lin(sd, "speech_decoder_postnet.prob_out", z)))
        ys = torch.cat((ys, outs[-1][-1].view(1, 1, odim)), dim=1)
        if int(sum(probs[-1] >= threshold)) > 0 or idx >= maxlen:
            if idx < minlen:
                continue
            mel = torch.cat(outs, dim=0).unsqueeze(0)  # [1, L, odim]
            _, after, _ = _postnet_only(sd, cfg, mel)
            return after.squeeze(0)


def _postnet_only(sd, cfg, before):
    x = before.transpose(1, 2)
    n = cfg.postnet_layers
    for i in range(n):
        p = f"speech_decoder_postnet.postnet.postnet.{i}."
        x = F.conv1d(x, sd[p + "0.weight"], padding=(cfg.postnet_filts - 1) // 2)
        x = F.batch_norm(x, sd[p + "1.running_mean"], sd[p + "1.running_var"], sd[p + "1.weight"], sd[p + "1.bias"],
                         training=False, eps=1e-5)
        if i < n - 1:
            x = torch.tanh(x)
    return before, before + x.transpose(1, 2), None


# --------------------------------------------------------------------------------------------
# configuration / synthetic weights (no fairseq): used by tests and bench
# --------------------------------------------------------------------------------------------
def base_config(**over):
    """Defaults of base_architecture + t5_transformer_base (speecht5.py:1252-1400) as a namespace."""
    c = dict(
        encoder_embed_dim=768, encoder_ffn_embed_dim=3072, encoder_layers=12, encoder_attention_heads=12,
        decoder_embed_dim=768, decoder_ffn_embed_dim=3072, decoder_layers=6, decoder_attention_heads=12,
        decoder_normalize_before=False, layer_norm_first=False, layer_norm_eps=1e-5, activation_fn="gelu",
        dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0, decoder_layerdrop=0.0,
        relative_position_embedding=True, encoder_max_relative_position=160, decoder_max_relative_position=160,
        extractor_mode="default", conv_feature_layers="[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2",
        conv_bias=False, feature_grad_mult=0.1, use_conv_pos=True, use_sinc_pos=True, conv_pos=128, conv_pos_groups=16,
        label_rates=50, sample_rate=16000, mask_prob=0.8, hubert_mask_length=10,
        enc_use_scaled_pos_enc=True, dec_use_scaled_pos_enc=True, no_scale_embedding=True,
        dprenet_layers=2, dprenet_units=256, dprenet_dropout_rate=0.0, postnet_layers=5, postnet_chans=256,
        postnet_filts=5, postnet_dropout_rate=0.0, reduction_factor=2, speech_odim=80, spk_embed_dim=512,
        spk_embed_integration_type="pre", use_codebook=True, latent_vars=100, latent_groups=2, latent_dim=0,
        codebook_prob=0.5, logit_temp=0.1, final_dim=256, untie_final_proj=True, share_input_output_embed=True,
        max_text_positions=450, max_speech_positions=4000,
    )
    c.update(over)
    return SimpleNamespace(**c)


# --------------------------------------------------------------------------------------------
# HiFi-GAN generator (not in the reference tree; spec = HF SpeechT5HifiGan.forward,
# transformers/models/speecht5/modeling_speecht5.py:3029-3066).  Pinned by tests/golden/tiny_hifigan.pt.
# --------------------------------------------------------------------------------------------
def hifigan(sd, cfg, spectrogram):
    """cfg: dict with upsample_rates, upsample_kernel_sizes, resblock_kernel_sizes, resblock_dilation_sizes,
    normalize_before.  spectrogram [B, L, 80] -> waveform [B, prod(rates) * L]."""
    x = spectrogram
    if cfg.get("normalize_before", True):
        x = (x - sd["mean"]) / sd["scale"]
    h = F.conv1d(x.transpose(2, 1), sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        h = F.leaky_relu(h, 0.1)
        h = F.conv_transpose1d(h, sd[f"upsampler.{i}.weight"], sd[f"upsampler.{i}.bias"], stride=u, padding=(k - u) // 2)
        acc = None
        for j, (rk, dil) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            p = f"resblocks.{i * nk + j}."
            r = h
            for q, d in enumerate(dil):
                t = F.conv1d(F.leaky_relu(r, 0.1), sd[p + f"convs1.{q}.weight"], sd[p + f"convs1.{q}.bias"], dilation=d,
                             padding=(rk * d - d) // 2)
                t = F.conv1d(F.leaky_relu(t, 0.1), sd[p + f"convs2.{q}.weight"], sd[p + f"convs2.{q}.bias"], padding=(rk - 1) // 2)
                r = t + r
            acc = r if acc is None else acc + r
        h = acc / nk
    h = F.leaky_relu(h)
    return torch.tanh(F.conv1d(h, sd["conv_post.weight"], sd["conv_post.bias"], padding=3)).squeeze(1)

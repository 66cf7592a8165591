"""SpeechEncoderPrenet / ConvFeatureExtractionModel mirrors of
SpeechT5/speecht5/models/modules/speech_encoder_prenet.py:58-374 on the HIP kernels."""
import contextlib
import math

import numpy as np
import torch
import torch.nn as nn

from .. import functional as Fn
from ..data_utils import compute_mask_indices, lengths_to_padding_mask
from .common import LayerNorm, SinusoidalPositionalEmbedding


class ConvFeatureExtractionModel(nn.Module):
    """Parameter container with the reference's names (conv_layers.{i}.0.weight, conv_layers.0.2.{weight,bias});
    forward = conv0+GroupNorm+GELU kernel followed by 6 implicit-GEMM conv layers, output channels-last."""

    def __init__(self, conv_layers, dropout=0.0, mode="default", conv_bias=False):
        super().__init__()
        assert mode in {"default", "layer_norm"}
        if dropout != 0.0:
            raise NotImplementedError("feature-extractor dropout != 0 is used by no SpeechT5 recipe and has no HIP path")
        if mode == "default" and conv_bias:
            raise NotImplementedError("extractor_mode=default with conv bias is used by no SpeechT5 recipe")
        self.mode = mode
        self.conv_layers = nn.ModuleList()
        self.conv_layers_infos = conv_layers
        in_d = 1
        for i, (dim, k, stride) in enumerate(conv_layers):
            conv = nn.Conv1d(in_d, dim, k, stride=stride, bias=conv_bias)
            nn.init.kaiming_normal_(conv.weight)
            if mode == "layer_norm":   # parameter names of the reference: conv_layers.{i}.0.*, conv_layers.{i}.2.1.*
                block = nn.Sequential(conv, nn.Dropout(p=dropout),
                                      nn.Sequential(nn.Identity(), nn.LayerNorm(dim, elementwise_affine=True), nn.Identity()), nn.GELU())
            elif i == 0:
                block = nn.Sequential(conv, nn.Dropout(p=dropout), nn.GroupNorm(dim, dim, affine=True), nn.GELU())
            else:
                block = nn.Sequential(conv, nn.Dropout(p=dropout), nn.GELU())
            self.conv_layers.append(block)
            in_d = dim

    def forward(self, x, grad_scale=1.0):
        """x: waveform [B, S] fp32 -> features [B, T, C] (channels-last, compute dtype)."""
        if self.mode == "layer_norm":
            params = [(blk[0].weight, blk[0].bias, blk[2][1].weight, blk[2][1].bias) for blk in self.conv_layers]
            return Fn.conv_feature_extractor_layer_norm(x.float(), self.conv_layers_infos, grad_scale, params)
        l0 = self.conv_layers[0]
        ws = [blk[0].weight for blk in list(self.conv_layers)[1:]]
        return Fn.conv_feature_extractor(x.float(), self.conv_layers_infos, grad_scale, l0[0].weight, l0[2].weight, l0[2].bias, ws)

    def get_out_seq_lens_tensor(self, in_seq_lens_tensor):
        out = in_seq_lens_tensor.clone()
        for _, k, s in self.conv_layers_infos:
            out = ((out.float() - (k - 1) - 1) / s + 1).floor().long()
        return out

    def get_out_seq_lens_nonmask_after_a_layer(self, in_seq_lens_tensor, i):
        out = in_seq_lens_tensor.clone()
        out = ((out.float() - (self.conv_layers_infos[i][1] - 1) - 1) / self.conv_layers_infos[i][-1] + 1).floor().long()
        return (~lengths_to_padding_mask(out)).float(), out


class SpeechEncoderPrenet(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.dropout = args.dropout
        self.padding_idx = 1
        self.freeze_encoder_updates = args.freeze_encoder_updates
        self.num_updates = 0
        assert args.encoder_speech_prenet == "conv", args.encoder_speech_prenet
        feature_enc_layers = eval(args.conv_feature_layers)  # noqa
        self.embed = feature_enc_layers[-1][0]
        self.feature_extractor = ConvFeatureExtractionModel(conv_layers=feature_enc_layers, dropout=0.0,
                                                            mode=args.extractor_mode, conv_bias=args.conv_bias)
        feature_ds_rate = np.prod([s for _, _, s in feature_enc_layers])
        self.feat2tar_ratio = args.label_rates * feature_ds_rate / args.sample_rate
        self.post_extract_proj = nn.Linear(self.embed, args.encoder_embed_dim) if self.embed != args.encoder_embed_dim else None
        self.use_conv_pos = args.use_conv_pos
        self.use_sinc_pos = args.use_sinc_pos
        self.use_abs_pos = getattr(args, "use_abs_pos", False)
        assert not self.use_abs_pos
        self.feature_grad_mult = args.feature_grad_mult
        self.conv_pos_groups = args.conv_pos_groups
        if self.use_conv_pos:
            self.layer_norm = LayerNorm(self.embed)
            pos_conv = nn.Conv1d(args.encoder_embed_dim, args.encoder_embed_dim, kernel_size=args.conv_pos,
                                 padding=args.conv_pos // 2, groups=args.conv_pos_groups)
            std = math.sqrt(4 / (args.conv_pos * args.encoder_embed_dim))
            nn.init.normal_(pos_conv.weight, mean=0, std=std)
            nn.init.constant_(pos_conv.bias, 0)
            pos_conv = nn.utils.weight_norm(pos_conv, name="weight", dim=2)
            self.pos_conv = nn.Sequential(pos_conv, nn.Identity(), nn.GELU())  # index 0 keeps "pos_conv.0.*" names
        if self.use_sinc_pos:
            self.embed_positions = SinusoidalPositionalEmbedding(args.encoder_embed_dim, self.padding_idx)
        self.mask_prob = args.mask_prob
        self.mask_selection = args.mask_selection
        self.mask_other = args.mask_other
        self.hubert_mask_length = args.hubert_mask_length
        self.no_mask_overlap = args.no_mask_overlap
        self.mask_min_space = args.mask_min_space
        self.mask_channel_prob = args.mask_channel_prob
        self.mask_channel_selection = args.mask_channel_selection
        self.mask_channel_other = args.mask_channel_other
        self.mask_channel_length = args.mask_channel_length
        self.no_mask_channel_overlap = args.no_mask_channel_overlap
        self.mask_channel_min_space = args.mask_channel_min_space
        self.mask_emb = nn.Parameter(torch.FloatTensor(args.encoder_embed_dim).uniform_())

    def forward(self, src_tokens, require_feat_pen=False, target_list=None, padding_mask=None, mask=True):
        ft = self.freeze_encoder_updates <= self.num_updates
        with torch.no_grad() if not ft else contextlib.ExitStack():
            return self._forward(src_tokens, require_feat_pen, target_list, padding_mask, mask)

    def _frames(self, n_samples):
        n = n_samples
        for _, k, s in self.feature_extractor.conv_layers_infos:
            n = (n - k) // s + 1
        return n

    def _forward(self, src_tokens, require_feat_pen=False, target_list=None, padding_mask=None, mask=True):
        # Everything that needs host data (frame padding mask, HuBERT span mask from the numpy RNG) depends only on the
        # INPUTS, so it is computed first, while the GPU is still idle / busy with the previous step: the single
        # device->host read of `padding_mask` happens here instead of in the middle of the forward pass.
        T = self._frames(src_tokens.size(1))
        if target_list is not None:
            targ_tsz = min([t.size(1) for t in target_list])
            if self.feat2tar_ratio * T > targ_tsz:
                T = int(targ_tsz / self.feat2tar_ratio)
        pm_host = getattr(padding_mask, "_st5_host", None)
        if pm_host is None:
            # one device->host read of the input padding mask, remembered on the tensor (a recorded / replayed step keeps the
            # same sample tensors; the collater could hand over the host copy it already has the same way)
            pm_host = padding_mask.cpu()
            padding_mask._st5_host = pm_host
        extra = pm_host.size(1) % T
        pmh = pm_host[:, :-extra] if extra > 0 else pm_host
        frame_pad_host = pmh.view(pmh.size(0), T, -1).all(-1)  # == forward_padding_mask() on the host
        pre_mask = None
        if mask and self.mask_prob > 0 and frame_pad_host is not None:
            pre_mask = self._sample_mask(src_tokens.size(0), T, frame_pad_host, src_tokens.device)
        if self.feature_grad_mult > 0:
            x = self.feature_extractor(src_tokens, grad_scale=self.feature_grad_mult)  # [B, T, C]
        else:
            with torch.no_grad():
                x = self.feature_extractor(src_tokens)
        if target_list is not None:
            x, target_list = self.forward_targets(x, target_list)
        # (the penalty reads the features through a "bypass" boundary: a path from the loss to the convolution stack that crosses none of
        #  the encoder's layer boundaries -- a backward cut at those boundaries must hold its gradient back until the stack's own phase)
        features_pen = Fn.mean_square(Fn.layer_boundary(x, self, "bypass")) if require_feat_pen else None
        x = self.layer_norm(x)
        encoder_padding_mask = self.forward_padding_mask(x, padding_mask)
        if self.post_extract_proj is not None:
            x = Fn.linear(x, self.post_extract_proj.weight, self.post_extract_proj.bias)
        x = Fn.dropout(x, self.dropout, self.training)
        if mask:
            if pre_mask is None and self.mask_prob > 0:
                # (target-trimmed path: the frame padding mask depends on the trimmed feature length)
                assert not Fn.static_shapes(), "graph capture needs the frame padding mask to follow from the input lengths"
                fp_host = encoder_padding_mask.cpu()
                pre_mask = self._sample_mask(x.size(0), x.size(1), fp_host, x.device)
            x, mask_indices = self.apply_hubert_mask(x, encoder_padding_mask, pre_mask)
        else:
            mask_indices = None
        if self.use_conv_pos:
            conv = self.pos_conv[0]
            # weight_norm(dim=2): w = g * v / ||v||, the norm over (out, in) per tap.  Written as a reduction + one
            # broadcast multiply: torch._weight_norm's dim=2 kernels need 0.75 ms fwd+bwd for this 4.7 M-element
            # parameter on MI355X, this form 0.17 ms
            v = conv.weight_v
            w = v * (conv.weight_g / torch.linalg.vector_norm(v, dim=(0, 1), keepdim=True))
            x = Fn.pos_conv(x, w, conv.bias, self.conv_pos_groups)
        if self.use_sinc_pos:
            positions = self.embed_positions.positions(~encoder_padding_mask)
            table = self.embed_positions.table(x.shape[1] + self.padding_idx + 2, x.device)
            x = Fn.add_table_rows(x, table, positions.reshape(-1))
        if require_feat_pen:
            return (x, features_pen, mask_indices, target_list), encoder_padding_mask
        return x, encoder_padding_mask

    def forward_targets(self, features, target_list):
        """features are channels-last here ([B,T,C]); same trimming rule as the reference (:206-217)."""
        feat_tsz = features.size(1)
        targ_tsz = min([t.size(1) for t in target_list])
        if self.feat2tar_ratio * feat_tsz > targ_tsz:
            feat_tsz = int(targ_tsz / self.feat2tar_ratio)
            features = features[:, :feat_tsz].contiguous()
        ratio = self.feat2tar_ratio
        target_inds = Fn.stage_host(lambda: (torch.arange(feat_tsz).float() * ratio).long(), features.device)
        target_list = [t.index_select(1, target_inds) for t in target_list]
        return features, target_list

    def forward_padding_mask(self, features, padding_mask):
        extra = padding_mask.size(1) % features.size(1)
        if extra > 0:
            padding_mask = padding_mask[:, :-extra]
        padding_mask = padding_mask.view(padding_mask.size(0), features.size(1), -1)
        return padding_mask.all(-1)

    def get_src_lengths(self, src_lengths):
        return self.feature_extractor.get_out_seq_lens_tensor(src_lengths)

    def _sample_mask(self, B, T, frame_pad_host, device):
        """HuBERT span mask from the numpy RNG (speech_encoder_prenet.py:237-247) -> (device bool [B,T], host copies)."""
        box = {}

        def draw():
            m = compute_mask_indices((B, T), frame_pad_host, self.mask_prob, self.hubert_mask_length, self.mask_selection,
                                     self.mask_other, min_masks=2, no_overlap=self.no_mask_overlap, min_space=self.mask_min_space)
            box["mt"] = torch.from_numpy(m)
            return box["mt"]
        dev = Fn.stage_host(draw, device)
        if "mt" in box:   # host copies: the NCE head derives its gather indices without a device sync (eager mode)
            dev._st5_host = (box["mt"], frame_pad_host)
        return dev

    def apply_hubert_mask(self, x, padding_mask, pre_mask=None):
        B, T, C = x.shape
        mask_indices = None
        if self.mask_prob > 0:
            mask_indices = pre_mask if pre_mask is not None else self._sample_mask(B, T, padding_mask.cpu(), x.device)
            x = Fn.masked_fill_rows(x, mask_indices.reshape(-1), self.mask_emb)
        if self.mask_channel_prob > 0:
            # same numpy draw as the reference (speech_encoder_prenet.py:253-263), taken right after the time mask
            mc = compute_mask_indices((B, C), None, self.mask_channel_prob, self.mask_channel_length, self.mask_channel_selection,
                                      self.mask_channel_other, no_overlap=self.no_mask_channel_overlap,
                                      min_space=self.mask_channel_min_space)
            x = Fn.mask_channels(x, torch.from_numpy(mc).to(x.device, non_blocking=True))
        return x, mask_indices

    def set_num_updates(self, num_updates):
        self.num_updates = num_updates

"""SpeechDecoderPostnet mirror of SpeechT5/speecht5/models/modules/speech_decoder_postnet.py:17-76
(feat_out / prob_out heads + espnet Tacotron Postnet: 5 x [Conv1d k5, BatchNorm1d, tanh, dropout])."""
import contextlib

import torch
import torch.nn as nn

from .. import functional as Fn


class Postnet(nn.Module):
    """Parameter layout of espnet's Postnet (postnet.{i}.0 = Conv1d, postnet.{i}.1 = BatchNorm1d)."""

    def __init__(self, idim, odim, n_layers=5, n_chans=512, n_filts=5, dropout_rate=0.5, use_batch_norm=True):
        super().__init__()
        assert use_batch_norm
        self.dropout_rate = dropout_rate
        self.postnet = nn.ModuleList()
        for layer in range(n_layers):
            ichans = odim if layer == 0 else n_chans
            ochans = odim if layer == n_layers - 1 else n_chans
            mods = [nn.Conv1d(ichans, ochans, n_filts, stride=1, padding=(n_filts - 1) // 2, bias=False), nn.BatchNorm1d(ochans)]
            if layer < n_layers - 1:
                mods.append(nn.Tanh())
            mods.append(nn.Dropout(dropout_rate))
            self.postnet.append(nn.Sequential(*mods))

    def forward(self, x):
        """x = `before` [B, L, odim] channels-last (compute dtype) -> `after` = before + postnet(before), fp32 [B, L, odim].
        Convolutions on the implicit-GEMM kernel (fp32 accumulators handed over), BatchNorm (batch statistics, running
        statistics update) + tanh + dropout + the final residual on st5_batchnorm_act_* (functional.PostnetFunction)."""
        training = self.postnet[0][1].training
        return Fn.postnet(x, self.postnet, training, self.dropout_rate if self.training else 0.0)


class SpeechDecoderPostnet(nn.Module):
    def __init__(self, odim, args):
        super().__init__()
        self.feat_out = nn.Linear(args.decoder_embed_dim, odim * args.reduction_factor)
        self.prob_out = nn.Linear(args.decoder_embed_dim, args.reduction_factor)
        self.postnet = None if args.postnet_layers == 0 else Postnet(
            idim=0, odim=odim, n_layers=args.postnet_layers, n_chans=args.postnet_chans, n_filts=args.postnet_filts,
            use_batch_norm=args.use_batch_norm, dropout_rate=args.postnet_dropout_rate)
        self.odim = odim
        self.num_updates = 0
        self.freeze_decoder_updates = args.freeze_decoder_updates

    def forward(self, zs):
        ft = self.freeze_decoder_updates <= self.num_updates
        with torch.no_grad() if not ft else contextlib.ExitStack():
            zs = Fn.as_compute(zs)
            B = zs.size(0)
            # one fused GEMM for both heads: [feat_out ; prob_out]
            both = Fn.linear(zs, [self.feat_out.weight, self.prob_out.weight], [self.feat_out.bias, self.prob_out.bias])
            nf = self.feat_out.weight.shape[0]
            before = both[..., :nf].reshape(B, -1, self.odim)
            logits = both[..., nf:].reshape(B, -1)
            if self.postnet is None:
                after = Fn.as_float(before.contiguous())
            else:
                after = self.postnet(before.contiguous())   # fp32: before + post-net residual (fused into the last BatchNorm kernel)
        return Fn.as_float(before.contiguous()), after, Fn.as_float(logits.contiguous())

    def set_num_updates(self, num_updates):
        self.num_updates = num_updates

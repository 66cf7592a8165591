"""SpeechDecoderPostnet mirror of SpeechT5/speecht5/models/modules/speech_decoder_postnet.py:17-76
(feat_out / prob_out heads + espnet Tacotron Postnet: 5 x [Conv1d k5, BatchNorm1d, tanh, dropout])."""
import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F

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
        """x [B, L, odim] channels-last (compute dtype).  Convolutions run on the implicit-GEMM kernel; the
        BatchNorm statistics/affine and tanh on [B*L, C] (<1 % of the step) are still torch ops -- see DESIGN.md."""
        n = len(self.postnet)
        for i, blk in enumerate(self.postnet):
            x = Fn.conv1d_same(x, blk[0].weight)
            bn = blk[1]
            B, L, C = x.shape
            y = F.batch_norm(x.reshape(B * L, C).float(), bn.running_mean, bn.running_var, bn.weight, bn.bias,
                             bn.training, bn.momentum, bn.eps)
            if bn.training and bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
            if i < n - 1:
                y = torch.tanh(y)
            x = y.to(x.dtype).view(B, L, C)
            x = Fn.dropout(x, self.dropout_rate, self.training)
        return x


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
                after = before
            else:
                after = Fn.add(before.contiguous(), self.postnet(before.contiguous()))
        return Fn.as_float(before.contiguous()), Fn.as_float(after), Fn.as_float(logits.contiguous())

    def set_num_updates(self, num_updates):
        self.num_updates = num_updates

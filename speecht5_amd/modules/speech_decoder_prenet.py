"""SpeechDecoderPrenet mirror of SpeechT5/speecht5/models/modules/speech_decoder_prenet.py:21-110
(espnet Tacotron Prenet + Linear + ScaledPositionalEncoding + speaker-embedding integration)."""
import contextlib

import torch
import torch.nn as nn

from .. import functional as Fn
from .common import ScaledPEAdd, ScaledPositionalEncoding


class TacotronDecoderPrenet(nn.Module):
    """espnet tacotron2 Prenet: n x (Linear -> ReLU -> dropout), the dropout ALWAYS active."""

    def __init__(self, idim, n_layers=2, n_units=256, dropout_rate=0.5):
        super().__init__()
        self.dropout_rate = dropout_rate
        self.prenet = nn.ModuleList()
        for layer in range(n_layers):
            self.prenet += [nn.Sequential(nn.Linear(idim if layer == 0 else n_units, n_units), nn.ReLU())]

    def forward(self, x):
        for blk in self.prenet:
            x = Fn.linear(x, blk[0].weight, blk[0].bias, act=Fn.ACT_RELU, dropout_p=self.dropout_rate)
        return x


class SpeechDecoderPrenet(nn.Module):
    def __init__(self, odim, args):
        super().__init__()
        assert args.dprenet_layers != 0 and args.dec_use_scaled_pos_enc
        decoder_input_layer = nn.Sequential(
            TacotronDecoderPrenet(idim=odim, n_layers=args.dprenet_layers, n_units=args.dprenet_units,
                                  dropout_rate=args.dprenet_dropout_rate),
            nn.Linear(args.dprenet_units, args.decoder_embed_dim),
        )
        self.decoder_prenet = nn.Sequential(
            decoder_input_layer,
            ScaledPositionalEncoding(args.decoder_embed_dim, args.transformer_dec_positional_dropout_rate,
                                     max_len=args.max_speech_positions),
        )
        if args.spk_embed_integration_type == "pre":
            self.spkembs_layer = nn.Sequential(nn.Linear(args.spk_embed_dim + args.decoder_embed_dim, args.decoder_embed_dim), nn.ReLU())
        self.num_updates = 0
        self.freeze_decoder_updates = args.freeze_decoder_updates

    def forward(self, prev_output_tokens, tgt_lengths_in=None, spkembs=None):
        ft = self.freeze_decoder_updates <= self.num_updates
        with torch.no_grad() if not ft else contextlib.ExitStack():
            inl, spe = self.decoder_prenet[0], self.decoder_prenet[1]
            x = Fn.as_compute(prev_output_tokens.contiguous())
            x = inl[0](x)
            x = Fn.linear(x, inl[1].weight, inl[1].bias)
            x = ScaledPEAdd.apply(x, spe.alpha, spe.pe(x.shape[1], x.device))
            x = Fn.dropout(x, spe.dropout_rate, self.training)
            if spkembs is not None:
                # [x ; normalize(spk)] -> Linear -> ReLU (:81-83); the concat is assembled by torch (glue)
                s = torch.nn.functional.normalize(spkembs.float()).unsqueeze(1).expand(-1, x.size(1), -1)
                cat = torch.cat([x, Fn.as_compute(s.contiguous())], dim=-1)
                lin = self.spkembs_layer[0]
                x = Fn.linear(cat, lin.weight, lin.bias, act=Fn.ACT_RELU)
            tgt_frames_mask = None
            if tgt_lengths_in is not None:
                lens = torch.as_tensor(tgt_lengths_in, device=x.device)
                # the batch is padded to its longest target, so max(lens) == T: no host read needed
                tgt_frames_mask = torch.arange(x.size(1), device=x.device)[None, :] >= lens[:, None]
            return x, tgt_frames_mask

    def set_num_updates(self, num_updates):
        self.num_updates = num_updates

"""TextDecoderPostnet mirror of SpeechT5/speecht5/models/modules/text_decoder_postnet.py:19-93."""
import contextlib

import torch
import torch.nn as nn

from .. import functional as Fn


class TextDecoderPostnet(nn.Module):
    def __init__(self, embed_tokens, dictionary, args, output_projection=None):
        super().__init__()
        self.output_embed_dim = args.decoder_output_dim
        self.output_projection = output_projection
        self.adaptive_softmax = None
        assert args.adaptive_softmax_cutoff is None
        self.share_input_output_embed = args.share_input_output_embed
        if self.output_projection is None:
            if self.share_input_output_embed:
                self.output_projection = nn.Linear(embed_tokens.weight.shape[1], embed_tokens.weight.shape[0], bias=False)
                self.output_projection.weight = embed_tokens.weight
            else:
                self.output_projection = nn.Linear(self.output_embed_dim, len(dictionary), bias=False)
                nn.init.normal_(self.output_projection.weight, mean=0, std=self.output_embed_dim ** -0.5)
        self.freeze_decoder_updates = args.freeze_decoder_updates
        self.num_updates = 0

    def output_layer(self, features):
        return Fn.as_float(Fn.linear(Fn.as_compute(features), self.output_projection.weight, None))

    def forward(self, x):
        ft = self.freeze_decoder_updates <= self.num_updates
        with torch.no_grad() if not ft else contextlib.ExitStack():
            return self.output_layer(x)

    def set_num_updates(self, num_updates):
        self.num_updates = num_updates

"""TextDecoderPrenet mirror of SpeechT5/speecht5/models/modules/text_decoder_prenet.py:23-128."""
import contextlib
import math

import torch
import torch.nn as nn

from .. import functional as Fn
from .common import SinusoidalPositionalEmbedding


class TextDecoderPrenet(nn.Module):
    def __init__(self, embed_tokens, args):
        super().__init__()
        self.dropout = args.dropout
        self.decoder_layerdrop = args.decoder_layerdrop
        self.num_updates = 0
        embed_dim = args.decoder_embed_dim
        assert embed_tokens.embedding_dim == embed_dim and not getattr(args, "layernorm_embedding", False)
        assert not args.decoder_learned_pos and not args.no_token_positional_embeddings
        self.embed_dim = embed_dim
        self.output_embed_dim = args.decoder_output_dim
        self.padding_idx = embed_tokens.padding_idx
        self.embed_tokens = embed_tokens
        self.embed_scale = 1.0 if args.no_scale_embedding else math.sqrt(embed_dim)
        self.embed_positions = SinusoidalPositionalEmbedding(embed_dim, self.padding_idx)
        self.freeze_decoder_updates = args.freeze_decoder_updates

    def forward(self, prev_output_tokens, incremental_state=None):
        ft = self.freeze_decoder_updates <= self.num_updates
        with torch.no_grad() if not ft else contextlib.ExitStack():
            return self._forward(prev_output_tokens, incremental_state)

    def _forward(self, prev_output_tokens, incremental_state=None):
        """With an incremental state (inference) only the newest token is embedded (:102-105 of the reference); the padding
        mask still covers the whole prefix, it is the key mask of the cached self-attention."""
        pad = prev_output_tokens.eq(self.padding_idx)
        x_mask = pad  # the reference returns None when nothing is padded (a host sync); an all-False mask is equivalent
        positions = self.embed_positions.positions(~pad)
        table = self.embed_positions.table(prev_output_tokens.shape[1] + self.padding_idx + 2, prev_output_tokens.device)
        if incremental_state is not None and not self.training:
            prev_output_tokens, positions = prev_output_tokens[:, -1:], positions[:, -1:]
        x = Fn.embed_rows(self.embed_tokens.weight, prev_output_tokens, pos=table, pidx=positions, emb_scale=self.embed_scale)
        x = Fn.dropout(x, self.dropout, self.training)
        return x, x_mask, incremental_state

    def set_num_updates(self, num_updates):
        self.num_updates = num_updates

"""TextEncoderPrenet mirror of SpeechT5/speecht5/models/modules/text_encoder_prenet.py:16-45."""
import torch
import torch.nn as nn

from .. import functional as Fn
from .common import ScaledPEAdd, ScaledPositionalEncoding


class TextEncoderPrenet(nn.Module):
    def __init__(self, embed_tokens, args):
        super().__init__()
        self.padding_idx = embed_tokens.padding_idx
        assert args.enc_use_scaled_pos_enc, "PositionalEncoding (unscaled) is not used by the SpeechT5 recipes"
        self.encoder_prenet = nn.Sequential(
            embed_tokens,
            ScaledPositionalEncoding(args.encoder_embed_dim, args.transformer_enc_positional_dropout_rate,
                                     max_len=args.max_text_positions),
        )

    def forward(self, src_tokens):
        emb, spe = self.encoder_prenet[0], self.encoder_prenet[1]
        x = Fn.embed_rows(emb.weight, src_tokens)  # [B,T,d]; the pad row of the table is zero as in nn.Embedding
        x = ScaledPEAdd.apply(x, spe.alpha, spe.pe(src_tokens.shape[1], x.device))
        x = Fn.dropout(x, spe.dropout_rate, self.training)
        return x, src_tokens.eq(self.padding_idx)

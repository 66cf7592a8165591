"""`transformer_lm_t5` (SpeechT5/speecht5/models/t5_transformer_lm.py:18-27): the architecture preset of the shallow-fusion
language model the ASR decoding recipe loads through fairseq's own `transformer_lm` model -- a registration only, no model code.
With fairseq present it is registered on fairseq's `transformer_lm`; without, the preset is kept in the local registry."""
from .fairseq_compat import HAVE_FAIRSEQ, register_model_architecture

_PRESET = (("decoder_embed_dim", 1280), ("decoder_ffn_embed_dim", 6144), ("decoder_layers", 20), ("decoder_attention_heads", 16),
           ("dropout", 0.1), ("attention_dropout", 0.1), ("activation_fn", "gelu"))


@register_model_architecture("transformer_lm", "transformer_lm_t5")
def transformer_lm_t5(args):
    for name, value in _PRESET:
        if not hasattr(args, name):
            setattr(args, name, value)
    if HAVE_FAIRSEQ:   # the remaining defaults are fairseq's own
        from fairseq.models.transformer_lm import base_lm_architecture
        base_lm_architecture(args)

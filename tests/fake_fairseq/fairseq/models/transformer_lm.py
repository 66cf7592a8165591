from . import FairseqLanguageModel, register_model


@register_model("transformer_lm")
class TransformerLanguageModel(FairseqLanguageModel):
    pass


def base_lm_architecture(args):
    args.decoder_input_dim = getattr(args, "decoder_input_dim", args.decoder_embed_dim)

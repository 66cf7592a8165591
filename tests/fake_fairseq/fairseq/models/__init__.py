import torch.nn as nn

MODEL_REGISTRY, ARCH_MODEL_REGISTRY, ARCH_CONFIG_REGISTRY = {}, {}, {}


class BaseFairseqModel(nn.Module):
    pass


class FairseqEncoder(nn.Module):
    def __init__(self, dictionary):
        super().__init__()
        self.dictionary = dictionary


class FairseqDecoder(nn.Module):
    def __init__(self, dictionary):
        super().__init__()
        self.dictionary = dictionary


class FairseqIncrementalDecoder(FairseqDecoder):
    pass


class FairseqEncoderDecoderModel(BaseFairseqModel):
    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder, self.decoder = encoder, decoder


class FairseqLanguageModel(BaseFairseqModel):
    pass


def register_model(name, dataclass=None):
    def wrap(cls):
        if name in MODEL_REGISTRY:
            raise ValueError(f"Cannot register duplicate model ({name})")
        if not issubclass(cls, BaseFairseqModel):
            raise ValueError(f"Model ({name}: {cls.__name__}) must extend BaseFairseqModel")
        MODEL_REGISTRY[name] = cls
        return cls
    return wrap


def register_model_architecture(model_name, arch_name):
    def wrap(fn):
        if model_name not in MODEL_REGISTRY:
            raise ValueError(f"Cannot register model architecture for unknown model type ({model_name})")
        if arch_name in ARCH_MODEL_REGISTRY:
            raise ValueError(f"Cannot register duplicate model architecture ({arch_name})")
        if not callable(fn):
            raise ValueError(f"Model architecture must be callable ({arch_name})")
        ARCH_MODEL_REGISTRY[arch_name] = MODEL_REGISTRY[model_name]
        ARCH_CONFIG_REGISTRY[arch_name] = fn
        return fn
    return wrap


from . import transformer_lm  # noqa: E402,F401  (fairseq imports its model files, which register themselves)

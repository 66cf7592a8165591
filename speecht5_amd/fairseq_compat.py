"""Minimal stand-ins for the fairseq base classes / registries the SpeechT5 plug-in hooks into.

When fairseq is importable the real decorators and base classes are used, so `--user-dir speecht5_amd`
registers task `speecht5`, model `t5_transformer` (+ archs) and criterion `speecht5` exactly like the
reference package does (SpeechT5/speecht5/__init__.py:1).  Without fairseq (this image) the same
names are kept in local registries so that tests and bench.py can build everything by name."""
import uuid

import torch.nn as nn

try:  # pragma: no cover - fairseq is not installed in the build image
    from fairseq.models import (FairseqEncoder, FairseqEncoderDecoderModel, FairseqIncrementalDecoder, register_model,
                                register_model_architecture)
    from fairseq.tasks import register_task
    from fairseq.criterions import register_criterion
    HAVE_FAIRSEQ = True
except Exception:  # noqa: BLE001
    HAVE_FAIRSEQ = False
    MODEL_REGISTRY, ARCH_REGISTRY, TASK_REGISTRY, CRITERION_REGISTRY = {}, {}, {}, {}

    class FairseqEncoder(nn.Module):
        def __init__(self, dictionary):
            super().__init__()
            self.dictionary = dictionary

    class FairseqIncrementalDecoder(nn.Module):
        def __init__(self, dictionary):
            super().__init__()
            self.dictionary = dictionary

    class FairseqEncoderDecoderModel(nn.Module):
        def __init__(self, encoder, decoder):
            super().__init__()
            self.encoder = encoder
            self.decoder = decoder

    def register_model(name, dataclass=None):
        def wrap(cls):
            MODEL_REGISTRY[name] = cls
            return cls
        return wrap

    def register_model_architecture(model_name, arch_name):
        def wrap(fn):
            ARCH_REGISTRY[arch_name] = fn
            return fn
        return wrap

    def register_task(name, dataclass=None):
        def wrap(cls):
            TASK_REGISTRY[name] = cls
            return cls
        return wrap

    def register_criterion(name, dataclass=None):
        def wrap(cls):
            CRITERION_REGISTRY[name] = cls
            return cls
        return wrap


class IncrementalState:
    """uuid-keyed per-module incremental state (fairseq/incremental_decoding_utils.py semantics)."""

    def init_incremental_state(self):
        self._incremental_state_id = str(uuid.uuid4())

    def _full_key(self, key):
        return f"{self._incremental_state_id}.{key}"

    def get_incremental_state(self, incremental_state, key):
        if incremental_state is None:
            return None
        return incremental_state.get(self._full_key(key))

    def set_incremental_state(self, incremental_state, key, value):
        if incremental_state is not None:
            incremental_state[self._full_key(key)] = value
        return incremental_state

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
    from fairseq.criterions import FairseqCriterion, register_criterion
    try:
        from fairseq.tasks import LegacyFairseqTask
    except ImportError:   # fairseq < 0.10: FairseqTask is the argparse-style base
        from fairseq.tasks import FairseqTask as LegacyFairseqTask
    try:
        from fairseq.logging import metrics
    except ImportError:
        from fairseq import metrics
    from fairseq import utils as _fs_utils
    try:
        from fairseq.logging.meters import safe_round
    except ImportError:
        def safe_round(number, ndigits):
            return round(float(number), ndigits)
    HAVE_FAIRSEQ = True

    def utils_item(x):
        return _fs_utils.item(x)

    def get_perplexity(loss, round=2, base=2):
        return _fs_utils.get_perplexity(loss, round, base)
except ImportError:
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

    class FairseqCriterion(nn.Module):
        """fairseq/criterions/fairseq_criterion.py: holds the task, pad index from its target dictionary."""

        def __init__(self, task):
            super().__init__()
            self.task = task
            tgt = getattr(task, "target_dictionary", None)
            self.padding_idx = tgt.pad() if tgt is not None else -100

        @staticmethod
        def logging_outputs_can_be_summed():
            return False

    class LegacyFairseqTask:
        """fairseq/tasks/fairseq_task.py LegacyFairseqTask: argparse-namespace task, `datasets` / `dataset_to_epoch_iter` dicts."""

        def __init__(self, args):
            self.args = args
            self.datasets = {}
            self.dataset_to_epoch_iter = {}

        def dataset(self, split):
            if split not in self.datasets:
                raise KeyError("Dataset not loaded: " + split)
            return self.datasets[split]

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

    class _Meter:
        def __init__(self):
            self.sum, self.count, self.val, self.round = 0.0, 0.0, 0.0, None

        @property
        def avg(self):
            return self.sum / self.count if self.count > 0 else self.val

    class _Metrics:
        """Stand-in for fairseq.logging.metrics with the two calls reduce_metrics makes: log_scalar(key, value, weight, priority,
        round) accumulates a weighted average meter, log_derived(key, fn, priority) registers fn(meters).  `recorded` keeps the
        call sequence (tests compare it against the reference's)."""

        def __init__(self):
            self.reset()

        def reset(self):
            self.meters, self.derived, self.recorded = {}, {}, []

        def log_scalar(self, key, value, weight=1, priority=10, round=None):
            m = self.meters.setdefault(key, _Meter())
            v = float(value)
            m.val, m.round = v, round
            m.sum += v * weight
            m.count += weight
            self.recorded.append(("scalar", key, v, float(weight), priority, round))

        def log_derived(self, key, fn, priority=20):
            self.derived[key] = fn
            self.recorded.append(("derived", key, priority))

        def get_smoothed_values(self):
            out = {k: (m.avg if m.round is None else __import__("builtins").round(m.avg, m.round)) for k, m in self.meters.items()}
            for k, fn in self.derived.items():
                out[k] = fn(self.meters)
            return out

    metrics = _Metrics()

    def utils_item(x):
        return x.item() if hasattr(x, "item") else x

    def get_perplexity(loss, round=2, base=2):
        """fairseq.utils.get_perplexity."""
        import builtins
        if loss is None:
            return 0.0
        try:
            return builtins.round(base ** float(loss), round)
        except OverflowError:
            return float("inf")

    def safe_round(number, ndigits):
        """fairseq.logging.meters.safe_round."""
        if hasattr(number, "item"):
            number = number.item()
        return round(number, ndigits)


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

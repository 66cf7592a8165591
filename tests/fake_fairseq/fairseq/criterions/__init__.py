import torch.nn as nn

CRITERION_REGISTRY = {}


class FairseqCriterion(nn.Module):
    def __init__(self, task):
        super().__init__()
        self.task = task
        tgt = getattr(task, "target_dictionary", None)
        self.padding_idx = tgt.pad() if tgt is not None else -100


def register_criterion(name, dataclass=None):
    def wrap(cls):
        if name in CRITERION_REGISTRY:
            raise ValueError(f"Cannot register duplicate criterion ({name})")
        if not issubclass(cls, FairseqCriterion):
            raise ValueError(f"criterion must extend FairseqCriterion ({name}: {cls.__name__})")
        CRITERION_REGISTRY[name] = cls
        return cls
    return wrap

TASK_REGISTRY, TASK_CLASS_NAMES = {}, set()


class FairseqTask:
    def __init__(self, cfg=None, **kwargs):
        self.cfg = cfg
        self.datasets, self.dataset_to_epoch_iter = {}, {}


class LegacyFairseqTask(FairseqTask):
    def __init__(self, args):
        super().__init__(None)
        self.args = args


def register_task(name, dataclass=None):
    def wrap(cls):
        if name in TASK_REGISTRY:
            raise ValueError(f"Cannot register duplicate task ({name})")
        if not issubclass(cls, FairseqTask):
            raise ValueError(f"Task ({name}: {cls.__name__}) must extend FairseqTask")
        if cls.__name__ in TASK_CLASS_NAMES:
            raise ValueError(f"Cannot register task with duplicate class name ({cls.__name__})")
        TASK_REGISTRY[name] = cls
        TASK_CLASS_NAMES.add(cls.__name__)
        return cls
    return wrap

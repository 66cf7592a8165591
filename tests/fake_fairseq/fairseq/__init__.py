"""A 60-line stand-in for the fairseq package, ONLY for tests/test_fairseq_surface.py: its registries enforce the same class
checks the real ones do (fairseq/tasks/__init__.py `register_task`: "must extend FairseqTask"; fairseq/criterions:
`register_criterion` via registry.setup_registry(base_class=FairseqCriterion): "must extend FairseqCriterion";
fairseq/models/__init__.py `register_model`: "must extend BaseFairseqModel"; duplicate names raise)."""
from . import utils  # noqa: F401

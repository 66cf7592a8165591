from . import metrics, meters  # noqa: F401

"""speecht5_amd: the SpeechT5 forward/backward hot path as hand-written gfx950 (MI355X) HIP kernels behind the
reference's fairseq plug-in surface.  `--user-dir speecht5_amd` registers task `speecht5`, model `t5_transformer`
(+ archs `t5_transformer_base/_large/_base_asr`), arch `transformer_lm_t5` and criterion `speecht5` like SpeechT5/speecht5/__init__.py:1."""
import os as _os

# kernel arguments in device memory (lower launch latency; read by the HIP runtime when it initialises, i.e. at the first
# device call -- harmless if the process already did that)
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from . import fairseq_compat  # noqa: F401


def _register():
    from . import criterions, speecht5, t5_transformer_lm, task  # noqa: F401


# Registration is the whole point of importing this package (`--user-dir`): a failure here must not be swallowed -- a job
# would otherwise start with the task / arch / criterion names silently missing.
_register()

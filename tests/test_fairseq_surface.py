"""The plug-in surface against fairseq's own rules.

  * `reduce_metrics` (the trainer-side aggregation hook, speecht5_criterion.py:122-437): the sequence of metric calls -- name,
    value, weight, priority, rounding -- and the derived meters must equal what the VERBATIM reference function produces on the
    same two-rank logging outputs (tests/golden/reduce_metrics.json, oracle/make_golden_metrics.py).
  * registration under a fairseq whose registries enforce the real class checks (tests/fake_fairseq: `register_task` requires a
    FairseqTask, `register_criterion` a FairseqCriterion, `register_model` a BaseFairseqModel, duplicate names raise, an
    architecture needs its model registered): importing speecht5_amd must register task `speecht5`, model `t5_transformer` with
    its four architectures, `transformer_lm_t5` on fairseq's `transformer_lm`, criterion `speecht5`; `load_dataset` must reach
    the reference plug-in's data plane without registering the task name twice.  Runs in a subprocess (own sys.modules)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "reduce_metrics.json")


def _close(a, b):
    return a == b or (isinstance(a, float) and isinstance(b, float) and abs(a - b) <= 1e-12 * max(1.0, abs(b)))


def test_reduce_metrics_matches_the_verbatim_reference():
    from speecht5_amd import fairseq_compat as fc
    from speecht5_amd.criterions import SpeechT5Criterion
    gold = json.load(open(G))
    assert not fc.HAVE_FAIRSEQ
    fc.metrics.reset()
    SpeechT5Criterion.reduce_metrics(gold["logging_outputs"])
    got = [list(c) for c in fc.metrics.recorded]
    assert len(got) == len(gold["calls"]), (len(got), len(gold["calls"]))
    for g, r in zip(got, gold["calls"]):
        assert len(g) == len(r) and all(_close(x, y) for x, y in zip(g, r)), (g, r)
    derived = {k: fn(fc.metrics.meters) for k, fn in fc.metrics.derived.items()}
    assert set(derived) == set(gold["derived"])
    for k, v in gold["derived"].items():
        assert _close(float(derived[k]), float(v)), (k, derived[k], v)


_CHILD = r'''
import sys, json
from argparse import Namespace
import fairseq
from fairseq import models, tasks, criterions
import speecht5_amd
from speecht5_amd import fairseq_compat as fc
assert fc.HAVE_FAIRSEQ
assert set(models.ARCH_MODEL_REGISTRY) >= {"t5_transformer", "t5_transformer_base", "t5_transformer_large", "t5_transformer_base_asr", "transformer_lm_t5"}, sorted(models.ARCH_MODEL_REGISTRY)
assert models.ARCH_MODEL_REGISTRY["transformer_lm_t5"] is models.MODEL_REGISTRY["transformer_lm"]
assert issubclass(models.MODEL_REGISTRY["t5_transformer"], models.BaseFairseqModel)
T, C = tasks.TASK_REGISTRY["speecht5"], criterions.CRITERION_REGISTRY["speecht5"]
assert issubclass(T, tasks.LegacyFairseqTask) and issubclass(C, criterions.FairseqCriterion)
a = Namespace(); models.ARCH_CONFIG_REGISTRY["transformer_lm_t5"](a)
assert (a.decoder_embed_dim, a.decoder_ffn_embed_dim, a.decoder_layers, a.decoder_attention_heads, a.activation_fn, a.decoder_input_dim) == (1280, 6144, 20, 16, "gelu", 1280)
task = T.synthetic(Namespace(data="/some/manifests", t5_task="pretrain"))
assert isinstance(task, tasks.FairseqTask) and task.datasets == {}
crit = task.build_criterion(Namespace())
assert isinstance(crit, criterions.FairseqCriterion) and crit.task is task and crit.padding_idx == 1
# load_dataset -> the reference plug-in's data plane, on this task object, without a duplicate registration
task.load_dataset("train", epoch=3)
assert task.datasets["train"] == ("reference data plane", "train", 3, "/some/manifests", len(task.dicts["text"])), task.datasets
assert tasks.TASK_REGISTRY["speecht5"] is T
# reduce_metrics goes to fairseq's metrics module
from fairseq.logging import metrics
gold = json.load(open(sys.argv[1]))
C.reduce_metrics(gold["logging_outputs"])
assert len(metrics.CALLS) == len(gold["calls"]) and [c[:2] for c in metrics.CALLS] == [c[:2] for c in gold["calls"]]
print("FAIRSEQ-SURFACE-OK")
'''


def test_registers_under_a_fairseq_that_enforces_its_base_classes():
    fake = os.path.join(ROOT, "tests", "fake_fairseq")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([fake, ROOT, os.environ.get("PYTHONPATH", "")]))
    out = subprocess.run([sys.executable, "-c", _CHILD, G], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "FAIRSEQ-SURFACE-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]

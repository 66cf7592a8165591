"""TEST INFRASTRUCTURE (build container only: needs /root/reference).  Runs the VERBATIM `SpeechT5Criterion.reduce_metrics`
(SpeechT5/speecht5/criterions/speecht5_criterion.py:122-437) on synthetic per-task logging outputs of two ranks with a recording
stand-in for `fairseq.metrics` and writes every call it makes (name, value, weight, priority, round) plus the derived values to
tests/golden/reduce_metrics.json.  tests/test_fairseq_surface.py replays the same inputs through speecht5_amd's reduce_metrics."""
import importlib
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_stubs  # noqa: E402


def logging_outputs():
    """What task.train_step returns per rank and micro-batch: {"loss", "sample_size": 1, <task_name>: criterion log}."""
    def sp(i):
        return {"loss": 5123.5 + 7 * i, "ntokens": 492 + i, "nsentences": 8, "sample_size": 492 + i, "ngpu": 1, "loss_m_0": 3090.25 + i,
                "loss_u_0": 2550.125 - i, "loss_features_pen": 12.5 + i, "loss_prob_perplexity": 31.25 * (i + 1), "code_perplexity": 187.5 + i,
                "count_m_0": 492 + i, "correct_m_0": 40 + i, "count_u_0": 3500, "correct_u_0": 33, "dec_loss": 2.97 + 0.01 * i, "l1_loss": 2.33,
                "l2_loss": 4.06, "bce_loss": 0.64 + 0.1 * i, "enc_dec_attn_loss": 0.0123 * (i + 1)}
    def tx(i):
        return {"loss": 8897.6 + i, "ntokens": 1936 + 2 * i, "nsentences": 4, "bart_loss": 8878.7 + i, "sample_size": 1936 + 2 * i,
                "loss_prob_perplexity": 18.9, "code_perplexity": 190.0 - i}
    def s2t(i):
        return {"loss": 210.5 + i, "ce_loss": 180.25, "ctc_loss": 240.75, "nll_loss": 170.5 + i, "ntokens": 50, "nsentences": 1, "sample_size": 50,
                "total": 50, "n_correct": 11 + i, "c_errors": 20, "c_total": 48, "w_errors": 7, "wv_errors": 9, "w_total": 11}
    def t2s(i):
        return {"loss": 3.2 + i, "l1_loss": 1.1, "l2_loss": 1.7, "bce_loss": 0.4, "sample_size": 1, "ngpu": 1, "encoder_alpha": 1.01, "decoder_alpha": 0.99 + i,
                "enc_dec_attn_loss": 0.002}
    def s2c(i):
        return {"loss": 12.5 + i, "nll_loss": 12.5 + i, "ntokens": 8, "sample_size": 8, "total": 8, "n_correct": 3 + i}
    def s2s(i):
        return {"loss": 2.2 + i, "l1_loss": 0.9, "l2_loss": 1.0, "bce_loss": 0.3, "sample_size": 1, "ngpu": 1, "decoder_alpha": 1.0 + i}
    outs = []
    for i in range(2):
        for name, fn in (("speech_pretrain", sp), ("text_pretrain", tx), ("s2t", s2t), ("t2s", t2s), ("s2c", s2c), ("s2s", s2s)):
            log = fn(i)
            outs.append({"loss": log["loss"] / log["sample_size"], "sample_size": 1, "ntokens": log.get("ntokens", 0), name: log})
    return outs


class Recorder:
    """fairseq.logging.metrics semantics of the two calls: weighted-average meters + derived meters."""

    class Meter:
        def __init__(self):
            self.sum, self.count, self.val = 0.0, 0.0, 0.0

        @property
        def avg(self):
            return self.sum / self.count if self.count > 0 else self.val

    def __init__(self):
        self.meters, self.calls, self.derived = {}, [], {}

    def log_scalar(self, key, value, weight=1, priority=10, round=None):
        m = self.meters.setdefault(key, Recorder.Meter())
        v = float(value)
        m.val = v
        m.sum += v * weight
        m.count += weight
        self.calls.append(["scalar", key, v, float(weight), priority, round])

    def log_derived(self, key, fn, priority=20):
        self.derived[key] = fn
        self.calls.append(["derived", key, priority])


def main():
    ref_stubs.load_reference_criterions()
    fs = sys.modules["fairseq"]
    rec = Recorder()
    fs.metrics = rec
    fs.utils.get_perplexity = lambda loss, round=2, base=2: (__import__("builtins").round(base ** loss, round) if loss is not None else 0.0)
    mod = importlib.import_module("speecht5.criterions.speecht5_criterion")
    mod.metrics = rec
    outs = logging_outputs()
    mod.SpeechT5Criterion.reduce_metrics(outs)
    derived = {k: fn(rec.meters) for k, fn in rec.derived.items()}
    path = os.path.join(ROOT, "tests", "golden", "reduce_metrics.json")
    json.dump({"logging_outputs": outs, "calls": rec.calls, "derived": derived,
               "source": "verbatim SpeechT5/speecht5/criterions/speecht5_criterion.py reduce_metrics under oracle/ref_stubs.py"}, open(path, "w"), indent=1)
    print(f"wrote {path}: {len(rec.calls)} metric calls, {len(derived)} derived meters")


if __name__ == "__main__":
    main()

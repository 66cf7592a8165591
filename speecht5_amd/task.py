"""`speecht5` task mirror (SpeechT5/speecht5/tasks/speecht5.py): the registration name and the pieces the hot
path touches -- build_model, train_step (local loss normalisation, :519-556) and valid_step.  Dataset wiring
(manifests, librosa log-mel, multitask batching) is the reference's CPU data plane and is out of scope
(SURVEY.md 2.1 #4); synthetic samples with the same dict schemas feed bench.py/tests."""
import torch

from .fairseq_compat import register_task


class _Dictionary(list):
    """Symbol table with fairseq's special-symbol layout (<s>=0, <pad>=1, </s>=2, <unk>=3)."""

    def __init__(self, symbols=(), extra=()):
        super().__init__(["<s>", "<pad>", "</s>", "<unk>"] + list(symbols) + list(extra))

    def pad(self):
        return 1

    def eos(self):
        return 2

    def bos(self):
        return 0

    def unk(self):
        return 3

    def index(self, sym):
        return list.index(self, sym)


@register_task("speecht5")
class SpeechT5Task:
    def __init__(self, args, dicts, config=None):
        self.args = args
        self.dicts = dicts
        self.config = config
        self.t5_task = getattr(args, "t5_task", "pretrain")
        self.blank_symbol = "<ctc_blank>"

    @classmethod
    def synthetic(cls, args, text_symbols=77, hubert_units=500):
        """Task with synthetic dictionaries: char vocabulary (+<mask>, <ctc_blank> as tasks/speecht5.py:283-287 adds)
        and k-means label dictionary (500 units -> 504 entries)."""
        text = _Dictionary([f"c{i}" for i in range(text_symbols)], ["<mask>", "<ctc_blank>"])
        hub = _Dictionary([str(i) for i in range(hubert_units)])
        return cls(args, {"text": text, "hubert": [hub]})

    @property
    def target_dictionary(self):
        return self.dicts["text"]

    def build_model(self, args):
        from .speecht5 import T5TransformerModel
        args.label_rates = getattr(args, "label_rates", 50)
        args.sample_rate = getattr(args, "sample_rate", 16000)
        return T5TransformerModel.build_model(args, self)

    def train_step(self, sample, model, criterion, optimizer, update_num, ignore_grad=False, sync=True):
        model.train()
        model.set_num_updates(update_num)
        loss, sample_size, logging_output = criterion(model, sample)
        if ignore_grad:
            loss = loss * 0
        loss = loss / sample_size  # normalised locally, sample_size 1 is returned (tasks/speecht5.py:538,556)
        if optimizer is not None and hasattr(optimizer, "backward"):
            optimizer.backward(loss)
        else:
            loss.backward()
        agg = {"sample_size": 1, sample["task_name"]: logging_output}
        for k in ("ntokens", "nsentences"):
            if k in logging_output:
                agg[k] = logging_output[k]
        agg["loss"] = loss.detach().item() if sync else loss.detach()
        return agg["loss"], 1.0, agg

    def valid_step(self, sample, model, criterion):
        model.eval()
        with torch.no_grad():
            loss, sample_size, logging_output = criterion(model, sample)
        return loss, sample_size, logging_output

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

    @classmethod
    def load(cls, path):
        """fairseq dictionary file: one `<symbol> <count>` per line, in index order after the four specials."""
        syms = []
        with open(path, encoding="utf-8") as f:
            for line in f:
                line = line.rstrip("\n")
                if line:
                    syms.append(line.rsplit(" ", 1)[0] if " " in line else line)
        return cls(syms)

    def add_symbol(self, sym):
        if sym in self:
            return list.index(self, sym)
        self.append(sym)
        return len(self) - 1

    def string(self, tokens, bpe_symbol=None, extra_symbols_to_ignore=()):
        skip = {self.pad(), self.eos(), self.bos()} | set(extra_symbols_to_ignore)
        return " ".join(self[int(t)] for t in tokens if int(t) not in skip)

    def pad(self):
        return 1

    def eos(self):
        return 2

    def bos(self):
        return 0

    def unk(self):
        return 3

    def index(self, sym):
        """fairseq Dictionary.index: the symbol's id, <unk> for a symbol the table does not hold."""
        try:
            return list.index(self, sym)
        except ValueError:
            return self.unk()


@register_task("speecht5")
class SpeechT5Task:
    def __init__(self, args, dicts, config=None):
        self.args = args
        self.dicts = dicts
        self.config = config
        self.t5_task = getattr(args, "t5_task", None) or "pretrain"
        # size filter of the data plane (tasks/speecht5.py:277-281)
        msp, mtp = getattr(args, "max_speech_positions", 4000), getattr(args, "max_text_positions", 450)
        self.max_pos = [msp * 256, mtp] if self.t5_task == "pretrain" else [msp * 256]
        # <mask> and the CTC blank are appended to the text dictionary by the task (tasks/speecht5.py:283-287)
        text = dicts.get("text") if isinstance(dicts, dict) else None
        if text is not None and hasattr(text, "add_symbol"):
            self.mask_idx = text.add_symbol("<mask>")
            self.blank_symbol_idx = text.add_symbol("<ctc_blank>")
        self.blank_symbol = "<ctc_blank>"
        self.seed = getattr(args, "seed", 1)

    @staticmethod
    def add_args(parser):
        """Task options, tasks/speecht5.py:44-270 of the reference (speecht5_amd/cli.py holds the table)."""
        from . import cli
        cli.declare(parser, cli.TASK_OPTIONS)

    @classmethod
    def setup_task(cls, args, **kwargs):
        """tasks/speecht5.py:298-318: dictionaries from `<data>/dict.txt` (text) and, for pre-training,
        `<hubert-label-dir>/dict.<label>.txt` per label set."""
        import os.path as op
        dicts = {}
        if args.t5_task == "pretrain":
            if not hasattr(args, "shuffle_instance"):
                args.shuffle_instance = False
            dicts["hubert"] = [_Dictionary.load(f"{args.hubert_label_dir}/dict.{label}.txt") for label in args.hubert_labels]
        dicts["text"] = _Dictionary.load(op.join(args.data, "dict.txt"))
        return cls(args, dicts, None)

    def build_criterion(self, args):
        from . import cli
        from .criterions import SpeechT5Criterion
        return SpeechT5Criterion(self, **cli.criterion_kwargs(args))

    def build_generator(self, models, args, seq_gen_cls=None, extra_gen_cls_kwargs=None):
        """tasks/speecht5.py:599-613: the plug-in's own SequenceGenerator (joint CTC / attention scoring) with the task's
        --ctc-weight; beam / length options come from the generation namespace like fairseq's build_generator reads them."""
        from .sequence_generator import SequenceGenerator
        kw = dict(ctc_weight=getattr(self.args, "ctc_weight", 0.0))
        kw.update(extra_gen_cls_kwargs or {})
        cls_ = seq_gen_cls or SequenceGenerator
        return cls_(models, self.target_dictionary, beam_size=getattr(args, "beam", 5), max_len_a=getattr(args, "max_len_a", 0),
                    max_len_b=getattr(args, "max_len_b", 200), min_len=getattr(args, "min_len", 1),
                    normalize_scores=not getattr(args, "unnormalized", False), len_penalty=getattr(args, "lenpen", 1),
                    unk_penalty=getattr(args, "unkpen", 0), temperature=getattr(args, "temperature", 1.0),
                    no_repeat_ngram_size=getattr(args, "no_repeat_ngram_size", 0), **kw)

    @property
    def source_dictionary(self):
        return None

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
        """tasks/speecht5.py:581-597: 80-bin single-channel mel targets unless a data config says otherwise; label / sample
        rates flow from the task options into the model namespace."""
        from .speecht5 import T5TransformerModel
        cfg = self.config
        args.input_feat_per_channel = getattr(cfg, "input_feat_per_channel", 80) if cfg is not None else 80
        args.input_channels = getattr(cfg, "input_channels", 1) if cfg is not None else 1
        if getattr(args, "speech_odim", None) is None:
            args.speech_odim = args.input_feat_per_channel * args.input_channels
        for k, d in (("label_rates", 50), ("sample_rate", 16000)):
            v = getattr(self.args, k, None)
            setattr(args, k, v if v is not None else getattr(args, k, d))
        model = T5TransformerModel.build_model(args, self)
        self.args.reduction_factor = args.reduction_factor
        return model

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

    def forward_loss(self, sample, model, criterion, update_num):
        """The forward half of train_step (tasks/speecht5.py:519-538): the locally normalised loss of one micro-batch; the
        caller runs .backward() (ddp.accumulate_overlapped runs the forward passes of an update's micro-batches side by side)."""
        model.train()
        model.set_num_updates(update_num)
        loss, sample_size, _ = criterion(model, sample)
        return loss / sample_size

    def valid_step(self, sample, model, criterion):
        model.eval()
        with torch.no_grad():
            loss, sample_size, logging_output = criterion(model, sample)
        return loss, sample_size, logging_output

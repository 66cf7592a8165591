"""`speecht5` task mirror (SpeechT5/speecht5/tasks/speecht5.py): the registration name and the pieces the hot
path touches -- build_model, train_step (local loss normalisation, :519-556) and valid_step.  Dataset wiring
(manifests, librosa log-mel, multitask batching) is the reference's CPU data plane and is out of scope
(SURVEY.md 2.1 #4); synthetic samples with the same dict schemas feed bench.py/tests."""
import torch

from .fairseq_compat import LegacyFairseqTask, register_task


def post_process(sentence, symbol):
    """fairseq/data/data_utils.py post_process."""
    if symbol == "sentencepiece":
        sentence = sentence.replace(" ", "").replace("\u2581", " ").strip()
    elif symbol == "wordpiece":
        sentence = sentence.replace(" ", "").replace("_", " ").strip()
    elif symbol == "letter":
        sentence = sentence.replace(" ", "").replace("|", " ").strip()
    elif symbol == "silence":
        import re
        sentence = re.sub(" +", " ", sentence.replace("<SIL>", "")).strip()
    elif symbol == "_EOW":
        sentence = sentence.replace(" ", "").replace("_EOW", " ").strip()
    elif symbol in {"subword_nmt", "@@ ", "@@"}:
        if symbol == "subword_nmt":
            symbol = "@@ "
        sentence = (sentence + " ").replace(symbol, "").rstrip()
    elif symbol == "none":
        pass
    elif symbol is not None:
        raise NotImplementedError(f"Unknown post_process option: {symbol}")
    return sentence


class _Dictionary(list):
    """Symbol table with fairseq's special-symbol layout (<s>=0, <pad>=1, </s>=2, <unk>=3) and the part of
    fairseq.data.Dictionary's interface the plug-in and the reference's data plane use: index / add_symbol (with fairseq's
    `overwrite` row semantics) / encode_line (what the reference's LabelEncoder calls, tasks/speecht5.py:24-36) / string /
    pad / eos / bos / unk / count / indices / nspecial.  When fairseq is importable setup_task() builds real
    fairseq.data.Dictionary objects instead (the data plane's mmap readers want those)."""
    nspecial = 4

    def __init__(self, symbols=(), extra=()):
        super().__init__()
        self.count = []
        self.indices = {}
        for s in ["<s>", "<pad>", "</s>", "<unk>"] + list(symbols) + list(extra):
            self.add_symbol(s)

    @property
    def symbols(self):
        return self

    def __contains__(self, sym):
        return sym in self.indices

    @classmethod
    def load(cls, path):
        """fairseq dictionary file (fairseq/data/dictionary.py add_from_file): one `<symbol> <count>` per line, in index order
        after the four specials; a duplicate symbol is an error unless its line ends with the `#fairseq:overwrite` flag -- then
        fairseq APPENDS a new row and re-points the symbol to it, so every later symbol keeps fairseq's index."""
        d = cls()
        with open(path, encoding="utf-8") as f:
            for line in f:
                line = line.rstrip()       # (fairseq's add_from_file: line.rstrip() -- CRLF files and trailing blanks parse the same)
                if not line:
                    continue
                sym, _, count = line.rpartition(" ")
                overwrite = False
                if count == "#fairseq:overwrite":
                    overwrite = True
                    sym, _, count = sym.rpartition(" ")
                if not sym or not count.lstrip("-").isdigit():
                    raise ValueError(f"Incorrect dictionary format, expected '<token> <cnt> [flags]': {line!r}")
                if sym in d and not overwrite:
                    raise RuntimeError(f"Duplicate word found when loading Dictionary: '{sym}'. Duplicate words can overwrite earlier "
                                       "ones by adding the #fairseq:overwrite flag at the end of the corresponding row in the dictionary file.")
                d.add_symbol(sym, n=int(count), overwrite=overwrite)
        return d

    def add_symbol(self, sym, n=1, overwrite=False):
        """fairseq Dictionary.add_symbol: an existing symbol only gains count -- unless overwrite, which appends a new row and
        makes it the symbol's index (the old row stays, unreachable by name)."""
        if sym in self.indices and not overwrite:
            idx = self.indices[sym]
            self.count[idx] += n
            return idx
        self.append(sym)
        self.count.append(n)
        self.indices[sym] = len(self) - 1
        return len(self) - 1

    def unk_string(self, escape=False):
        return "<<unk>>" if escape else "<unk>"

    def string(self, tokens, bpe_symbol=None, escape_unk=False, extra_symbols_to_ignore=None, unk_string=None, include_eos=False,
               separator=" "):
        """fairseq Dictionary.string: symbols joined by `separator`, <s> / (unless include_eos) </s> / the extra ids dropped,
        <unk> rendered as `unk_string` (or escaped), then the BPE post-processing fairseq-generate's --post-process / --remove-bpe
        selects (fairseq/data/data_utils.py post_process: sentencepiece, wordpiece, letter, silence, _EOW, subword_nmt / '@@ ')."""
        if hasattr(tokens, "dim") and tokens.dim() == 2:
            return "\n".join(self.string(t, bpe_symbol, escape_unk, extra_symbols_to_ignore, unk_string, include_eos, separator) for t in tokens)
        skip = set(extra_symbols_to_ignore or ())
        if not include_eos:
            skip.add(self.eos())
        skip.add(self.bos())

        def sym(i):
            if i == self.unk():
                return unk_string if unk_string is not None else self.unk_string(escape_unk)
            return self[i]
        sent = separator.join(sym(int(t)) for t in tokens if int(t) not in skip)
        return post_process(sent, bpe_symbol)

    def encode_line(self, line, line_tokenizer=None, add_if_not_exist=True, consumer=None, append_eos=True, reverse_order=False):
        """fairseq Dictionary.encode_line: whitespace tokens -> IntTensor of ids (+ </s>); unknown words are added or map to <unk>."""
        import re
        words = (line_tokenizer(line) if line_tokenizer is not None else re.sub(r"\s+", " ", line).strip().split())
        if reverse_order:
            words = list(reversed(words))
        ids = torch.IntTensor(len(words) + (1 if append_eos else 0))
        for i, w in enumerate(words):
            idx = self.add_symbol(w) if add_if_not_exist else self.index(w)
            if consumer is not None:
                consumer(w, idx)
            ids[i] = idx
        if append_eos:
            ids[len(words)] = self.eos()
        return ids

    def pad(self):
        return 1

    def eos(self):
        return 2

    def bos(self):
        return 0

    def unk(self):
        return 3

    def index(self, sym):
        """fairseq Dictionary.index: the symbol's id (its LAST row when rows were overwritten), <unk> for an unknown symbol."""
        return self.indices.get(sym, self.unk())


def _load_dictionary(path):
    """A real fairseq.data.Dictionary when fairseq imports (the reference's data plane -- LabelEncoder, mmap text readers,
    MultitaskDataset -- runs on these objects, tasks/speecht5.py:298-318), else the in-tree table with the same interface."""
    from .fairseq_compat import HAVE_FAIRSEQ
    if HAVE_FAIRSEQ:
        try:
            from fairseq.data import Dictionary
            return Dictionary.load(path)
        except ImportError:
            pass
    return _Dictionary.load(path)


def _reference_task_class():
    """The reference's SpeechT5Task class (for its data-plane methods), imported without re-registering the task name."""
    import importlib
    import sys
    try:
        import fairseq.tasks as ft
    except ImportError:
        return None
    mod = sys.modules.get("speecht5.tasks.speecht5")
    if mod is None:
        orig = ft.register_task
        ft.register_task = lambda name, dataclass=None: (lambda cls: cls)
        try:
            mod = importlib.import_module("speecht5.tasks.speecht5")
        except ImportError:
            return None
        finally:
            ft.register_task = orig
    return getattr(mod, "SpeechT5Task", None)


@register_task("speecht5")
class SpeechT5Task(LegacyFairseqTask):
    """tasks/speecht5.py:42-43: an argparse-style fairseq task (fairseq's register_task refuses classes that are not FairseqTasks)."""

    def __init__(self, args, dicts, config=None):
        super().__init__(args)
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
        # --iid-noise-target: 600 sentinel tokens <mask0> .. <mask599> (tasks/speecht5.py:289-294), read by the reference's
        # TextPretrainDataset through load_dataset (uni_mask_idxs)
        self.uni_mask_idxs = None
        if getattr(args, "iid_noise_target", False) and text is not None and hasattr(text, "add_symbol"):
            self.uni_mask_idxs = torch.tensor([text.add_symbol("<mask>" + str(i)) for i in range(600)])
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
            dicts["hubert"] = [_load_dictionary(f"{args.hubert_label_dir}/dict.{label}.txt") for label in args.hubert_labels]
        dicts["text"] = _load_dictionary(op.join(args.data, "dict.txt"))
        return cls(args, dicts, None)

    def load_dataset(self, split, epoch=1, combine=False, **kwargs):
        """tasks/speecht5.py:324-517.  The data plane (tsv manifests, librosa log-mel, k-means label files, fairseq's mmap text,
        multitask batching) is the reference's own CPU code and stays there: when the reference plug-in is
        importable (its directory on sys.path, as `--user-dir` puts it) its `load_dataset` runs ON THIS TASK OBJECT -- it only
        touches self.args / self.dicts / self.config / self.datasets / self.mask_idx / self.build_bpe.  Importing the reference's
        task module would register the task name `speecht5` a second time, so the registration decorator is neutralised for the
        duration of that import."""
        ref = _reference_task_class()
        if ref is None:
            raise RuntimeError("SpeechT5Task.load_dataset needs the reference plug-in's data package (speecht5.data.*, "
                               "SpeechT5/speecht5 on sys.path) and fairseq; bench.py / tests feed speecht5_amd.synthetic samples instead")
        return ref.load_dataset(self, split, epoch=epoch, combine=combine, **kwargs)

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
        args.speech_odim = args.input_feat_per_channel * args.input_channels   # (unconditionally, tasks/speecht5.py:590)
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

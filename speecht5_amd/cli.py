"""Command-line surface of the plug-in: the options the reference declares in `T5TransformerModel.add_args`
(SpeechT5/speecht5/models/speecht5.py:117-614), `SpeechT5Task.add_args` (tasks/speecht5.py:44-270) and the criterion's config
dataclass (criterions/speecht5_criterion.py:24-30), kept name for name / dest for dest / default for default so that the
README recipes (`fairseq-train ... --task speecht5 --arch t5_transformer_base --criterion speecht5 ...`) parse unchanged.
tests/test_cli_surface.py compares these tables with the surface recorded from the reference (tests/golden/cli_surface.json).

Table rows: (flags, kind, default, choices).  kind: int / float / str = typed store; "raw" = untyped store; "eval" = python
literal; "flag" = store_true."""
import ast

_DROP = ["relu", "gelu", "gelu_fast", "gelu_accurate", "tanh", "linear"]
_MASKSEL = ["static", "uniform", "normal", "poisson"]

MODEL_OPTIONS = [
    # transformer body
    ("--activation-fn", "str", None, _DROP), ("--dropout", "float", None, None), ("--attention-dropout", "float", None, None),
    ("--activation-dropout --relu-dropout", "float", None, None),
    ("--encoder-embed-dim", "int", None, None), ("--encoder-ffn-embed-dim", "int", None, None), ("--encoder-layers", "int", None, None),
    ("--encoder-attention-heads", "int", None, None), ("--encoder-normalize-before", "flag", False, None),
    ("--decoder-normalize-before", "flag", False, None), ("--decoder-embed-dim", "int", None, None),
    ("--decoder-ffn-embed-dim", "int", None, None), ("--decoder-layers", "int", None, None),
    ("--decoder-attention-heads", "int", None, None), ("--reduction-factor", "int", None, None), ("--spk-embed-dim", "int", None, None),
    ("--layernorm-embedding", "flag", False, None), ("--load-pretrained-encoder-from", "str", None, None),
    ("--freeze-encoder-updates", "int", None, None), ("--freeze-decoder-updates", "int", None, None),
    ("--no-freeze-encoder-layer", "str", None, None), ("--share-input-output-embed", "flag", False, None),
    ("--share-ctc-embed", "flag", False, None), ("--encoder-sliding-window-attn", "int", None, None),
    # speech pre-net variants / speaker options (parsed for recipe compatibility; the SID / SE heads are out of scope)
    ("--encoder-speech-prenet", "str", "conv", ["conv", "linear"]), ("--conv-kernel-sizes", "str", "5,5", None),
    ("--conv-channels", "int", 1024, None), ("--subsample-stride", "str", "2,2", None),
    ("--spk-embed-integration-type", "str", None, ["pre", "add"]), ("--dprenet-dropout-rate", "float", 0.5, None),
    ("--se-predict", "raw", None, ["masking", "target", "delta"]), ("--se-decoder-input", "str", "previous_target", ["previous_target", "source"]),
    ("--modules-filter", "str", None, None), ("--sid-pad-prenet", "flag", False, None), ("--encoder-attn-branch", "str", "identity,full", None),
    ("--encoder-block-branch", "str", None, None), ("--sid-encoder-cls", "raw", None, ["encoder"]),
    ("--sid-shuffle-encoder-input", "flag", False, None), ("--sid-decoder-speaker", "flag", False, None),
    ("--sid-decoder-attn-dim", "int", 128, None), ("--sid-t5-postnet", "flag", False, None), ("--sid-embed-dim", "int", 128, None),
    ("--sid-pooling-layer", "str", "decoder", ["decoder-las", "decoder", "encoder", "encoder-cls", "encoder-speaker"]),
    ("--sid-no-pooling-bn", "flag", False, None), ("--sid-no-embed-postnet", "flag", False, None),
    ("--sid-normalize-postnet", "flag", False, None), ("--sid-softmax-type", "raw", "softmax", ["softmax", "amsoftmax", "aamsoftmax"]),
    ("--softmax-scale", "float", 1.0, None), ("--softmax-margin", "float", 0.0, None), ("--softmax-easy-margin", "flag", False, None),
    ("--encoder-layerdrop", "float", None, None), ("--decoder-layerdrop", "float", None, None),
    # HuBERT-style masking / NCE head
    ("--feature-grad-mult", "float", None, None), ("--logit-temp", "float", None, None), ("--final-dim", "int", None, None),
    ("--hubert-mask-length", "int", None, None), ("--mask-prob", "float", None, None), ("--mask-selection", "raw", None, _MASKSEL),
    ("--mask-other", "float", None, None), ("--mask-min-space", "int", None, None), ("--mask-channel-length", "int", None, None),
    ("--mask-channel-prob", "float", None, None), ("--mask-channel-selection", "raw", None, _MASKSEL),
    ("--mask-channel-other", "float", None, None), ("--mask-channel-min-space", "int", None, None),
    ("--conv-pos", "int", None, None), ("--conv-pos-groups", "int", None, None),
    # code book
    ("--use-codebook", "flag", False, None), ("--codebook-prob", "float", None, None), ("--latent-vars", "int", None, None),
    ("--latent-groups", "int", None, None), ("--latent-dim", "int", None, None), ("--latent-temp", "eval", None, None),
    ("--quantizer-depth", "int", None, None), ("--quantizer-factor", "int", None, None), ("--get-code-distribution", "flag", False, None),
    # relative positions / conv front end / init
    ("--relative-position-embedding", "flag", False, None), ("--num-buckets", "int", 320, None), ("--max-distance", "int", 1280, None),
    ("--encoder-max-relative-position", "int", None, None), ("--decoder-max-relative-position", "int", None, None),
    ("--conv-feature-layers", "str", None, None), ("--conv-bias", "flag", False, None), ("--extractor-mode", "raw", None, ["default", "layer_norm"]),
    ("--bert-init", "flag", False, None), ("--unb-enc-layer", "int", -1, None),
]

TASK_NAMES = ["s2t", "t2s", "s2s", "s2c", "pretrain"]

TASK_OPTIONS = [
    ("data", "positional", None, None),
    ("--config-yaml", "str", "config.yaml", None), ("--max-speech-sample-size", "int", None, None),
    ("--min-speech-sample-size", "int", None, None), ("--max-speech-positions", "int", 4000, None),
    ("--max-text-positions", "int", 450, None), ("--t5-task", "raw", None, TASK_NAMES), ("--bpe-tokenizer", "str", None, None),
    ("--finetune-from-modules", "raw", None, None), ("--finetune-out-of-modules", "raw", None, None),
    # BART text-infilling data options (the text branch of pre-training)
    ("--shorten-method", "raw", "none", ["none", "truncate", "random_crop"]), ("--shorten-data-split-list", "raw", "", None),
    ("--tokens-per-sample", "int", 512, None), ("--sample-break-mode", "str", "eos", None), ("--mask", "float", 0.3, None),
    ("--mask-random", "float", 0.1, None), ("--insert", "float", 0.0, None), ("--permute", "float", 0.0, None),
    ("--rotate", "float", 0.0, None), ("--poisson-lambda", "float", 3.5, None), ("--permute-sentences", "float", 0.0, None),
    ("--mask-length", "str", "span-poisson", ["subword", "word", "span-poisson"]), ("--replace-length", "int", 1, None),
    ("--iid-noise-target", "flag", False, None),
    # HuBERT labels / audio handling
    ("--hubert-labels", "strlist", ["km"], None), ("--hubert-label-dir", "str", None, None), ("--sample-rate", "float", 100, None),
    ("--label-rates", "float", -1, None), ("--normalize", "flag", False, None), ("--enable-padding", "flag", False, None),
    ("--pad-audio", "flag", False, None), ("--random-crop", "flag", False, None), ("--single-target", "flag", False, None),
    ("--batch-ratio", "str", None, None), ("--sample-ratios", "str", None, None), ("--ctc-weight", "float", 0.0, None),
]

# SpeechT5CriterionConfig = LabelSmoothedCrossEntropyCriterionConfig + TextPretrainCriterionConfig + SpeechPretrainCriterionConfig +
# SpeechtoTextLossConfig: field -> (kind, default).  fairseq turns dataclass fields into `--field-name` options.
CRITERION_FIELDS = [
    ("label_smoothing", "float", 0.0), ("report_accuracy", "flag", False), ("ignore_prefix_size", "int", 0), ("sentence_avg", "flag", False),
    ("loss_weights", "eval", None), ("bart_weight", "float", 1.0), ("pred_masked_weight", "float", 1.0), ("pred_nomask_weight", "float", 0.0),
    ("log_keys", "eval", None), ("hubert_weight", "float", 1.0), ("dec_weight", "float", 1.0), ("zero_infinity", "flag", False),
    ("post_process", "str", "sentencepiece"), ("wer_kenlm_model", "str", None), ("wer_lexicon", "str", None), ("wer_lm_weight", "float", 2.0),
    ("wer_word_score", "float", -1.0), ("wer_args", "str", None), ("ce_weight", "float", 1.0), ("ctc_weight", "float", 0.0),
]


def literal_eval(s):
    return ast.literal_eval(s)


def declare(parser, table):
    """Adds every row of an options table to an argparse parser (options already present -- fairseq declares a few of the
    names itself, e.g. --ctc-weight for both the task and the criterion -- are left alone)."""
    have = {f for a in parser._actions for f in a.option_strings} | {a.dest for a in parser._actions if not a.option_strings}
    for flags, kind, default, choices in table:
        names = flags.split()
        if any(n in have for n in names):
            continue
        if kind == "positional":
            parser.add_argument(names[0])
        elif kind == "flag":
            parser.add_argument(*names, action="store_true", default=default)
        elif kind == "strlist":
            parser.add_argument(*names, type=str, nargs="*", default=default)
        else:
            typ = {"int": int, "float": float, "str": str, "raw": None, "eval": literal_eval}[kind]
            # no explicit default for None: fairseq adds model options to a group with argument_default=SUPPRESS, so an
            # option the user did not give stays ABSENT from the namespace and the architecture function's getattr
            # defaults apply (speecht5.py:1252-1447)
            kw = {} if default is None else dict(default=default)
            if typ is not None:
                kw["type"] = typ
            if choices is not None:
                kw["choices"] = choices
            parser.add_argument(*names, **kw)
    return parser


def declare_criterion(parser):
    rows = [("--" + n.replace("_", "-"), kind, default, None) for n, kind, default in CRITERION_FIELDS]
    return declare(parser, rows)


def criterion_kwargs(args):
    """Constructor keyword arguments of SpeechT5Criterion from a parsed namespace (what fairseq's build_criterion does from the
    dataclass: every __init__ parameter that the config has a field for)."""
    out = {}
    for n, _k, default in CRITERION_FIELDS:
        if n in ("zero_infinity", "post_process", "wer_kenlm_model", "wer_lexicon", "wer_lm_weight", "wer_word_score", "wer_args"):
            continue   # SpeechtoTextLossConfig fields that SpeechT5Criterion.__init__ does not take
        out[n] = getattr(args, n, default)
    if out.get("loss_weights") is None:
        out["loss_weights"] = [0.1]
    if out.get("log_keys") is None:
        out["log_keys"] = []
    return out

"""Drop-in surface: the options the registered model / task / criterion
classes declare equal the reference's (tests/golden/cli_surface.json, recorded from /root/reference by
oracle/make_cli_surface.py), and the README pre-training and ASR fine-tuning command lines (SpeechT5/README.md:80-132,
146-210) parse through them.  fairseq itself is absent, so its own generic options (optimizer, scheduler, checkpointing,
distributed) are declared by a stand-in parser here -- they are not this plug-in's surface."""
import argparse
import json
import os

import pytest
import torch

from speecht5_amd import cli
from speecht5_amd.fairseq_compat import ARCH_REGISTRY, CRITERION_REGISTRY, MODEL_REGISTRY, TASK_REGISTRY

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SURF = json.load(open(os.path.join(G, "cli_surface.json")))


def _describe(parser):
    out = {}
    for a in parser._actions:
        if a.dest == "help":
            continue
        t = a.type
        out[a.dest] = dict(flags=list(a.option_strings), type=None if t is None else getattr(t, "__name__", str(t)), default=a.default,
                           choices=list(a.choices) if a.choices else None, action=type(a).__name__)
    return out


@pytest.mark.parametrize("which", ["model", "task"])
def test_declared_options_equal_the_reference(which):
    p = argparse.ArgumentParser()
    (MODEL_REGISTRY["t5_transformer"] if which == "model" else TASK_REGISTRY["speecht5"]).add_args(p)
    ours = _describe(p)
    ref = {a["dest"]: a for a in SURF[which]}
    assert set(ours) == set(ref), (sorted(set(ref) - set(ours)), sorted(set(ours) - set(ref)))
    for dest, r in ref.items():
        o = ours[dest]
        assert o["flags"] == r["flags"], dest
        assert o["type"] == r["type"], (dest, o["type"], r["type"])
        assert o["default"] == r["default"], (dest, o["default"], r["default"])
        assert o["choices"] == r["choices"], dest
        assert o["action"] == r["action"], dest


def test_criterion_fields_equal_the_reference():
    ref = {f["name"]: f for f in SURF["criterion"]}
    ours = {n: (k, d) for n, k, d in cli.CRITERION_FIELDS}
    assert set(ours) == set(ref)
    for n, f in ref.items():
        d = f["default"]
        if isinstance(d, str) and (d.startswith("lambda") or d.startswith("II(")):
            continue   # default_factory / interpolated: resolved at build time (cli.criterion_kwargs)
        assert ours[n][1] == d, (n, ours[n][1], d)


def _fairseq_generic(p):
    """fairseq-train's own options that the README command lines use (stand-in: names and arities only)."""
    for f in ("--save-dir", "--tensorboard-logdir", "--train-subset", "--valid-subset", "--ddp-backend", "--user-dir", "--log-format",
              "--task", "--criterion", "--optimizer", "--adam-betas", "--lr-scheduler", "--arch", "--phase-ratio",
              "--best-checkpoint-metric", "--finetune-from-model"):
        p.add_argument(f, type=str)
    for f in ("--distributed-world-size", "--distributed-port", "--seed", "--num-workers", "--max-tokens", "--update-freq",
              "--max-update", "--warmup-updates", "--total-num-update", "--save-interval-updates", "--required-batch-size-multiple",
              "--keep-last-epochs"):
        p.add_argument(f, type=int)
    for f in ("--adam-eps", "--weight-decay", "--power", "--clip-norm", "--lr", "--final-lr-scale"):
        p.add_argument(f, type=float)
    for f in ("--fp16", "--reset-optimizer", "--skip-invalid-size-inputs-valid-test", "--find-unused-parameters",
              "--maximize-best-checkpoint-metric"):
        p.add_argument(f, action="store_true")


PRETRAIN = """DATA --save-dir SAVE --tensorboard-logdir SAVE --train-subset speech_train|text_train --valid-subset speech_valid|text_valid
 --hubert-label-dir LABELS --distributed-world-size 32 --distributed-port 0 --ddp-backend legacy_ddp --user-dir SpeechT5/speecht5
 --log-format json --seed 1337 --fp16 --task speecht5 --t5-task pretrain --label-rates 50 --sample-rate 16000 --random-crop
 --num-workers 0 --max-tokens 1400000 --max-speech-sample-size 250000 --update-freq 2 --batch-ratio [1,0.0086] --criterion speecht5
 --optimizer adam --reset-optimizer --adam-betas (0.9,0.98) --adam-eps 1e-06 --weight-decay 0.01 --power 1 --clip-norm 5.0 --lr 0.0002
 --lr-scheduler polynomial_decay --max-update 800000 --warmup-updates 64000 --total-num-update 800000 --save-interval-updates 3000
 --skip-invalid-size-inputs-valid-test --required-batch-size-multiple 1 --arch t5_transformer_base --share-input-output-embed
 --find-unused-parameters --bert-init --relative-position-embedding --use-codebook --codebook-prob 0.1 --loss-weights=[10,0.1]
 --max-text-positions 600""".split()

ASR = """DATA --save-dir SAVE --tensorboard-logdir SAVE --train-subset train --valid-subset valid --hubert-label-dir LABELS
 --distributed-world-size 8 --distributed-port 0 --ddp-backend legacy_ddp --user-dir USER --log-format json --seed 1 --fp16
 --task speecht5 --t5-task s2t --sample-rate 16000 --num-workers 0 --max-tokens 1600000 --update-freq 2 --bpe-tokenizer BPE
 --criterion speecht5 --report-accuracy --zero-infinity --ce-weight 0.5 --ctc-weight 0.5 --sentence-avg --optimizer adam
 --adam-betas (0.9,0.98) --adam-eps 1e-08 --weight-decay 0.1 --clip-norm 25.0 --lr 0.00006 --lr-scheduler tri_stage
 --phase-ratio [0.1,0.4,0.5] --final-lr-scale 0.05 --max-update 80000 --max-text-positions 600 --required-batch-size-multiple 1
 --save-interval-updates 3000 --skip-invalid-size-inputs-valid-test --arch t5_transformer_base_asr --share-input-output-embed
 --find-unused-parameters --bert-init --relative-position-embedding --freeze-encoder-updates 13000 --keep-last-epochs 10
 --feature-grad-mult 1.0 --best-checkpoint-metric s2t_accuracy --maximize-best-checkpoint-metric --finetune-from-model CKPT""".split()


def _parse(argv):
    """What fairseq's options.parse_args_and_arch does with --user-dir registered: generic options, then the options of the
    named task / criterion / model, then the architecture function fills the remaining model defaults."""
    p = argparse.ArgumentParser(allow_abbrev=False)
    _fairseq_generic(p)
    pre, _ = p.parse_known_args(argv)
    TASK_REGISTRY[pre.task].add_args(p)
    cli.declare_criterion(p)
    assert pre.criterion in CRITERION_REGISTRY
    # fairseq.options.parse_args_and_arch: model options live in a group whose unspecified options are suppressed
    MODEL_REGISTRY["t5_transformer"].add_args(p.add_argument_group("Model-specific configuration", argument_default=argparse.SUPPRESS))
    args = p.parse_args(argv)
    ARCH_REGISTRY[args.arch](args)
    return args


def test_readme_pretrain_command_line_parses_and_builds(tmp_path):
    args = _parse(PRETRAIN)
    assert args.t5_task == "pretrain" and args.label_rates == 50 and args.sample_rate == 16000 and args.random_crop
    assert args.use_codebook and args.codebook_prob == 0.1 and args.loss_weights == [10, 0.1] and args.bert_init
    assert args.encoder_layers == 12 and args.decoder_layers == 6 and args.encoder_embed_dim == 768   # filled by the arch function
    assert args.relative_position_embedding and args.share_input_output_embed and args.max_text_positions == 600
    # setup_task reads the dictionaries the data directory holds (tasks/speecht5.py:298-318)
    (tmp_path / "dict.txt").write_text("".join(f"{c} 1\n" for c in "abcdefg"))
    (tmp_path / "dict.km.txt").write_text("".join(f"{i} 1\n" for i in range(10)))
    args.data, args.hubert_label_dir = str(tmp_path), str(tmp_path)
    task = TASK_REGISTRY["speecht5"].setup_task(args)
    text = task.dicts["text"]
    assert len(text) == 4 + 7 + 2 and text[-2:] == ["<mask>", "<ctc_blank>"] and len(task.dicts["hubert"][0]) == 14
    assert task.max_pos == [4000 * 256, 600]
    crit = task.build_criterion(args)
    assert crit.speech_pretrain_criterion.loss_weights == [10, 0.1]
    # building the model from the parsed namespace (tiny dims so that this stays a CPU-second test)
    args.encoder_layers, args.decoder_layers, args.encoder_embed_dim, args.encoder_ffn_embed_dim = 1, 1, 64, 128
    args.decoder_embed_dim, args.decoder_ffn_embed_dim, args.encoder_attention_heads, args.decoder_attention_heads = 64, 128, 2, 2
    args.conv_feature_layers = "[(32,10,5)] + [(32,3,2)] * 4 + [(32,2,2)] * 2"
    model = task.build_model(args)
    assert args.speech_odim == 80 and hasattr(model, "quantizer") and model.text_decoder_postnet.output_projection.weight.shape[0] == 13


def test_readme_asr_finetune_command_line_parses():
    args = _parse(ASR)
    assert args.t5_task == "s2t" and args.ce_weight == 0.5 and args.ctc_weight == 0.5 and args.report_accuracy and args.zero_infinity
    assert args.freeze_encoder_updates == 13000 and args.feature_grad_mult == 1.0 and args.mask_channel_prob == 0.5
    assert args.sentence_avg and args.bpe_tokenizer == "BPE"
    kw = cli.criterion_kwargs(args)
    assert kw["ce_weight"] == 0.5 and kw["ctc_weight"] == 0.5 and kw["report_accuracy"] and kw["sentence_avg"]


def test_prune_modules_and_upgrade_state_dict():
    from argparse import Namespace
    from tests.util import Task
    from speecht5_amd.speecht5 import T5TransformerModel
    m = torch.load(os.path.join(G, "tiny_model.pt"), weights_only=False)
    model = T5TransformerModel.build_model(Namespace(**m["args"]), Task())
    # an old-style checkpoint: fused in_proj tensors and numbered decoder layer norms
    sd = {k: v.clone() for k, v in m["state_dict"].items()}
    p = "encoder.layers.0.self_attn."
    sd[p + "in_proj_weight"] = torch.cat([sd.pop(p + f"{n}_proj.weight") for n in "qkv"], 0)
    sd[p + "in_proj_bias"] = torch.cat([sd.pop(p + f"{n}_proj.bias") for n in "qkv"], 0)
    for i, n in enumerate(("self_attn_layer_norm", "encoder_attn_layer_norm", "final_layer_norm")):
        for w in ("weight", "bias"):
            sd[f"decoder.layers.1.layer_norms.{i}.{w}"] = sd.pop(f"decoder.layers.1.{n}.{w}")
    res = model.load_state_dict(sd)
    assert not res.missing_keys and not res.unexpected_keys, res
    for k, v in m["state_dict"].items():
        if v.is_floating_point() and "_float_tensor" not in k:
            assert torch.equal(model.state_dict()[k], v), k
    # a checkpoint with another dictionary size keeps the model's own dictionary-sized tensors
    sd2 = {k: v.clone() for k, v in m["state_dict"].items()}
    for k in list(sd2):
        if k.startswith(("text_", "encoder.proj")) and sd2[k].dim() == 2 and sd2[k].shape[0] == 36:
            sd2[k] = torch.zeros(50, sd2[k].shape[1])
    before = model.text_decoder_postnet.output_projection.weight.clone()
    model.load_state_dict(sd2)
    assert torch.equal(model.text_decoder_postnet.output_projection.weight, before)
    model.prune_modules("t2s")
    for n in ("speech_encoder_prenet", "text_decoder_prenet", "text_decoder_postnet", "quantizer"):
        assert not hasattr(model, n), n
    # the reference's prune_modules deletes `speech_encoder_postnet`, which is not the attribute the HuBERT head lives under
    # (`hubert_layer`, speecht5.py:1070/1084/1093): the head survives pruning and stays in fine-tuned checkpoints
    assert hasattr(model, "hubert_layer")
    assert model.encoder.proj is None and hasattr(model, "speech_decoder_postnet") and hasattr(model, "text_encoder_prenet")
    with pytest.raises(ValueError):
        model.prune_modules("nope")

"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Makes the *verbatim* reference modules under /root/reference/SpeechT5/speecht5
importable in the build container, where `fairseq` and `espnet` are absent
(SURVEY.md section 8c).  Used by `oracle/make_golden.py` to generate the golden
fixtures in `tests/golden/`.  Does NOT travel to the GPU box (needs
/root/reference) and is never used at GPU-test / bench time.

Two kinds of symbols are installed into `sys.modules`:

* re-exports of the dependency-free vendored fairseq helpers that live inside the
  reference tree itself: /root/reference/SpeechLM/modules.py (compute_mask_indices
  :219, init_bert_params :355, PositionalEmbedding :414, LayerNorm :439,
  FairseqDropout :1208, SinusoidalPositionalEmbedding :1296, Fp32LayerNorm :1407,
  LayerDropModuleList :1422, TransposeLast :2070, Fp32GroupNorm :2081,
  GradMultiply :2096, SamePad :2119, get_activation_fn :189, quant_noise :77);
* restatements of third-party pieces that are NOT under /root/reference
  (fairseq is an empty, un-pinned submodule: .gitmodules:1-3; espnet is an
  un-pinned pip dependency: SpeechT5/README.md:32).  Their published algorithms
  are restated here following SURVEY.md Appendix A; where the independently written
  HuggingFace port covers a piece it was used as the cross-check (HiFi-GAN and the
  log-mel front end have committed HF fixtures: tests/test_hifigan.py,
  tests/test_logmel.py; the espnet / fairseq restatements below have no second
  implementation in this image and are pinned only through the whole-model goldens).
"""
import importlib.util
import math
import sys
import types
import uuid

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.onnx.operators  # noqa: F401  (vendored SinusoidalPositionalEmbedding needs it, SpeechLM/modules.py:1349)

REFERENCE_ROOT = "/root/reference"


def _load_vendored():
    spec = importlib.util.spec_from_file_location(
        "_speechlm_vendored", f"{REFERENCE_ROOT}/SpeechLM/modules.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ----------------------------------------------------------------------------------------------
# fairseq restatements (third-party, absent)
# ----------------------------------------------------------------------------------------------
class FairseqIncrementalState(object):
    """fairseq/incremental_decoding_utils.py: per-module uuid-keyed dict entries."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.init_incremental_state()

    def init_incremental_state(self):
        self._incremental_state_id = str(uuid.uuid4())

    def _get_full_incremental_state_key(self, key):
        return "{}.{}".format(self._incremental_state_id, key)

    def get_incremental_state(self, incremental_state, key):
        full_key = self._get_full_incremental_state_key(key)
        if incremental_state is None or full_key not in incremental_state:
            return None
        return incremental_state[full_key]

    def set_incremental_state(self, incremental_state, key, value):
        if incremental_state is not None:
            full_key = self._get_full_incremental_state_key(key)
            incremental_state[full_key] = value
        return incremental_state


def with_incremental_state(cls):
    cls.__bases__ = (FairseqIncrementalState,) + tuple(
        b for b in cls.__bases__ if b != FairseqIncrementalState)
    return cls


class FairseqEncoder(nn.Module):
    def __init__(self, dictionary):
        super().__init__()
        self.dictionary = dictionary


class FairseqDecoder(nn.Module):
    def __init__(self, dictionary):
        super().__init__()
        self.dictionary = dictionary
        self.onnx_trace = False
        self.adaptive_softmax = None


@with_incremental_state
class FairseqIncrementalDecoder(FairseqDecoder):
    def __init__(self, dictionary):
        super().__init__(dictionary)

    def reorder_incremental_state_scripting(self, incremental_state, new_order):
        for module in self.modules():
            if hasattr(module, "reorder_incremental_state"):
                result = module.reorder_incremental_state(incremental_state, new_order)
                if result is not None:
                    incremental_state = result


class BaseFairseqModel(nn.Module):
    def __init__(self):
        super().__init__()

    def get_normalized_probs_scriptable(self, net_output, log_probs, sample=None):
        logits = net_output[0]
        if log_probs:
            return F.log_softmax(logits.float(), dim=-1)
        return F.softmax(logits.float(), dim=-1)


class FairseqEncoderDecoderModel(BaseFairseqModel):
    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder


MODEL_REGISTRY = {}
ARCH_REGISTRY = {}


def register_model(name, dataclass=None):
    def wrap(cls):
        MODEL_REGISTRY[name] = cls
        return cls
    return wrap


def register_model_architecture(model_name, arch_name):
    def wrap(fn):
        ARCH_REGISTRY[arch_name] = fn
        return fn
    return wrap


def Embedding(num_embeddings, embedding_dim, padding_idx):
    """fairseq/models/transformer: N(0, d^-0.5) init, zero pad row."""
    m = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)
    nn.init.normal_(m.weight, mean=0, std=embedding_dim ** -0.5)
    nn.init.constant_(m.weight[padding_idx], 0)
    return m


def Linear(in_features, out_features, bias=True):
    m = nn.Linear(in_features, out_features, bias)
    nn.init.xavier_uniform_(m.weight)
    if bias:
        nn.init.constant_(m.bias, 0.0)
    return m


def lengths_to_padding_mask(lens):
    bsz, max_lens = lens.size(0), torch.max(lens).item()
    mask = torch.arange(max_lens).to(lens.device).view(1, max_lens)
    mask = mask.expand(bsz, -1) >= lens.view(bsz, 1).expand(-1, max_lens)
    return mask


class GumbelVectorQuantizer(nn.Module):
    """Restatement of fairseq/modules/gumbel_vector_quantizer.py;
    call site: SpeechT5/speecht5/models/speecht5.py:97-107, 858-882."""

    def __init__(self, dim, num_vars, temp, groups, combine_groups, vq_dim, time_first,
                 activation=nn.GELU(), weight_proj_depth=1, weight_proj_factor=1):
        super().__init__()
        self.groups = groups
        self.combine_groups = combine_groups
        self.input_dim = dim
        self.num_vars = num_vars
        self.time_first = time_first
        assert vq_dim % groups == 0
        var_dim = vq_dim // groups
        num_groups = groups if not combine_groups else 1
        self.vars = nn.Parameter(torch.FloatTensor(1, num_groups * num_vars, var_dim))
        nn.init.uniform_(self.vars)
        if weight_proj_depth > 1:
            def block(input_dim, output_dim):
                return nn.Sequential(nn.Linear(input_dim, output_dim), activation)
            inner_dim = self.input_dim * weight_proj_factor
            self.weight_proj = nn.Sequential(
                *[block(self.input_dim if i == 0 else inner_dim, inner_dim)
                  for i in range(weight_proj_depth - 1)],
                nn.Linear(inner_dim, groups * num_vars))
        else:
            self.weight_proj = nn.Linear(self.input_dim, groups * num_vars)
            nn.init.normal_(self.weight_proj.weight, mean=0, std=1)
            nn.init.zeros_(self.weight_proj.bias)
        if isinstance(temp, str):
            import ast
            temp = ast.literal_eval(temp)
        assert len(temp) == 3
        self.max_temp, self.min_temp, self.temp_decay = temp
        self.curr_temp = self.max_temp
        self.codebook_indices = None

    def set_num_updates(self, num_updates):
        self.curr_temp = max(self.max_temp * self.temp_decay ** num_updates, self.min_temp)

    def forward(self, x, produce_targets=False):
        result = {"num_vars": self.num_vars * self.groups}
        if not self.time_first:
            x = x.transpose(1, 2)
        bsz, tsz, fsz = x.shape
        x = x.reshape(-1, fsz)
        x = self.weight_proj(x)
        x = x.view(bsz * tsz * self.groups, -1)
        _, k = x.max(-1)
        hard_x = x.new_zeros(*x.shape).scatter_(-1, k.view(-1, 1), 1.0).view(
            bsz * tsz, self.groups, -1)
        hard_probs = torch.mean(hard_x.float(), dim=0)
        result["code_perplexity"] = torch.exp(
            -torch.sum(hard_probs * torch.log(hard_probs + 1e-7), dim=-1)).sum()
        avg_probs = torch.softmax(
            x.view(bsz * tsz, self.groups, -1).float(), dim=-1).mean(dim=0)
        result["prob_perplexity"] = torch.exp(
            -torch.sum(avg_probs * torch.log(avg_probs + 1e-7), dim=-1)).sum()
        result["temp"] = self.curr_temp
        if self.training:
            x = F.gumbel_softmax(x.float(), tau=self.curr_temp, hard=True).type_as(x)
        else:
            x = hard_x
        x = x.view(bsz * tsz, -1)
        vars = self.vars
        if self.combine_groups:
            vars = vars.repeat(1, self.groups, 1)
        if produce_targets:
            result["targets"] = (
                x.view(bsz * tsz * self.groups, -1).argmax(dim=-1)
                .view(bsz, tsz, self.groups).detach())
        x = x.unsqueeze(-1) * vars
        x = x.view(bsz * tsz, self.groups, self.num_vars, -1)
        x = x.sum(-2)
        x = x.view(bsz, tsz, -1)
        if not self.time_first:
            x = x.transpose(1, 2)
        result["x"] = x
        return result


# ----------------------------------------------------------------------------------------------
# espnet restatements (third-party, absent)
# ----------------------------------------------------------------------------------------------
class PositionalEncoding(nn.Module):
    """espnet transformer/embedding.py: x*sqrt(d) + pe (interleaved sin/cos), dropout."""

    def __init__(self, d_model, dropout_rate, max_len=5000, reverse=False):
        super().__init__()
        self.d_model = d_model
        self.reverse = reverse
        self.xscale = math.sqrt(self.d_model)
        self.dropout = nn.Dropout(p=dropout_rate)
        self.pe = None
        self.extend_pe(torch.tensor(0.0).expand(1, max_len))

    def extend_pe(self, x):
        if self.pe is not None and self.pe.size(1) >= x.size(1):
            if self.pe.dtype != x.dtype or self.pe.device != x.device:
                self.pe = self.pe.to(dtype=x.dtype, device=x.device)
            return
        pe = torch.zeros(x.size(1), self.d_model)
        position = torch.arange(0, x.size(1), dtype=torch.float32).unsqueeze(1)
        div_term = torch.exp(
            torch.arange(0, self.d_model, 2, dtype=torch.float32)
            * -(math.log(10000.0) / self.d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.pe = pe.unsqueeze(0).to(device=x.device, dtype=x.dtype)

    def forward(self, x):
        self.extend_pe(x)
        x = x * self.xscale + self.pe[:, : x.size(1)]
        return self.dropout(x)


class ScaledPositionalEncoding(PositionalEncoding):
    """espnet: x + alpha*pe, alpha learnable scalar (init 1.0), dropout."""

    def __init__(self, d_model, dropout_rate, max_len=5000):
        super().__init__(d_model=d_model, dropout_rate=dropout_rate, max_len=max_len)
        self.alpha = nn.Parameter(torch.tensor(1.0))

    def reset_parameters(self):
        self.alpha.data = torch.tensor(1.0)

    def forward(self, x):
        self.extend_pe(x)
        x = x + self.alpha * self.pe[:, : x.size(1)]
        return self.dropout(x)


class TacotronPrenet(nn.Module):
    """espnet tacotron2/decoder.py Prenet: Linear->ReLU->dropout (ALWAYS on)."""

    def __init__(self, idim, n_layers=2, n_units=256, dropout_rate=0.5):
        super().__init__()
        self.dropout_rate = dropout_rate
        self.prenet = nn.ModuleList()
        for layer in range(n_layers):
            n_inputs = idim if layer == 0 else n_units
            self.prenet += [nn.Sequential(nn.Linear(n_inputs, n_units), nn.ReLU())]

    def forward(self, x):
        for i in range(len(self.prenet)):
            x = F.dropout(self.prenet[i](x), self.dropout_rate)
        return x


class TacotronPostnet(nn.Module):
    """espnet tacotron2/decoder.py Postnet: 5x Conv1d(no bias)+BN+tanh(+none last)+dropout."""

    def __init__(self, idim, odim, n_layers=5, n_chans=512, n_filts=5, dropout_rate=0.5,
                 use_batch_norm=True):
        super().__init__()
        self.postnet = nn.ModuleList()
        for layer in range(n_layers - 1):
            ichans = odim if layer == 0 else n_chans
            ochans = odim if layer == n_layers - 1 else n_chans
            if use_batch_norm:
                self.postnet += [nn.Sequential(
                    nn.Conv1d(ichans, ochans, n_filts, stride=1, padding=(n_filts - 1) // 2,
                              bias=False),
                    nn.BatchNorm1d(ochans), nn.Tanh(), nn.Dropout(dropout_rate))]
            else:
                self.postnet += [nn.Sequential(
                    nn.Conv1d(ichans, ochans, n_filts, stride=1, padding=(n_filts - 1) // 2,
                              bias=False),
                    nn.Tanh(), nn.Dropout(dropout_rate))]
        ichans = n_chans if n_layers != 1 else odim
        if use_batch_norm:
            self.postnet += [nn.Sequential(
                nn.Conv1d(ichans, odim, n_filts, stride=1, padding=(n_filts - 1) // 2,
                          bias=False),
                nn.BatchNorm1d(odim), nn.Dropout(dropout_rate))]
        else:
            self.postnet += [nn.Sequential(
                nn.Conv1d(ichans, odim, n_filts, stride=1, padding=(n_filts - 1) // 2,
                          bias=False),
                nn.Dropout(dropout_rate))]

    def forward(self, xs):
        for i in range(len(self.postnet)):
            xs = self.postnet[i](xs)
        return xs


def make_pad_mask(lengths, xs=None, length_dim=-1, maxlen=None):
    if not isinstance(lengths, list):
        lengths = [int(l) for l in lengths]
    bs = len(lengths)
    if maxlen is None:
        maxlen = int(max(lengths))
    seq_range = torch.arange(0, maxlen, dtype=torch.int64)
    seq_range_expand = seq_range.unsqueeze(0).expand(bs, maxlen)
    seq_length_expand = seq_range_expand.new(lengths).unsqueeze(-1)
    return seq_range_expand >= seq_length_expand


def make_non_pad_mask(lengths, xs=None, length_dim=-1):
    return ~make_pad_mask(lengths, xs, length_dim)


class GuidedAttentionLoss(nn.Module):
    """espnet e2e_tts_tacotron2.GuidedAttentionLoss (only what the in-tree subclass
    SpeechT5/speecht5/criterions/text_to_speech_loss.py:370-427 relies on)."""

    def __init__(self, sigma=0.4, alpha=1.0, reset_always=True):
        super().__init__()
        self.sigma = sigma
        self.alpha = alpha
        self.reset_always = reset_always
        self.guided_attn_masks = None
        self.masks = None

    def _reset_masks(self):
        self.guided_attn_masks = None
        self.masks = None


# ----------------------------------------------------------------------------------------------
def install():
    """Install the stub `fairseq` / `espnet` packages into sys.modules (idempotent)."""
    if "fairseq" in sys.modules and getattr(sys.modules["fairseq"], "_st5_stub", False):
        return sys.modules["fairseq"]._vendored
    v = _load_vendored()

    def mk(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        for k, val in attrs.items():
            setattr(m, k, val)
        sys.modules[name] = m
        return m

    utils = mk(
        "fairseq.utils",
        softmax=v.softmax,
        log_softmax=lambda x, dim, onnx_trace=False: F.log_softmax(x, dim=dim, dtype=torch.float32),
        get_activation_fn=v.get_activation_fn,
        get_available_activation_fns=lambda: ["relu", "gelu", "gelu_fast", "gelu_accurate", "tanh", "linear"],
        fill_with_neg_inf=lambda t: t.float().fill_(float("-inf")).type_as(t),
        item=v.utils_item,
        eval_str_list=lambda x, type=float: [type(e) for e in eval(x)] if isinstance(x, str) else x,
        weight_norm=nn.utils.weight_norm,
    )
    fs = mk("fairseq", utils=utils, _st5_stub=True, _vendored=v)
    mk("fairseq.models", FairseqEncoder=FairseqEncoder, FairseqDecoder=FairseqDecoder,
       FairseqIncrementalDecoder=FairseqIncrementalDecoder,
       FairseqEncoderDecoderModel=FairseqEncoderDecoderModel, BaseFairseqModel=BaseFairseqModel,
       register_model=register_model, register_model_architecture=register_model_architecture)
    mk("fairseq.models.transformer", Embedding=Embedding, Linear=Linear, LayerNorm=v.LayerNorm)
    modules = mk(
        "fairseq.modules",
        FairseqDropout=v.FairseqDropout, LayerNorm=v.LayerNorm, Fp32LayerNorm=v.Fp32LayerNorm,
        Fp32GroupNorm=v.Fp32GroupNorm, TransposeLast=v.TransposeLast, SamePad=v.SamePad,
        GradMultiply=v.GradMultiply, PositionalEmbedding=v.PositionalEmbedding,
        LayerDropModuleList=v.LayerDropModuleList, GumbelVectorQuantizer=GumbelVectorQuantizer,
        AdaptiveSoftmax=None, SinusoidalPositionalEmbedding=v.SinusoidalPositionalEmbedding,
        TransformerEncoderLayer=None,  # only used when use_sent_enc_layer=False (never in SpeechT5 recipes)
    )
    mk("fairseq.modules.fairseq_dropout", FairseqDropout=v.FairseqDropout)
    mk("fairseq.modules.quant_noise", quant_noise=v.quant_noise)
    mk("fairseq.modules.checkpoint_activations", checkpoint_wrapper=lambda m, **kw: m)
    mk("fairseq.modules.transformer_sentence_encoder", init_bert_params=v.init_bert_params)
    mk("fairseq.distributed", fsdp_wrap=v.fsdp_wrap)
    mk("fairseq.incremental_decoding_utils", with_incremental_state=with_incremental_state,
       FairseqIncrementalState=FairseqIncrementalState)
    mk("fairseq.data")
    mk("fairseq.data.data_utils", compute_mask_indices=v.compute_mask_indices,
       lengths_to_padding_mask=lengths_to_padding_mask)
    fs.modules = modules

    # ---- criterion-side stubs (only what the verbatim criterion files touch at import/forward time) ----
    class FairseqCriterion(nn.Module):
        def __init__(self, task):
            super().__init__()
            self.task = task
            if hasattr(task, "target_dictionary"):
                tgt_dict = task.target_dictionary
                self.padding_idx = tgt_dict.pad() if tgt_dict is not None else -100

    class _Metrics:
        def __getattr__(self, name):
            return lambda *a, **k: None

    import dataclasses

    @dataclasses.dataclass
    class FairseqDataclass:
        pass

    @dataclasses.dataclass
    class LabelSmoothedCrossEntropyCriterionConfig(FairseqDataclass):
        label_smoothing: float = 0.0
        report_accuracy: bool = False
        ignore_prefix_size: int = 0
        sentence_avg: bool = False

    fs.metrics = _Metrics()
    mk("fairseq.metrics")
    mk("fairseq.criterions", FairseqCriterion=FairseqCriterion,
       register_criterion=lambda name, dataclass=None: (lambda cls: cls))
    mk("fairseq.criterions.label_smoothed_cross_entropy",
       LabelSmoothedCrossEntropyCriterionConfig=LabelSmoothedCrossEntropyCriterionConfig)
    mk("fairseq.dataclass", FairseqDataclass=FairseqDataclass)
    mk("fairseq.tasks", FairseqTask=object)
    mk("fairseq.logging")
    mk("fairseq.logging.meters", safe_round=lambda x, n: round(float(x), n))
    sys.modules["fairseq.data.data_utils"].post_process = lambda s, sym: s
    if "omegaconf" not in sys.modules:
        mk("omegaconf", II=lambda key: False)

    mk("espnet")
    mk("espnet.nets")
    mk("espnet.nets.pytorch_backend")
    mk("espnet.nets.pytorch_backend.transformer")
    mk("espnet.nets.pytorch_backend.transformer.embedding",
       PositionalEncoding=PositionalEncoding, ScaledPositionalEncoding=ScaledPositionalEncoding)
    mk("espnet.nets.pytorch_backend.tacotron2")
    mk("espnet.nets.pytorch_backend.tacotron2.decoder", Prenet=TacotronPrenet, Postnet=TacotronPostnet)
    mk("espnet.nets.pytorch_backend.nets_utils", make_non_pad_mask=make_non_pad_mask,
       make_pad_mask=make_pad_mask)
    mk("espnet.nets.pytorch_backend.e2e_tts_tacotron2", GuidedAttentionLoss=GuidedAttentionLoss)
    return v


def load_reference_criterions():
    """Import the verbatim reference criterion modules (after load_reference_models())."""
    load_reference_models()
    root = f"{REFERENCE_ROOT}/SpeechT5/speecht5"
    if "speecht5.criterions" not in sys.modules:
        m = types.ModuleType("speecht5.criterions")
        m.__path__ = [f"{root}/criterions"]
        sys.modules["speecht5.criterions"] = m
    import importlib
    return SimpleNamespaceLike(
        tts=importlib.import_module("speecht5.criterions.text_to_speech_loss"),
        speech_pretrain=importlib.import_module("speecht5.criterions.speech_pretrain_criterion"),
        text_pretrain=importlib.import_module("speecht5.criterions.text_pretrain_criterion"),
        s2t=importlib.import_module("speecht5.criterions.speech_to_text_loss"),
    )


class SimpleNamespaceLike:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def load_reference_models():
    """Import the verbatim reference package `speecht5.models` (no criterions/tasks/data,
    which need more of fairseq) and return the module `speecht5.models.speecht5`."""
    install()
    root = f"{REFERENCE_ROOT}/SpeechT5/speecht5"
    for name, path in (("speecht5", root), ("speecht5.models", f"{root}/models"),
                       ("speecht5.models.modules", f"{root}/models/modules")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
    import importlib
    return importlib.import_module("speecht5.models.speecht5")


# ---- generation-side stubs: fairseq.search.BeamSearch / fairseq.ngram_repeat_block.NGramRepeatBlock are not in the snapshot.
# Restated from fairseq's published algorithm (fairseq/search.py BeamSearch.step; fairseq/ngram_repeat_block.py, Python path):
# only what SpeechT5/speecht5/sequence_generator.py:122-125,476-488 calls.
class BeamSearch(nn.Module):
    def __init__(self, tgt_dict):
        super().__init__()
        self.pad, self.unk, self.eos = tgt_dict.pad(), tgt_dict.unk(), tgt_dict.eos()
        self.vocab_size = len(tgt_dict)
        self.src_lengths = torch.tensor(-1)
        self.supports_constraints = False
        self.stop_on_max_len = False

    def init_constraints(self, batch_constraints, beam_size):
        pass

    def prune_sentences(self, batch_idxs):
        pass

    def update_constraints(self, active_hypos):
        pass

    def step(self, step, lprobs, scores, prev_output_tokens=None, original_batch_idxs=None):
        bsz, beam_size, vocab_size = lprobs.size()
        if step == 0:
            # at the first step all hypotheses are equally likely, so use only the first beam
            lprobs = lprobs[:, ::beam_size, :].contiguous()
        else:
            assert scores is not None
            lprobs = lprobs + scores[:, :, step - 1].unsqueeze(-1)
        top_prediction = torch.topk(
            lprobs.view(bsz, -1),
            k=min(beam_size * 2, lprobs.view(bsz, -1).size(1) - 1),  # -1 so we never select pad
        )
        scores_buf = top_prediction[0]
        indices_buf = top_prediction[1]
        beams_buf = torch.div(indices_buf, vocab_size, rounding_mode="trunc")
        indices_buf = indices_buf.fmod(vocab_size)
        return scores_buf, indices_buf, beams_buf


class NGramRepeatBlock(nn.Module):
    def __init__(self, no_repeat_ngram_size, use_extension=True):
        super().__init__()
        self.no_repeat_ngram_size = no_repeat_ngram_size

    def forward(self, tokens, lprobs, bsz, beam_size, step):
        n = self.no_repeat_ngram_size
        banned_tokens = [[] for _ in range(bsz * beam_size)]
        if step + 2 - n >= 0:
            cpu_tokens = tokens.cpu()
            gen_ngrams = [{} for _ in range(bsz * beam_size)]
            for bbsz_idx in range(bsz * beam_size):
                gen_tokens = cpu_tokens[bbsz_idx].tolist()
                for ngram in zip(*[gen_tokens[i:] for i in range(n)]):
                    key = ",".join(str(x) for x in ngram[:-1])
                    gen_ngrams[bbsz_idx][key] = gen_ngrams[bbsz_idx].get(key, []) + [ngram[-1]]
            for bbsz_idx in range(bsz * beam_size):
                key = ",".join(str(x) for x in cpu_tokens[bbsz_idx, step + 2 - n: step + 1].tolist())
                banned_tokens[bbsz_idx] = gen_ngrams[bbsz_idx].get(key, [])
        for bbsz_idx in range(bsz * beam_size):
            lprobs[bbsz_idx][torch.tensor(banned_tokens[bbsz_idx], dtype=torch.int64)] = torch.tensor(-math.inf).to(lprobs)
        return lprobs


def load_reference_generator():
    """Import the verbatim SpeechT5/speecht5/sequence_generator.py: fairseq.search / ngram_repeat_block from the restatements above,
    espnet's CTCPrefixScore from the snapshot's own copy (Speech2C/speech2c/models/modules/ctc_prefix_score.py)."""
    import importlib
    import importlib.util
    load_reference_models()
    fs = sys.modules["fairseq"]
    search = types.ModuleType("fairseq.search")
    search.BeamSearch = BeamSearch
    sys.modules["fairseq.search"] = search
    fs.search = search
    ng = types.ModuleType("fairseq.ngram_repeat_block")
    ng.NGramRepeatBlock = NGramRepeatBlock
    sys.modules["fairseq.ngram_repeat_block"] = ng
    sys.modules["fairseq.data"].data_utils = sys.modules["fairseq.data.data_utils"]
    spec = importlib.util.spec_from_file_location(
        "espnet.nets.ctc_prefix_score", f"{REFERENCE_ROOT}/Speech2C/speech2c/models/modules/ctc_prefix_score.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules["espnet.nets.ctc_prefix_score"] = mod
    return importlib.import_module("speecht5.sequence_generator")

"""Generates tests/golden/*.pt by running the VERBATIM reference (microsoft/SpeechT5 under
/root/reference, imported through oracle/ref_stubs.py) on seeded tiny configurations.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (needs /root/reference); the fixtures
it writes are committed and are what pins `oracle/speecht5_oracle.py` and the HIP path.

    python oracle/make_golden.py            # regenerates every fixture

Each fixture holds: the arg namespace (as a dict), the full state dict, the inputs (including every
random draw the reference made: HuBERT span mask, Gumbel noise, codebook time-mix indices), the
reference forward outputs, the reference criterion's loss, and reference gradients.
"""
import argparse
import os
import sys
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


class Dictionary(list):
    """Minimal fairseq.data.Dictionary look-alike: <s>=0 <pad>=1 </s>=2 <unk>=3 then symbols."""

    def __init__(self, n, extra=()):
        super().__init__(["<s>", "<pad>", "</s>", "<unk>"] + [f"s{i}" for i in range(n)] + list(extra))

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

    def string(self, t):
        return " ".join(self[int(i)] for i in t)


class Task:
    def __init__(self, vocab, n_units):
        self.dicts = {"text": Dictionary(vocab, ["<mask>", "<ctc_blank>"]), "hubert": [Dictionary(n_units)]}
        self.t5_task = "pretrain"
        self.target_dictionary = self.dicts["text"]
        self.blank_symbol_idx = self.dicts["text"].index("<ctc_blank>")
        self.blank_symbol = "<ctc_blank>"


def tiny_args(arch="t5_transformer_base", **over):
    a = Namespace()
    a.label_rates, a.sample_rate = 50, 16000
    a.encoder_embed_dim, a.encoder_ffn_embed_dim, a.encoder_layers, a.encoder_attention_heads = 128, 256, 2, 2
    a.decoder_layers, a.decoder_attention_heads = 2, 2
    a.conv_feature_layers = "[(64,10,5)] + [(64,3,2)] * 4 + [(64,2,2)] * 2"
    a.conv_pos, a.conv_pos_groups = 16, 4
    a.final_dim, a.latent_vars, a.latent_groups = 32, 10, 2
    a.dprenet_units, a.postnet_chans = 32, 32
    a.encoder_max_relative_position = a.decoder_max_relative_position = 8
    a.speech_odim = 80
    a.spk_embed_dim = 64
    a.bert_init, a.use_codebook, a.share_input_output_embed = True, True, True
    # deterministic parity setting: every dropout / layerdrop off
    for k in ("dropout", "attention_dropout", "activation_dropout", "encoder_layerdrop", "decoder_layerdrop",
              "dprenet_dropout_rate", "postnet_dropout_rate", "transformer_enc_positional_dropout_rate",
              "transformer_dec_positional_dropout_rate"):
        setattr(a, k, 0.0)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def build(ref, args, arch, task, seed):
    torch.manual_seed(seed)
    np.random.seed(seed)
    ref_stubs.ARCH_REGISTRY[arch](args)
    model = ref.T5TransformerModel.build_model(args, task)
    # perturb the parameters that bert_init / defaults leave at trivial values, so parity is not vacuous
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias") or "layer_norm" in n or "norm_k" in n or n.endswith(".2.weight") or n.endswith(".1.weight"):
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
            if n.endswith("alpha"):
                p.fill_(0.9)
            if n == "speech_decoder_postnet.prob_out.bias":
                p.fill_(-1.0)  # keeps the random-init stop head from firing at the very first frame
    return model


class Recorder:
    """Captures the random draws made inside the reference forward."""

    def __init__(self, ref):
        import importlib
        self.sep = importlib.import_module("speecht5.models.modules.speech_encoder_prenet")
        self.rec = {}

    def __enter__(self):
        self._cmi = self.sep.compute_mask_indices
        self._randperm = torch.randperm
        self._gs = torch.nn.functional.gumbel_softmax

        def cmi(*a, **k):
            m = self._cmi(*a, **k)
            self.rec.setdefault("mask_indices", torch.from_numpy(m.copy()))
            return m

        def randperm(n, *a, **k):
            r = self._randperm(n, *a, **k)
            self.rec["randperm"] = r.clone()
            return r

        def gumbel_softmax(logits, tau=1, hard=False, dim=-1):
            g = -torch.empty_like(logits).exponential_().log()
            self.rec["gumbel_noise"] = g.clone()
            self.rec["tau"] = float(tau)
            y = ((logits + g) / tau).softmax(dim)
            if hard:
                idx = y.max(dim, keepdim=True)[1]
                yh = torch.zeros_like(logits).scatter_(dim, idx, 1.0)
                return yh - y.detach() + y
            return y

        self.sep.compute_mask_indices = cmi
        torch.randperm = randperm
        torch.nn.functional.gumbel_softmax = gumbel_softmax
        return self

    def __exit__(self, *exc):
        self.sep.compute_mask_indices = self._cmi
        torch.randperm = self._randperm
        torch.nn.functional.gumbel_softmax = self._gs


def grads_of(model):
    """Compact gradient record: L2 norm of every gradient; the full tensor when it is small (biases,
    norms, small matrices) and the first 8 rows of the large ones.  Clears the gradients."""
    norms, full, rows = {}, {}, {}
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach()
        norms[n] = float(g.double().norm())
        if g.numel() <= 8192:
            full[n] = g.clone()
        else:
            rows[n] = g.reshape(g.shape[0], -1)[:8].clone()
        p.grad = None
    return dict(norms=norms, full=full, rows=rows)


_MODEL = {}


def shared_model(ref):
    """One tiny model (t5_transformer_base hyper-parameters scaled down) shared by all fixtures; its
    state dict is stored once in tests/golden/tiny_model.pt."""
    if not _MODEL:
        args = tiny_args()
        task = Task(vocab=30, n_units=20)
        model = build(ref, args, "t5_transformer_base", task, seed=11)
        torch.save(dict(args=vars(args), state_dict={k: v.clone() for k, v in model.state_dict().items()}),
                   os.path.join(OUT, "tiny_model.pt"))
        _MODEL.update(args=args, task=task, model=model)
    return _MODEL["args"], _MODEL["task"], _MODEL["model"]


def speech_batch(args, B, S, n_units, pad_last=0, seed=0):
    g = torch.Generator().manual_seed(seed)
    src = torch.randn(B, S, generator=g)
    pm = torch.zeros(B, S, dtype=torch.bool)
    if pad_last:
        pm[-1, S - pad_last:] = True
        src[-1, S - pad_last:] = 0
    L = 1 + S // 256
    L -= L % args.reduction_factor if hasattr(args, "reduction_factor") else 0
    mel = torch.randn(B, L, 80, generator=g) * 0.5 - 1
    r = 2
    prev = torch.cat([mel.new_zeros(B, 1, 80), mel[:, r - 1::r][:, :-1]], 1)
    labels = torch.zeros(B, L)
    labels[:, -1] = 1.0
    return dict(
        net_input=dict(source=src, padding_mask=pm, prev_output_tokens=prev,
                       tgt_lengths=torch.full((B,), prev.shape[1], dtype=torch.long), spkembs=torch.randn(B, 64, generator=g)),
        target_list=[torch.randint(0, n_units, (B, S // 320 + 1), generator=g) + 4],
        labels=labels, dec_target=mel, dec_target_lengths=torch.full((B,), L, dtype=torch.long),
        src_lengths=[S] * B, id=torch.arange(B), task_name="speech_pretrain")


def gen_speech_pretrain(ref, crit):
    args, task, model = shared_model(ref)
    model.train()
    sample = speech_batch(args, B=2, S=6000, n_units=20, pad_last=0, seed=3)
    c = crit.speech_pretrain.SpeechPretrainCriterion(task, False, 1.0, 0.0, loss_weights=[10, 0.1])
    np.random.seed(5)
    torch.manual_seed(5)
    with Recorder(ref) as r:
        net_output, net_output_dec = model(target_list=sample["target_list"], **sample["net_input"])
    rec = dict(r.rec)
    # second, recorded pass through the criterion with identical randomness
    np.random.seed(5)
    torch.manual_seed(5)
    with Recorder(ref):
        loss, sample_size, log = c(model, sample)
    (loss / sample_size).backward()
    before, after, logits, attn = net_output_dec
    n_mix = int(rec["randperm"].numel() * args.codebook_prob)
    fx = dict(
        sample=sample,
        mask_indices=rec["mask_indices"], mix_idx=rec["randperm"][:n_mix], gumbel_noise=rec["gumbel_noise"], tau=rec["tau"],
        out=dict(logit_m=net_output["logit_m_list"][0].detach(), logit_u=net_output["logit_u_list"][0].detach(),
                 features_pen=net_output["features_pen"].detach(), prob_perplexity=net_output["prob_perplexity"].detach(),
                 code_perplexity=net_output["code_perplexity"].detach(), before=before.detach(), after=after.detach(),
                 stop_logits=logits.detach(), attn=attn.detach()),
        loss=loss.detach(), sample_size=sample_size, log={k: v for k, v in log.items() if isinstance(v, (int, float))},
        grads=grads_of(model))
    torch.save(fx, os.path.join(OUT, "tiny_speech_pretrain.pt"))
    print("speech_pretrain: loss", float(loss), "sample_size", sample_size, "grads", len(fx["grads"]))


def text_batch(task, B, T, seed=0, pad=0):
    g = torch.Generator().manual_seed(seed)
    V = len(task.dicts["text"])
    tgt = torch.randint(4, V - 2, (B, T), generator=g)
    tgt[:, -1] = 2
    src = tgt.clone()
    src[:, 3:6] = task.dicts["text"].index("<mask>")
    if pad:
        tgt[-1, T - pad:] = 1
        tgt[-1, T - pad - 1] = 2
        src[-1, T - pad:] = 1
    prev = torch.cat([torch.full((B, 1), 2), tgt[:, :-1]], 1)
    prev[prev == 2] = 2
    prev = torch.where(torch.cat([torch.zeros(B, 1, dtype=torch.bool), tgt[:, :-1].eq(1)], 1), torch.ones_like(prev), prev)
    return dict(net_input=dict(src_tokens=src, src_lengths=torch.full((B,), T), prev_output_tokens=prev),
                target=tgt, ntokens=int(tgt.ne(1).sum()), id=torch.arange(B), task_name="text_pretrain")


def gen_text_pretrain(ref, crit):
    args, task, model = shared_model(ref)
    model.train()
    sample = text_batch(task, B=3, T=21, seed=4, pad=5)
    c = crit.text_pretrain.TextPretrainCriterion(task, False, 1.0, loss_weights=[0.1])
    np.random.seed(6)
    torch.manual_seed(6)
    with Recorder(ref) as r:
        loss, sample_size, log = c(model, sample)
    rec = dict(r.rec)
    (loss / sample_size).backward()
    model.eval()
    with torch.no_grad(), Recorder(ref):
        torch.manual_seed(6)
    n_mix = int(rec["randperm"].numel() * args.codebook_prob)
    fx = dict(sample=sample,
              mix_idx=rec["randperm"][:n_mix], gumbel_noise=rec["gumbel_noise"], tau=rec["tau"],
              loss=loss.detach(), sample_size=sample_size, grads=grads_of(model))
    torch.save(fx, os.path.join(OUT, "tiny_text_pretrain.pt"))
    print("text_pretrain: loss", float(loss), "sample_size", sample_size)


def gen_s2t(ref, crit):
    args, task, model = shared_model(ref)
    model.speech_encoder_prenet.mask_prob = 0.5  # fine-tuning style span mask (recorded below)
    model.train()
    B, S = 2, 7000
    sb = speech_batch(args, B=B, S=S, n_units=20, pad_last=900, seed=7)
    tb = text_batch(task, B=B, T=9, seed=8, pad=2)
    sample = dict(net_input=dict(source=sb["net_input"]["source"], padding_mask=sb["net_input"]["padding_mask"],
                                 prev_output_tokens=tb["net_input"]["prev_output_tokens"], task_name="s2t"),
                  target=tb["target"], target_lengths=tb["target"].ne(1).sum(-1), ntokens=tb["ntokens"], id=torch.arange(B),
                  task_name="s2t")
    cfgc = crit.s2t.SpeechtoTextLossConfig()
    cfgc.zero_infinity, cfgc.post_process, cfgc.wer_args, cfgc.wer_kenlm_model = True, "letter", None, None
    c = crit.s2t.SpeechtoTextLoss(cfgc, task, sentence_avg=False, label_smoothing=0.1, ce_weight=0.5, ctc_weight=0.5)
    np.random.seed(9)
    torch.manual_seed(9)
    with Recorder(ref) as r:
        loss, sample_size, log = c(model, sample)
    rec = dict(r.rec)
    (loss / sample_size).backward()
    grads = grads_of(model)
    # eval: logits + greedy decode through the reference's incremental decoder API
    model.eval()
    with torch.no_grad():
        enc = model.forward_encoder(sample["net_input"]["source"], sample["net_input"]["padding_mask"])
        tokens = torch.full((B, 1), 2, dtype=torch.long)
        inc = {}
        done = torch.zeros(B, dtype=torch.bool)
        forbid = [1, task.dicts["text"].index("<ctc_blank>"), task.dicts["text"].index("<mask>")]
        max_len = 12
        for step in range(max_len):
            out, _ = model.forward_decoder(tokens, enc, inc)
            lp = torch.log_softmax(out[:, -1].float(), -1)
            for f in forbid:
                lp[:, f] = -float("inf")
            if step == max_len - 1:
                lp[:, :2] = -float("inf")
                lp[:, 3:] = -float("inf")
            nxt = lp.argmax(-1)
            nxt = torch.where(done, torch.ones_like(nxt), nxt)
            tokens = torch.cat([tokens, nxt[:, None]], 1)
            done |= nxt.eq(2)
            if bool(done.all()):
                break
        dec_logits, _ = model.forward_decoder(sample["net_input"]["prev_output_tokens"], enc, None)
    n_mix = int(rec["randperm"].numel() * args.codebook_prob)
    fx = dict(sample=sample, mix_idx=rec["randperm"][:n_mix], gumbel_noise=rec["gumbel_noise"], tau=rec["tau"],
              mask_indices=rec["mask_indices"], loss=loss.detach(), sample_size=sample_size,
              log={k: v for k, v in log.items() if isinstance(v, (int, float))}, grads=grads,
              blank_idx=task.dicts["text"].index("<ctc_blank>"), forbid=forbid, max_len=max_len,
              eval_logits=dec_logits.detach(), greedy_tokens=tokens[:, 1:])
    torch.save(fx, os.path.join(OUT, "tiny_s2t.pt"))
    print("s2t: loss", float(loss), "greedy", tokens[:, 1:].tolist())


def gen_t2s(ref, crit):
    args, task, model = shared_model(ref)
    model.train()
    B, T, L = 3, 11, 24
    tb = text_batch(task, B=B, T=T, seed=10, pad=3)
    g = torch.Generator().manual_seed(15)
    mel = torch.randn(B, L, 80, generator=g) * 0.5 - 1
    olens = torch.tensor([24, 20, 17])
    prev = torch.cat([mel.new_zeros(B, 1, 80), mel[:, 1::2][:, :-1]], 1)
    labels = torch.zeros(B, L)
    for b in range(B):
        labels[b, int(olens[b]) - 1:] = 1.0
    sample = dict(net_input=dict(src_tokens=tb["target"], src_lengths=tb["target"].ne(1).sum(-1), prev_output_tokens=prev,
                                 tgt_lengths=torch.div(olens, 2, rounding_mode="floor"), spkembs=torch.randn(B, 64, generator=g),
                                 task_name="t2s"),
                  labels=labels, dec_target=mel, dec_target_lengths=olens, src_lengths=tb["target"].ne(1).sum(-1),
                  target=tb["target"], ntokens=int(olens.sum()), id=torch.arange(B), task_name="t2s")
    c = crit.tts.TexttoSpeechLoss(task, False, use_guided_attn_loss=True, guided_attn_loss_sigma=0.4,
                                  guided_attn_loss_lambda=10.0, num_layers_applied_guided_attn=2,
                                  num_heads_applied_guided_attn=2)
    torch.manual_seed(16)
    # the reference forward raises for text->speech with use_codebook (speecht5.py:874: hubert_results is
    # undefined without target_list), so TTS runs with the codebook off, as its fine-tuning recipe does
    model.use_codebook = False
    net_output = model(**sample["net_input"])
    loss, l1, l2, bce, ga = c.compute_loss(model, net_output, sample)
    loss.backward()
    before, after, logits, attn = net_output
    # inference: generate_speech on the first utterance (eval; pre-net dropout is p=0 in this config).
    # The training forward above updated the BatchNorm running statistics: restore the stored weights.
    model.load_state_dict(torch.load(os.path.join(OUT, "tiny_model.pt"), weights_only=False)["state_dict"])
    model.eval()
    with torch.no_grad():
        n = int(sample["net_input"]["src_lengths"][0])
        mel_gen = model.generate_speech(source=None, src_tokens=sample["net_input"]["src_tokens"][:1, :n],
                                        spkembs=sample["net_input"]["spkembs"][:1])
        mel_gen = mel_gen[0] if isinstance(mel_gen, (tuple, list)) else mel_gen
    model.use_codebook = True
    fx = dict(sample=sample,
              out=dict(before=before.detach(), after=after.detach(), stop_logits=logits.detach(),
                       attn=[a.detach() for a in attn]),
              loss=loss.detach(), l1=l1.detach(), bce=bce.detach(), guided=ga.detach(), grads=grads_of(model),
              generated_mel=mel_gen.detach())
    torch.save(fx, os.path.join(OUT, "tiny_t2s.pt"))
    print("t2s: loss", float(loss), "guided", float(ga), "generated frames", tuple(mel_gen.shape))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    ref = ref_stubs.load_reference_models()
    crit = ref_stubs.load_reference_criterions()
    gens = dict(speech_pretrain=gen_speech_pretrain, text_pretrain=gen_text_pretrain, s2t=gen_s2t, t2s=gen_t2s)
    for name, fn in gens.items():
        if a.only is None or a.only == name:
            fn(ref, crit)

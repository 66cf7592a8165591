"""Criterion mirrors of SpeechT5/speecht5/criterions/*.py: same class names, constructor arguments,
`forward(model, sample) -> (loss, sample_size, logging_output)` contract and loss definitions.

The model forward/backward underneath runs on the HIP kernels.  The scalar loss arithmetic on the model
outputs (masked L1/MSE/BCE, guided attention, NCE/label-smoothed CE, CTC) is evaluated with fp32 torch
device ops in this round -- SURVEY.md 8(f) ranks fusing it as the next row after the model path.
`sync_logging=False` keeps the logged scalars as device tensors (no `.item()` host syncs inside a timed
step); the default mirrors the reference and returns Python floats."""
import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as Fn
from . import fairseq_compat as _fc
from .fairseq_compat import FairseqCriterion, register_criterion
from .modules.speech_encoder_prenet import SpeechEncoderPrenet


def _item(x, sync):
    if not torch.is_tensor(x):
        return x
    return x.detach().item() if sync else x.detach()


@contextlib.contextmanager
def alignment_weights(model, wanted):
    """The decoder hands back the last layer's head-averaged cross-attention weights with every forward (reference
    decoder.py:171-269, `alignment_layer`), which costs an unfused attention pass (scores, softmax and probabilities in HBM) in
    that layer.  Only the guided-attention term of the speech-decoder loss reads them; a criterion that does not says so here
    and, while training, the layer stays on the fused kernel and the `attn` entry of the output is None.  Loss and gradients are
    unchanged (tests/test_criterions_gpu.py pins them against the reference's)."""
    dec = getattr(model, "decoder", None)
    old = getattr(dec, "materialise_alignment", True)
    if dec is not None:
        dec.materialise_alignment = bool(wanted)
    try:
        yield
    finally:
        if dec is not None:
            dec.materialise_alignment = old


def make_non_pad_mask(lengths, maxlen=None, device=None):
    lengths = torch.as_tensor(lengths, device=device)
    maxlen = int(lengths.max()) if maxlen is None else maxlen
    return torch.arange(maxlen, device=lengths.device)[None, :] < lengths[:, None]


class Tacotron2Loss(nn.Module):
    """text_to_speech_loss.py:263-345 (use_masking; weighted masking is unused by the recipes)."""

    def __init__(self, use_masking=True, use_weighted_masking=False, bce_pos_weight=20.0):
        super().__init__()
        assert use_masking and not use_weighted_masking
        self.bce_pos_weight = float(bce_pos_weight)
        self._pw = {}

    def forward(self, after_outs, before_outs, logits, ys, labels, olens):
        # mean over the non-padded elements, written as mask-weighted sums: the reference's masked_select has a
        # data-dependent output size (= a host sync); the values are identical
        masks = make_non_pad_mask(olens, ys.shape[1], ys.device).unsqueeze(-1)
        m = masks.to(ys.dtype)
        n_frames = m.sum()
        n_elem = n_frames * ys.shape[2]
        da, db = (after_outs - ys) * m, (before_outs - ys) * m
        l1 = (da.abs().sum() + db.abs().sum()) / n_elem
        mse = (da.pow(2).sum() + db.pow(2).sum()) / n_elem
        pw = self._pw.get(ys.device)
        if pw is None:
            pw = self._pw[ys.device] = torch.tensor(self.bce_pos_weight, device=ys.device)
        bce_all = F.binary_cross_entropy_with_logits(logits, labels, pos_weight=pw, reduction="none")
        bce = (bce_all * m[:, :, 0]).sum() / n_frames
        return l1, mse, bce


class GuidedMultiHeadAttentionLoss(nn.Module):
    """text_to_speech_loss.py:370-427."""

    def __init__(self, sigma=0.4, alpha=1.0, reset_always=True):
        super().__init__()
        self.sigma, self.alpha = sigma, alpha

    def forward(self, att_ws, ilens, olens):
        B, _, To, Ti = att_ws.shape
        dev = att_ws.device
        if att_ws.is_cuda and att_ws.dtype == torch.float32:   # one reduction pass + one gradient pass (csrc/ctc_loss.hip)
            return Fn.guided_attention_loss(att_ws, torch.as_tensor(ilens, device=dev), torch.as_tensor(olens, device=dev),
                                            self.sigma, self.alpha)
        ilens = torch.as_tensor(ilens, device=dev).float()
        olens = torch.as_tensor(olens, device=dev).float()
        gx = torch.arange(To, device=dev).float()[None, :, None] / olens[:, None, None]
        gy = torch.arange(Ti, device=dev).float()[None, None, :] / ilens[:, None, None]
        w = 1.0 - torch.exp(-((gy - gx) ** 2) / (2 * self.sigma ** 2))
        mask = (torch.arange(To, device=dev)[None, :, None] < olens[:, None, None]) & \
               (torch.arange(Ti, device=dev)[None, None, :] < ilens[:, None, None])
        losses = w.unsqueeze(1) * att_ws
        return self.alpha * torch.mean(losses.masked_select(mask.unsqueeze(1)))


class TexttoSpeechLoss(nn.Module):
    """text_to_speech_loss.py:53-257."""

    def __init__(self, task, sentence_avg=False, use_masking=True, use_weighted_masking=False, loss_type="L1",
                 bce_pos_weight=5.0, bce_loss_lambda=1.0, use_guided_attn_loss=False, guided_attn_loss_sigma=0.4,
                 guided_attn_loss_lambda=1.0, num_layers_applied_guided_attn=2, num_heads_applied_guided_attn=2,
                 modules_applied_guided_attn=("encoder-decoder",), sync_logging=True):
        super().__init__()
        self.task = task
        self.loss_type = loss_type
        self.bce_loss_lambda = bce_loss_lambda
        self.use_guided_attn_loss = use_guided_attn_loss
        self.criterion = Tacotron2Loss(use_masking, use_weighted_masking, bce_pos_weight)
        self.num_heads_applied_guided_attn = num_heads_applied_guided_attn
        self.modules_applied_guided_attn = modules_applied_guided_attn
        if use_guided_attn_loss:
            self.attn_criterion = GuidedMultiHeadAttentionLoss(sigma=guided_attn_loss_sigma, alpha=guided_attn_loss_lambda)
        self.sync_logging = sync_logging

    def forward(self, model, sample):
        with alignment_weights(model, self.use_guided_attn_loss):
            net_output = model(**sample["net_input"])
        loss, l1, l2, bce, ga = self.compute_loss(model, net_output, sample)
        s = self.sync_logging
        log = {"loss": _item(loss, s), "l1_loss": _item(l1, s), "l2_loss": _item(l2, s), "bce_loss": _item(bce, s),
               "sample_size": 1, "ntokens": sample["ntokens"], "nsentences": sample["target"].size(0)}
        if ga is not None:
            log["enc_dec_attn_loss"] = _item(ga, s)
        return loss, 1, log

    def compute_loss(self, model, net_output, sample):
        before_outs, after_outs, logits, attn = net_output
        labels, ys = sample["labels"], sample["dec_target"]
        olens, ilens = sample["dec_target_lengths"], sample["src_lengths"]
        r = model.reduction_factor
        fused = (after_outs.is_cuda and after_outs.dtype == before_outs.dtype == logits.dtype == ys.dtype == labels.dtype == torch.float32
                 and after_outs.shape[1] == ys.shape[1] - ys.shape[1] % r and ys.is_contiguous() and labels.is_contiguous())
        if fused:
            # one reduction pass + one gradient pass (csrc/losses.hip); the length trim / stop-label fix-up below happen inside
            olens_in = torch.div(torch.as_tensor(olens), r, rounding_mode="floor") if (r > 1 and self.use_guided_attn_loss) else olens
            l1, l2, bce = Fn.tacotron_loss(after_outs, before_outs, logits, ys, labels, torch.as_tensor(olens), r, self.criterion.bce_pos_weight)
        else:
            if r > 1:
                olens_in = torch.div(torch.as_tensor(olens), r, rounding_mode="floor")
                olens = torch.as_tensor(olens) - torch.as_tensor(olens) % r
                # the collater pads to the longest utterance, so max(olens) = padded length rounded down to r (no host sync)
                max_olen = ys.shape[1] - ys.shape[1] % r
                ys = ys[:, :max_olen]
                labels = labels[:, :max_olen]
                labels = torch.scatter(labels, 1, (olens.to(labels.device) - 1).unsqueeze(1), 1.0)
            else:
                olens_in = olens
            l1, l2, bce = self.criterion(after_outs, before_outs, logits, ys, labels, olens)
        if self.loss_type == "L1":
            loss = l1 + self.bce_loss_lambda * bce if self.bce_loss_lambda > 0.0 else l1
        elif self.loss_type == "L2":
            loss = l2 + self.bce_loss_lambda * bce if self.bce_loss_lambda > 0.0 else l2
        elif self.loss_type == "L1+L2":
            loss = l1 + l2 + self.bce_loss_lambda * bce if self.bce_loss_lambda > 0.0 else l1 + l2
        else:
            raise ValueError("unknown --loss-type " + self.loss_type)
        ga = None
        if self.use_guided_attn_loss:
            ilens_in = ilens
            if sample.get("task_name") == "s2s" and isinstance(getattr(model, "speech_encoder_prenet", None), SpeechEncoderPrenet):
                ilens_in = model.speech_encoder_prenet.get_src_lengths(torch.as_tensor(ilens_in))
            if "encoder-decoder" in self.modules_applied_guided_attn:
                att = [a[:, : self.num_heads_applied_guided_attn] for a in attn]
                ga = self.attn_criterion(torch.cat(att, dim=1), ilens_in, olens_in)
                loss = loss + ga
        return loss, l1, l2, bce, ga


class SpeechPretrainCriterion(nn.Module):
    """speech_pretrain_criterion.py:49-198."""

    def __init__(self, task, sentence_avg=False, pred_masked_weight=1.0, pred_nomask_weight=0.0, loss_weights=None, log_keys=None,
                 use_masking=True, use_weighted_masking=False, loss_type="L1", bce_pos_weight=5.0, hubert_weight=1.0,
                 dec_weight=1.0, sync_logging=True):
        super().__init__()
        self.pred_masked_weight = pred_masked_weight
        self.pred_nomask_weight = pred_nomask_weight
        self.loss_weights = loss_weights
        self.log_keys = [] if log_keys is None else log_keys
        self.hubert_weight = hubert_weight
        self.dec_weight = dec_weight
        self.speech_criterion = TexttoSpeechLoss(task, sentence_avg, use_masking, use_weighted_masking, loss_type, bce_pos_weight,
                                                 sync_logging=sync_logging)
        self.sync_logging = sync_logging

    def forward(self, model, sample, reduce=True, log_pred=False):
        s = self.sync_logging
        if self.dec_weight == 0:
            sample["net_input"]["only_hubert"] = True
        with alignment_weights(model, getattr(self.speech_criterion, "use_guided_attn_loss", False)):
            net_output, net_output_dec = model(target_list=sample["target_list"], **sample["net_input"])
        loss, sample_size, log = 0.0, 0, {}
        _summed_only(reduce)
        logp_m_list = model.get_logits(net_output, True)
        targ_m_list = model.get_targets(None, net_output, True)
        loss_m_list = []
        for i, (lm, tm) in enumerate(zip(logp_m_list, targ_m_list)):
            l = Fn.cross_entropy_sum(lm, tm)[0]
            loss_m_list.append(l)
            log[f"loss_m_{i}"] = _item(l, s)
        if self.pred_masked_weight > 0:
            loss = loss + self.pred_masked_weight * sum(loss_m_list)
            sample_size = sample_size + model.get_target_count(net_output, True)
        logp_u_list = model.get_logits(net_output, False)
        targ_u_list = model.get_targets(None, net_output, False)
        loss_u_list = []
        for i, (lu, tu) in enumerate(zip(logp_u_list, targ_u_list)):
            l = Fn.cross_entropy_sum(lu, tu)[0]
            loss_u_list.append(l)
            log[f"loss_u_{i}"] = _item(l, s)
        if self.pred_nomask_weight > 0:
            loss = loss + self.pred_nomask_weight * sum(loss_u_list)
            sample_size = sample_size + model.get_target_count(net_output, False)
        if self.loss_weights is not None:
            extra_losses, names = model.get_extra_losses(net_output)
            lw = self.loss_weights
            if len(lw) == 1 and len(extra_losses) != 1:
                lw = [lw[0]] * len(extra_losses)
            lw = lw[:len(extra_losses)] if len(lw) > len(extra_losses) else lw
            for p, n, coef in zip(extra_losses, names, lw):
                if coef != 0 and p is not None:
                    p = coef * p.float() * sample_size
                    loss = loss + p
                    log[f"loss_{n}"] = _item(p, s)
        log = {"ntokens": sample_size, "nsentences": sample["id"].numel(), "sample_size": sample_size, "ngpu": 1, **log}
        if "loss_prob_perplexity" in log:
            log["code_perplexity"] = _item(net_output["code_perplexity"], s)
        if self.dec_weight == 0.0:
            log["loss"] = _item(loss, s)
            return loss, sample_size, log
        dec_loss, l1, l2, bce, ga = self.speech_criterion.compute_loss(model, net_output_dec, sample)
        log.update(dec_loss=_item(dec_loss, s), l1_loss=_item(l1, s), l2_loss=_item(l2, s), bce_loss=_item(bce, s))
        loss = self.hubert_weight * loss + self.dec_weight * sample_size * dec_loss
        log["loss"] = _item(loss, s)
        return loss, sample_size, log


class TextPretrainCriterion(nn.Module):
    """text_pretrain_criterion.py:35-101."""

    def __init__(self, task, sentence_avg=False, bart_weight=1.0, loss_weights=None, sync_logging=True):
        super().__init__()
        self.task = task
        self.padding_idx = task.target_dictionary.pad()
        self.sentence_avg = sentence_avg
        self.loss_weights = loss_weights
        self.bart_weight = bart_weight
        self.sync_logging = sync_logging

    def forward(self, model, sample, reduce=True):
        s = self.sync_logging
        with alignment_weights(model, False):
            net_output, codebook_out, encoder_output = model(**sample["net_input"])
        _summed_only(reduce)
        # fused log-softmax + NLL + logit gradient on the decoder logits (one kernel)
        logits = net_output[0]
        bart_loss = Fn.cross_entropy_sum(logits.reshape(-1, logits.size(-1)), sample["target"].view(-1), 0.0, self.padding_idx)[0]
        sample_size = sample["target"].size(0) if self.sentence_avg else sample["ntokens"]
        loss = self.bart_weight * bart_loss
        log = {"loss": _item(loss, s), "ntokens": sample["ntokens"], "nsentences": sample["target"].size(0),
               "bart_loss": _item(bart_loss, s), "sample_size": sample_size}
        if "prob_perplexity" in codebook_out:
            extra_losses, names = model.get_extra_losses(codebook_out)
            lw = self.loss_weights
            if len(lw) == 1 and len(extra_losses) != 1:
                lw = [lw[0]] * len(extra_losses)
            lw = lw[len(extra_losses):] if len(lw) > len(extra_losses) else lw  # (sic) text_pretrain_criterion.py:73-76
            for p, n, coef in zip(extra_losses, names, lw):
                if coef != 0 and p is not None:
                    p = coef * p.float() * sample_size
                    loss = loss + p
                    log[f"loss_{n}"] = _item(p, s)
        if "loss_prob_perplexity" in log:
            log["code_perplexity"] = _item(codebook_out["code_perplexity"], s)
        return loss, sample_size, log


def label_smoothed_nll_loss(lprobs, target, epsilon, ignore_index=None, reduce=True):
    """speech_to_text_loss.py:93-110."""
    if target.dim() == lprobs.dim() - 1:
        target = target.unsqueeze(-1)
    nll_loss = -lprobs.gather(dim=-1, index=target)
    smooth_loss = -lprobs.sum(dim=-1, keepdim=True)
    if ignore_index is not None:
        pad_mask = target.eq(ignore_index)
        nll_loss = nll_loss.masked_fill(pad_mask, 0.0)
        smooth_loss = smooth_loss.masked_fill(pad_mask, 0.0)
    if reduce:
        nll_loss, smooth_loss = nll_loss.sum(), smooth_loss.sum()
    eps_i = epsilon / (lprobs.size(-1) - 1)
    return (1.0 - epsilon - eps_i) * nll_loss + eps_i * smooth_loss, nll_loss


class SpeechtoTextLoss(nn.Module):
    """speech_to_text_loss.py:113-337 (CE + CTC; the WER/KenLM validation extras are not part of the hot path)."""

    def __init__(self, cfg, task, sentence_avg=True, label_smoothing=0.1, ignore_prefix_size=0, report_accuracy=False,
                 ce_weight=1.0, ctc_weight=0.0, sync_logging=True):
        super().__init__()
        self.task = task
        self.blank_idx = task.target_dictionary.index(getattr(task, "blank_symbol", "<ctc_blank>"))
        self.pad_idx = task.target_dictionary.pad()
        self.eos_idx = task.target_dictionary.eos()
        self.padding_idx = self.pad_idx
        self.ce_weight, self.ctc_weight = ce_weight, ctc_weight
        self.sentence_avg = sentence_avg
        self.eps = label_smoothing
        self.zero_infinity = getattr(cfg, "zero_infinity", True)
        self.sync_logging = sync_logging

    def forward(self, model, sample, reduce=True):
        s = self.sync_logging
        if self.ce_weight == 0 and self.ctc_weight > 0:
            sample["only_ctc"] = True
        with alignment_weights(model, False):
            net_output_decoder, net_output = model(**sample["net_input"])
        loss_ce = nll = loss_ctc = None
        if self.ce_weight > 0:
            if reduce:
                logits = net_output_decoder[0]
                loss_ce, nll = Fn.cross_entropy_sum(logits.reshape(-1, logits.size(-1)), sample["target"].view(-1), self.eps,
                                                    self.padding_idx)
            else:
                lprobs = model.get_normalized_probs(net_output_decoder, log_probs=True)
                loss_ce, nll = label_smoothed_nll_loss(lprobs.view(-1, lprobs.size(-1)), sample["target"].view(-1), self.eps,
                                                       ignore_index=self.padding_idx, reduce=reduce)
        if self.ctc_weight > 0:
            loss_ctc = self.compute_loss_ctc(model, net_output, sample)
        if self.ce_weight > 0 and self.ctc_weight > 0:
            loss = self.ce_weight * loss_ce + self.ctc_weight * loss_ctc
        else:
            loss = loss_ce if self.ce_weight > 0 else loss_ctc
        ntokens = sample["ntokens"] if "ntokens" in sample else sample["target_lengths"].sum().item()
        sample_size = sample["target"].size(0) if self.sentence_avg else ntokens
        log = {"loss": _item(loss, s), "ce_loss": _item(loss_ce, s) if loss_ce is not None else 0,
               "ctc_loss": _item(loss_ctc, s) if loss_ctc is not None else 0, "nll_loss": _item(nll, s) if nll is not None else 0,
               "ntokens": ntokens, "nsentences": sample["target"].size(0), "sample_size": sample_size}
        return loss, sample_size, log

    def compute_loss_ctc(self, model, net_output, sample):
        lprobs = model.get_normalized_probs_for_ctc(net_output, log_probs=True).contiguous()
        if net_output["encoder_padding_mask"] is not None:
            input_lengths = (~net_output["encoder_padding_mask"][0]).long().sum(-1)
        else:
            input_lengths = lprobs.new_full((lprobs.size(1),), lprobs.size(0), dtype=torch.long)
        pad_mask = (sample["target"] != self.pad_idx) & (sample["target"] != self.eos_idx)
        targets_flat = sample["target"].masked_select(pad_mask)
        target_lengths = sample["target_lengths"] if "target_lengths" in sample else pad_mask.sum(-1)
        target_lengths = target_lengths - 1
        # alpha / beta recursions as two kernels (csrc/ctc_loss.hip); no torch fallback: log-probabilities that are not fp32 on the
        # GPU mean the model did not run on the HIP path, and that must not pass silently
        if not (lprobs.is_cuda and lprobs.dtype == torch.float32):
            raise RuntimeError(f"CTC loss: expected fp32 log-probabilities on the GPU (csrc/ctc_loss.hip), got {lprobs.dtype} on {lprobs.device}")
        return Fn.ctc_loss_sum(lprobs, targets_flat, input_lengths, target_lengths, self.blank_idx, self.zero_infinity,
                               max_target_len=sample["target"].size(1))


def _summed_only(reduce):
    """The criteria compute SUMMED losses on the HIP kernels (what fairseq's trainer asks for: `reduce=True` in every
    `train_step` / `valid_step`, tasks/speecht5.py:519-579).  Per-element losses would have to come from torch ops -- a silent
    fallback the product path does not have."""
    if not reduce:
        raise NotImplementedError("reduce=False (per-element losses) is not on the HIP path; the trainer never asks for it")


@register_criterion("speecht5")
class SpeechT5Criterion(FairseqCriterion):
    """speecht5_criterion.py:32-120: dispatches on sample['task_name'].  A FairseqCriterion (fairseq's register_criterion
    refuses anything else); `reduce_metrics` below is the trainer-side aggregation hook (:122-437)."""

    def __init__(self, task, sentence_avg=False, label_smoothing=0.1, ignore_prefix_size=0, report_accuracy=False,
                 use_masking=True, use_weighted_masking=False, loss_type="L1", bce_pos_weight=5.0, bce_loss_lambda=1.0,
                 use_guided_attn_loss=False, num_heads_applied_guided_attn=2, ce_weight=1.0, ctc_weight=0.0, hubert_weight=1.0,
                 dec_weight=1.0, bart_weight=1.0, pred_masked_weight=1.0, pred_nomask_weight=0.0, loss_weights=None, cfg=None,
                 sync_logging=True, guided_attn_loss_lambda=1.0, guided_attn_loss_sigma=0.4, log_keys=None):
        super().__init__(task)
        self.speech_criterion = TexttoSpeechLoss(task, sentence_avg, use_masking, use_weighted_masking, loss_type, bce_pos_weight,
                                                 bce_loss_lambda, use_guided_attn_loss, guided_attn_loss_sigma, guided_attn_loss_lambda,
                                                 2, num_heads_applied_guided_attn, sync_logging=sync_logging)
        self.text_criterion = SpeechtoTextLoss(cfg, task, sentence_avg, label_smoothing, ignore_prefix_size, report_accuracy,
                                               ce_weight, ctc_weight, sync_logging=sync_logging)
        self.text_pretrain_criterion = TextPretrainCriterion(task, sentence_avg, bart_weight, loss_weights, sync_logging=sync_logging)
        self.speech_pretrain_criterion = SpeechPretrainCriterion(task, sentence_avg, pred_masked_weight, pred_nomask_weight,
                                                                 loss_weights, log_keys, use_masking, use_weighted_masking, loss_type,
                                                                 bce_pos_weight, hubert_weight, dec_weight, sync_logging=sync_logging)

    def forward(self, model, sample, reduce=True):
        task_name = sample["task_name"]
        if task_name in ("s2t", "s2c"):
            return self.text_criterion(model, sample, reduce)
        if task_name in ("t2s", "s2s"):
            return self.speech_criterion(model, sample)
        if task_name == "text_pretrain":
            return self.text_pretrain_criterion(model, sample, reduce)
        if task_name == "speech_pretrain":
            return self.speech_pretrain_criterion(model, sample, reduce)
        raise ValueError(task_name)

    @staticmethod
    def logging_outputs_can_be_summed():
        return False

    @classmethod
    def reduce_metrics(cls, logging_outputs):
        """speecht5_criterion.py:122-437: aggregate the per-task logging dicts that `train_step` nests under the task name
        (one entry per rank and micro-batch) into fairseq's metrics.  Same metric names, values, weights, priorities and
        rounding as the reference (tests/test_fairseq_surface.py compares the call sequence with the verbatim function's),
        written as tables: every family of metrics is `sum(key) / denominator [/ ln 2]`."""
        import math
        import re
        M = _fc.metrics
        ln2 = math.log(2)
        per_task = {}
        for out in logging_outputs:
            for name, log in out.items():
                if name in _TASK_NAMES:
                    per_task.setdefault(name, []).append(log)

        def total(logs, key):
            return sum(log.get(key, 0) for log in logs)

        def ratio_meter(num, den, ndigits, scale=100.0):
            def fn(meters):
                return _fc.safe_round(meters[num].sum * scale / meters[den].sum, ndigits) if meters[den].sum > 0 else float("nan")
            return fn

        def accuracy(prefix, logs):
            tot = _fc.utils_item(total(logs, "total"))
            if tot > 0:
                M.log_scalar(prefix + "_total", tot)
                M.log_scalar(prefix + "_n_correct", _fc.utils_item(total(logs, "n_correct")))
                num, den = prefix + "_n_correct", prefix + "_total"
                M.log_derived(prefix + "_accuracy",
                              lambda meters: round(meters[num].sum * 100.0 / meters[den].sum, 3) if meters[den].sum > 0 else float("nan"), 2)

        def perplexities(prefix, logs):
            pp = [(log["loss_prob_perplexity"], log["sample_size"]) for log in logs if "loss_prob_perplexity" in log]
            cp = [log["code_perplexity"] for log in logs if "code_perplexity" in log]
            if pp and sum(v for v, _ in pp) > 0:
                M.log_scalar(prefix + "_loss_prob_perplexity", sum(v for v, _ in pp) / sum(n for _, n in pp) / ln2, round=3)
            if cp and sum(cp) > 0:
                M.log_scalar(prefix + "_code_perplexity", sum(cp) / len(cp), round=3)

        def nll_or_loss_ppl(prefix, value_sum, ntokens, ss, nll_key, loss_key):
            if ss != ntokens:
                M.log_scalar(nll_key, value_sum / ntokens / ln2, ntokens, round=3)
                M.log_derived(prefix + "_ppl", lambda meters: _fc.get_perplexity(meters[nll_key].avg))
            else:
                M.log_derived(prefix + "_ppl", lambda meters: _fc.get_perplexity(meters[loss_key].avg))

        for name, logs in per_task.items():
            ss = max(1, total(logs, "sample_size"))
            ntok = total(logs, "ntokens")
            if name in ("s2t", "s2c"):
                M.log_scalar(name + "_loss", total(logs, "loss") / ss / ln2, ss, 1, round=3)
                M.log_scalar(name + "_nll_loss", total(logs, "nll_loss") / ntok / ln2, ntok, 2, round=3)
                if name == "s2t":
                    M.log_derived("s2t_ppl", lambda meters: _fc.get_perplexity(meters["s2t_nll_loss"].avg, 2))
                    M.log_scalar("ctc_loss", total(logs, "ctc_loss") / ss / ln2, ntok, 2, round=3)
                    M.log_scalar("ce_loss", total(logs, "ce_loss") / ntok, ntok, 2, round=3)
                accuracy(name, logs)
                if name == "s2t":   # error counters of the WER evaluation (speech_to_text_loss.py) and their derived rates
                    sums = {k: total(logs, k) for k in ("c_errors", "c_total", "w_errors", "wv_errors", "w_total")}
                    for k, v in sums.items():
                        M.log_scalar("_" + k, v)
                    if sums["c_total"] > 0:
                        M.log_derived("uer", ratio_meter("_c_errors", "_c_total", 3))
                    if sums["w_total"] > 0:
                        M.log_derived("wer", ratio_meter("_w_errors", "_w_total", 3))
                        M.log_derived("raw_wer", ratio_meter("_wv_errors", "_w_total", 3))
            elif name in ("t2s", "s2s"):
                M.log_scalar(name + "_loss", total(logs, "loss") / ss, ss, 1, round=5)
                for k in ("l1_loss", "l2_loss", "bce_loss"):
                    M.log_scalar(f"{name}_{k}", total(logs, k) / ss, ss, 2, round=5)
                for k in (("encoder_alpha", "decoder_alpha") if name == "t2s" else ("decoder_alpha",)):
                    M.log_scalar(f"{name}_{k}", total(logs, k) / ss, ss, round=5)
                if "enc_dec_attn_loss" in logs[0]:
                    M.log_scalar(name + "_enc_dec_attn_loss", total(logs, "enc_dec_attn_loss") / ss, ss, round=8)
            elif name == "text_pretrain":
                bart = total(logs, "bart_loss")
                M.log_scalar("text_loss", total(logs, "loss") / ss / ln2, ss, round=3)
                M.log_scalar("bart_loss", bart / ss / ln2, ntok, 2, round=3)
                nll_or_loss_ppl("bart", bart, ntok, ss, "bart_nll_loss", "bart_loss")
                M.log_scalar("bart_wpb", ntok, priority=180, round=1)
                perplexities("text", logs)
            elif name == "speech_pretrain":
                ngpu = total(logs, "ngpu")
                M.log_scalar("hubert_loss", total(logs, "loss") / ss / ln2, ss, round=3)
                nll_or_loss_ppl("hubert", total(logs, "loss"), ntok, ss, "hubert_nll_loss", "hubert_loss")
                counts = {}
                for k in logs[0]:
                    if k.startswith("count_"):
                        counts[k] = sum(log[k] for log in logs)
                        M.log_scalar("hubert_" + k, counts[k])
                for k in logs[0]:
                    if k.startswith("loss_") and k != "loss_prob_perplexity":
                        M.log_scalar("hubert_" + k, sum(log[k] for log in logs) / ss / ln2, round=3)
                    elif k.startswith("correct_"):
                        M.log_scalar("hubert_" + k, sum(log[k] for log in logs) / counts[re.sub("correct", "count", k)])
                perplexities("hubert", logs)
                for k in ("dec_loss", "l1_loss", "l2_loss", "bce_loss"):   # decoder-side terms are per-GPU means
                    M.log_scalar("hubert_" + k, total(logs, k) / ngpu, ss, 2, round=5)
                if "enc_dec_attn_loss" in logs[0]:
                    M.log_scalar("hubert_enc_dec_attn_loss", total(logs, "enc_dec_attn_loss") / ngpu, ss, round=8)
                M.log_scalar("hubert_wpb", ntok, priority=180, round=1)
        ss = max(1, total(logging_outputs, "sample_size"))
        M.log_scalar("loss", total(logging_outputs, "loss") / ss, ss, 1, round=5)


_TASK_NAMES = ("s2t", "t2s", "s2c", "s2s", "text_pretrain", "speech_pretrain")

"""Beam search with joint CTC / attention scoring for the speech-to-text path -- the generator the plug-in hands to
fairseq-generate (SpeechT5/speecht5/sequence_generator.py:26-654, built by tasks/speecht5.py:599-613).

Same constructor arguments, same `generate(models, sample, prefix_tokens=, bos_token=)` contract and the same output
(per sentence a list of hypotheses {tokens, score, attention, alignment, positional_scores}, best first), same search:
fairseq's beam search over 2 x beam candidates per step (fairseq/search.py BeamSearch.step, absent from the snapshot: top-k
over beam x vocabulary of cumulative log-probabilities, first step from the first beam only), hypotheses ending in </s> inside
the top `beam` are finalized, finished sentences leave the batch, length normalisation by (len)^lenpen.  With --ctc-weight > 0
every step re-scores the best `7 x beam` next tokens of each hypothesis with the CTC prefix probability of the FIRST sentence's
encoder output (sequence_generator.py:273-284, 370-418):  lprob' = (1 - w) lprob + w (log psi(h.c) - log psi(h)).

What differs is where it runs: the reference moves every hypothesis to the host, scores it with numpy and keeps the prefix
states in a Python dict keyed by the token string (one device sync per hypothesis and step); here the CTC posterior, the
prefix states of the live hypotheses and the re-scoring stay on the GPU (csrc/ctc_prefix.hip: one thread per (hypothesis,
candidate) pair walks the frames), the states follow the beam with the same index_select as the tokens."""
import math
import sys
from typing import Dict, List, Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import hip

CTC_SCORING_RATIO = 7.0   # sequence_generator.py:24


class BeamSearch:
    """fairseq.search.BeamSearch: top 2 x beam of (beam, token) by cumulative score; step 0 looks at the first beam only
    (all beams hold the same <bos>)."""
    needs_src_lengths = False
    supports_constraints = False
    stop_on_max_len = False

    def __init__(self, tgt_dict):
        self.vocab_size = len(tgt_dict)

    def init_constraints(self, constraints, beam_size):
        pass

    def prune_sentences(self, batch_idxs):
        pass

    def update_constraints(self, active_hypos):
        pass

    def step(self, step, lprobs, scores, prev_output_tokens=None, original_batch_idxs=None):
        bsz, beam_size, vocab = lprobs.size()
        if step == 0:
            lprobs = lprobs[:, ::beam_size, :].contiguous()
        else:
            lprobs = lprobs + scores[:, :, step - 1].unsqueeze(-1)
        flat = lprobs.view(bsz, -1)
        top_scores, top_idx = torch.topk(flat, k=min(beam_size * 2, flat.size(1) - 1))   # (-1: pad is never selected)
        return top_scores, top_idx.fmod(vocab), torch.div(top_idx, vocab, rounding_mode="trunc")


class NGramRepeatBlock:
    """fairseq.ngram_repeat_block.NGramRepeatBlock: a token that would complete an n-gram the hypothesis already holds gets
    -inf (--no-repeat-ngram-size)."""

    def __init__(self, no_repeat_ngram_size):
        self.n = no_repeat_ngram_size

    def __call__(self, tokens, lprobs, bsz, beam_size, step):
        n = self.n
        if step + 2 - n < 0:
            return lprobs
        rows = tokens[:, : step + 1].tolist()
        for b, row in enumerate(rows):
            tail = tuple(row[step + 2 - n: step + 1])
            banned = [row[i + n - 1] for i in range(len(row) - n + 1) if tuple(row[i: i + n - 1]) == tail]
            if banned:
                lprobs[b, banned] = -math.inf
        return lprobs


class CTCPrefixScorer:
    """Device-resident CTC prefix scorer over ONE utterance's posterior x [T, V] (fp32 log-probabilities).
    state: r [n, T, 2];  score(): log psi and states of every (hypothesis, candidate)."""

    def __init__(self, x, blank, eos):
        if not x.is_cuda:
            raise hip.HipKernelError("CTCPrefixScorer needs the CTC posterior on the GPU (there is no CPU path)")
        self.x = x.detach().float().contiguous()
        self.T, self.V = self.x.shape
        self.blank, self.eos = int(blank), int(eos)

    def initial_state(self):
        r = torch.empty(self.T, 2, dtype=torch.float32, device=self.x.device)
        hip.check(hip.lib().st5_ctc_initial_state(self.x.data_ptr(), self.T, self.V, self.blank, r.data_ptr(), hip.stream()),
                  "st5_ctc_initial_state")
        return r

    def score(self, last, out_len, cs, r_prev):
        """last [n] int64: the prefixes' newest label; out_len: labels after <sos>; cs [n, nc] int64; r_prev [n, T, 2]."""
        n, nc = cs.shape
        log_psi = torch.empty(n, nc, dtype=torch.float32, device=self.x.device)
        r_new = torch.empty(n, nc, self.T, 2, dtype=torch.float32, device=self.x.device)
        hip.check(hip.lib().st5_ctc_prefix_score(self.x.data_ptr(), self.T, self.V, self.blank, self.eos, r_prev.contiguous().data_ptr(),
                                                 last.contiguous().data_ptr(), int(out_len), cs.contiguous().data_ptr(), n, nc,
                                                 log_psi.data_ptr(), r_new.data_ptr(), hip.stream()), "st5_ctc_prefix_score")
        return log_psi, r_new


class SequenceGenerator(nn.Module):
    def __init__(self, models, tgt_dict, beam_size=1, max_len_a=0, max_len_b=200, max_len=0, min_len=1, normalize_scores=True,
                 len_penalty=1.0, unk_penalty=0.0, temperature=1.0, match_source_len=False, no_repeat_ngram_size=0,
                 search_strategy=None, eos=None, symbols_to_strip_from_output=None, lm_model=None, lm_weight=1.0, ctc_weight=0.0):
        super().__init__()
        self.model = models if isinstance(models, EnsembleModel) else EnsembleModel(models)
        self.tgt_dict = tgt_dict
        self.pad, self.unk = tgt_dict.pad(), tgt_dict.unk()
        self.eos = tgt_dict.eos() if eos is None else eos
        self.blank = tgt_dict.index("<ctc_blank>")
        self.mask = tgt_dict.index("<mask>")
        idxs = []
        while tgt_dict.index("<mask>" + str(len(idxs))) != self.unk:       # sentinel masks <mask>0, <mask>1, ... when present
            idxs.append(tgt_dict.index("<mask>" + str(len(idxs))))
        self.mask_idxs = torch.tensor(idxs, dtype=torch.long)
        self.symbols_to_strip_from_output = ({self.eos} if symbols_to_strip_from_output is None
                                             else set(symbols_to_strip_from_output) | {self.eos})
        self.vocab_size = len(tgt_dict)
        self.beam_size = min(beam_size, self.vocab_size - 1)                # pad is never selected
        self.max_len_a, self.max_len_b, self.min_len = max_len_a, max_len_b, min_len
        self.max_len = max_len or self.model.max_decoder_positions()
        self.normalize_scores, self.len_penalty, self.unk_penalty = normalize_scores, len_penalty, unk_penalty
        self.temperature, self.match_source_len = temperature, match_source_len
        self.repeat_ngram_blocker = NGramRepeatBlock(no_repeat_ngram_size) if no_repeat_ngram_size > 0 else None
        assert temperature > 0, "--temperature must be greater than 0"
        self.search = BeamSearch(tgt_dict) if search_strategy is None else search_strategy
        self.should_set_src_lengths = getattr(self.search, "needs_src_lengths", False)
        self.model.eval()
        self.lm_model, self.lm_weight, self.ctc_weight = lm_model, lm_weight, ctc_weight
        if self.lm_model is not None:
            self.lm_model.eval()

    def cuda(self):
        self.model.cuda()
        return self

    @torch.no_grad()
    def forward(self, sample, prefix_tokens: Optional[Tensor] = None, bos_token: Optional[int] = None):
        return self._generate(sample, prefix_tokens, bos_token=bos_token)

    @torch.no_grad()
    def generate(self, models, sample, **kwargs):
        """fairseq generator API: `models` is ignored (the ensemble was fixed at construction, :192-205)."""
        return self._generate(sample, **kwargs)

    # -------------------------------------------------------------------------------------------
    def _never(self, lprobs):
        """Symbols no hypothesis may contain: <ctc_blank>, <mask>, the sentinel masks (:446-450)."""
        lprobs[:, self.blank] = -math.inf
        if self.mask != self.unk:
            lprobs[:, self.mask] = -math.inf
        if self.mask_idxs.numel():
            lprobs[:, self.mask_idxs.to(lprobs.device)] = -math.inf
        return lprobs

    def _source_lengths(self, net_input):
        if "src_tokens" in net_input:
            src = net_input["src_tokens"]
            return src, (src.ne(self.eos) & src.ne(self.pad)).long().sum(dim=1)
        for key in ("source", "features"):
            if key in net_input:
                src, pm = net_input[key], net_input.get("padding_mask")
                lens = pm.size(-1) - pm.sum(-1) if pm is not None else torch.tensor(src.size(-1)).to(src)
                return src, lens
        raise Exception("expected src_tokens or source in net input. input keys: " + str(net_input.keys()))

    def _generate(self, sample, prefix_tokens: Optional[Tensor] = None, constraints: Optional[Tensor] = None,
                  bos_token: Optional[int] = None):
        net_input = sample["net_input"]
        incremental_states = [{} for _ in range(self.model.models_size)]
        src_tokens, src_lengths = self._source_lengths(net_input)
        dev = src_tokens.device
        bsz, src_len = src_tokens.size()[:2]
        beam = self.beam_size
        if constraints is not None and not self.search.supports_constraints:
            raise NotImplementedError("Target-side constraints were provided, but search method doesn't support them")
        self.search.init_constraints(constraints, beam)
        max_len = int(src_lengths.max().item()) if self.match_source_len else \
            min(int(self.max_len_a * src_len + self.max_len_b), self.max_len - 1)
        assert self.min_len <= max_len, "min_len cannot be larger than max_len, please adjust these!"

        encoder_outs = self.model.forward_encoder(net_input)

        # joint CTC scoring state (:273-284): posterior of sentence 0; one prefix state + prefix score per live hypothesis row
        ctc = None
        if self.ctc_weight > 0:
            post = self.model.models[0].get_normalized_probs_for_ctc(encoder_outs[0], log_probs=True).transpose(0, 1)  # B x T x V
            ctc = CTCPrefixScorer(post[0], self.blank, self.eos)
            ctc_beam = min(post.shape[-1] - self.mask_idxs.numel(), int(beam * CTC_SCORING_RATIO))
            ctc_state = ctc.initial_state().unsqueeze(0).repeat(bsz * beam, 1, 1)        # [rows, T, 2]
            ctc_prev = torch.zeros(bsz * beam, dtype=torch.float32, device=dev)          # log psi of each row's prefix
            ctc_known = torch.ones(bsz * beam, dtype=torch.bool, device=dev)             # row's prefix was re-scored when created

        new_order = torch.arange(bsz, device=dev).view(-1, 1).repeat(1, beam).view(-1)
        encoder_outs = self.model.reorder_encoder_out(encoder_outs, new_order)
        scores = torch.zeros(bsz * beam, max_len + 1, device=dev)
        tokens = torch.full((bsz * beam, max_len + 2), self.pad, dtype=torch.long, device=dev)
        tokens[:, 0] = self.eos if bos_token is None else bos_token
        attn = None
        cands_to_ignore = torch.zeros(bsz, beam, dtype=torch.bool, device=dev)
        finalized = [[] for _ in range(bsz)]
        finished = [False] * bsz
        num_remaining_sent = bsz
        cand_size = 2 * beam
        bbsz_offsets = (torch.arange(0, bsz, device=dev) * beam).unsqueeze(1)
        cand_offsets = torch.arange(0, cand_size, device=dev)
        reorder_state = None
        batch_idxs = None
        original_batch_idxs = sample["id"] if "id" in sample and isinstance(sample["id"], Tensor) else torch.arange(0, bsz, device=dev)

        for step in range(max_len + 1):                       # one extra step for the closing </s>
            if reorder_state is not None:
                if batch_idxs is not None:                    # sentences left the batch: renumber the surviving rows
                    corr = batch_idxs - torch.arange(batch_idxs.numel(), device=dev)
                    reorder_state.view(-1, beam).add_(corr.unsqueeze(-1) * beam)
                    original_batch_idxs = original_batch_idxs[batch_idxs]
                self.model.reorder_incremental_state(incremental_states, reorder_state)
                encoder_outs = self.model.reorder_encoder_out(encoder_outs, reorder_state)

            lprobs, avg_attn_scores = self.model.forward_decoder(tokens[:, : step + 1], encoder_outs, incremental_states,
                                                                 self.temperature)

            if ctc is not None:
                if not bool(ctc_known.all()):
                    # the reference looks the prefix up in its dict of re-scored extensions and raises KeyError here
                    raise KeyError("a hypothesis was extended by a token outside its CTC candidate list "
                                   f"(ctc_beam={ctc_beam}); its prefix state does not exist")
                cand_lp = self._never(lprobs.clone())
                cs = torch.topk(cand_lp, ctc_beam, dim=-1)[1]                              # [rows, ctc_beam]
                log_psi, states = ctc.score(tokens[:, step], step, cs, ctc_state)
                mixed = (1 - self.ctc_weight) * lprobs.gather(1, cs) + self.ctc_weight * (log_psi - ctc_prev.unsqueeze(1))
                lprobs.scatter_(1, cs, mixed.to(lprobs.dtype))
                slot_of = torch.full_like(lprobs, -1, dtype=torch.long).scatter_(
                    1, cs, torch.arange(ctc_beam, device=dev).expand_as(cs))

            if self.lm_model is not None:
                lm_out = self.lm_model(tokens[:, : step + 1])
                probs = self.lm_model.get_normalized_probs(lm_out, log_probs=True, sample=None)[:, -1, :] * self.lm_weight
                lprobs[:, : probs.size(1)] += probs

            if prefix_tokens is not None and step < prefix_tokens.size(1) and step < max_len:
                lprobs, tokens, scores = self._prefix_tokens(step, lprobs, scores, tokens, prefix_tokens, beam)
            elif step < self.min_len:
                lprobs[:, self.eos] = -math.inf
            lprobs[lprobs != lprobs] = -math.inf
            lprobs[:, self.pad] = -math.inf
            lprobs[:, self.unk] -= self.unk_penalty
            self._never(lprobs)
            if step >= max_len:                               # only </s> may follow
                lprobs[:, : self.eos] = -math.inf
                lprobs[:, self.eos + 1:] = -math.inf

            if avg_attn_scores is not None:
                if attn is None:
                    attn = torch.empty(bsz * beam, avg_attn_scores.size(1), max_len + 2, device=dev, dtype=scores.dtype)
                attn[:, :, step + 1].copy_(avg_attn_scores)
            scores = scores.type_as(lprobs)
            if self.should_set_src_lengths:
                self.search.set_src_lengths(src_lengths)
            if self.repeat_ngram_blocker is not None:
                lprobs = self.repeat_ngram_blocker(tokens, lprobs, bsz, beam, step)

            cand_scores, cand_indices, cand_beams = self.search.step(
                step, lprobs.view(bsz, -1, self.vocab_size), scores.view(bsz, beam, -1)[:, :, :step], tokens[:, : step + 1],
                original_batch_idxs)
            cand_bbsz_idx = cand_beams.add(bbsz_offsets)

            # hypotheses that close with </s> inside the best `beam` candidates are complete
            eos_mask = cand_indices.eq(self.eos) & cand_scores.ne(-math.inf)
            eos_mask[:, :beam][cands_to_ignore] = False
            eos_bbsz_idx = torch.masked_select(cand_bbsz_idx[:, :beam], mask=eos_mask[:, :beam])
            finalized_sents = []
            if eos_bbsz_idx.numel() > 0:
                eos_scores = torch.masked_select(cand_scores[:, :beam], mask=eos_mask[:, :beam])
                finalized_sents = self.finalize_hypos(step, eos_bbsz_idx, eos_scores, tokens, scores, finalized, finished, beam, attn,
                                                      src_lengths, max_len)
                num_remaining_sent -= len(finalized_sents)
            assert num_remaining_sent >= 0
            if num_remaining_sent == 0:
                break
            if self.search.stop_on_max_len and step >= max_len:
                break
            assert step < max_len, f"{step} < {max_len}"

            if len(finalized_sents) > 0:                      # drop the finished sentences from every per-row buffer
                new_bsz = bsz - len(finalized_sents)
                batch_mask = torch.ones(bsz, dtype=torch.bool, device=dev)
                batch_mask[finalized_sents] = False
                batch_idxs = torch.arange(bsz, device=dev).masked_select(batch_mask)
                self.search.prune_sentences(batch_idxs)
                eos_mask, cand_beams = eos_mask[batch_idxs], cand_beams[batch_idxs]
                bbsz_offsets = bbsz_offsets[:new_bsz]
                cand_bbsz_idx = cand_beams.add(bbsz_offsets)
                cand_scores, cand_indices = cand_scores[batch_idxs], cand_indices[batch_idxs]
                if prefix_tokens is not None:
                    prefix_tokens = prefix_tokens[batch_idxs]
                src_lengths = src_lengths[batch_idxs]
                cands_to_ignore = cands_to_ignore[batch_idxs]
                scores = scores.view(bsz, -1)[batch_idxs].view(new_bsz * beam, -1)
                tokens = tokens.view(bsz, -1)[batch_idxs].view(new_bsz * beam, -1)
                if attn is not None:
                    attn = attn.view(bsz, -1)[batch_idxs].view(new_bsz * beam, attn.size(1), -1)
                # (the CTC buffers of this step stay indexed by the rows before the shrink: surviving row -> old row)
                keep_rows = (batch_idxs.unsqueeze(1) * beam + torch.arange(beam, device=dev)).view(-1)
                bsz = new_bsz
            else:
                batch_idxs, keep_rows = None, None

            # the `beam` best candidates that are not complete hypotheses continue: eos candidates get rank + cand_size
            eos_mask[:, :beam] = ~((~cands_to_ignore) & (~eos_mask[:, :beam]))
            active_mask = torch.add(eos_mask.type_as(cand_offsets) * cand_size, cand_offsets[: eos_mask.size(1)])
            new_cands_to_ignore, active_hypos = torch.topk(active_mask, k=beam, dim=1, largest=False)
            cands_to_ignore = new_cands_to_ignore.ge(cand_size)[:, :beam]
            assert (~cands_to_ignore).any(dim=1).all()
            active_bbsz_idx = torch.gather(cand_bbsz_idx, dim=1, index=active_hypos).view(-1)
            active_scores = torch.gather(cand_scores, dim=1, index=active_hypos).view(-1)
            next_tokens = torch.gather(cand_indices, dim=1, index=active_hypos)

            if ctc is not None:
                # prefix state of every continued hypothesis = the re-scored extension (parent row, candidate slot) of this step
                parent_old = active_bbsz_idx if keep_rows is None else keep_rows[active_bbsz_idx]
                slot = slot_of[parent_old, next_tokens.view(-1)]
                ctc_known = slot.ge(0) | cands_to_ignore.view(-1)        # (ignored rows hold finished hypotheses)
                slot_c = slot.clamp(min=0)
                ctc_state = states[parent_old, slot_c]
                ctc_prev = log_psi[parent_old, slot_c]

            tokens[:, : step + 1] = torch.index_select(tokens[:, : step + 1], dim=0, index=active_bbsz_idx)
            tokens.view(bsz, beam, -1)[:, :, step + 1] = next_tokens
            if step > 0:
                scores[:, :step] = torch.index_select(scores[:, :step], dim=0, index=active_bbsz_idx)
            scores.view(bsz, beam, -1)[:, :, step] = active_scores.view(bsz, beam)
            self.search.update_constraints(active_hypos)
            if attn is not None:
                attn[:, :, : step + 2] = torch.index_select(attn[:, :, : step + 2], dim=0, index=active_bbsz_idx)
            reorder_state = active_bbsz_idx

        for sent in range(len(finalized)):                     # best first
            order = torch.sort(torch.tensor([float(h["score"].item()) for h in finalized[sent]]), descending=True)[1]
            finalized[sent] = [finalized[sent][i] for i in order]
        return finalized

    def _prefix_tokens(self, step, lprobs, scores, tokens, prefix_tokens, beam_size):
        """Forced prefix (:656-688): every row may only take its prefix token; rows whose prefix has ended (pad) search freely."""
        prefix_toks = prefix_tokens[:, step].unsqueeze(-1).repeat(1, beam_size).view(-1)
        prefix_lprobs = lprobs.gather(-1, prefix_toks.unsqueeze(-1))
        prefix_mask = prefix_toks.ne(self.pad)
        lprobs[prefix_mask] = torch.min(prefix_lprobs) - 1
        lprobs[prefix_mask] = lprobs[prefix_mask].scatter(-1, prefix_toks[prefix_mask].unsqueeze(-1), prefix_lprobs[prefix_mask])
        eos_mask = prefix_toks.eq(self.eos)
        if eos_mask.any():
            # the prefix closes the sentence: all beams must carry the same hypothesis
            first_beam = tokens[eos_mask].view(-1, beam_size, tokens.size(-1))[:, 0, 1: step + 1]
            eos_mask_batch_dim = eos_mask.view(-1, beam_size)[:, 0]
            target_prefix = prefix_tokens[eos_mask_batch_dim][:, :step]
            assert (first_beam == target_prefix).all()
            tokens = self.replicate_first_beam(tokens, eos_mask_batch_dim, beam_size)
            scores = self.replicate_first_beam(scores, eos_mask_batch_dim, beam_size)
            lprobs = self.replicate_first_beam(lprobs, eos_mask_batch_dim, beam_size)
        return lprobs, tokens, scores

    def replicate_first_beam(self, tensor, mask, beam_size):
        tensor = tensor.view(-1, beam_size, tensor.size(-1))
        tensor[mask] = tensor[mask][:, :1, :]
        return tensor.view(-1, tensor.size(-1))

    def finalize_hypos(self, step, bbsz_idx, eos_scores, tokens, scores, finalized, finished, beam_size, attn, src_lengths, max_len):
        """Store the hypotheses of rows `bbsz_idx` that end at this step (:690-798); returns the (current-batch) indices of the
        sentences that are now complete: `beam_size` hypotheses collected, or the length limit reached."""
        assert bbsz_idx.numel() == eos_scores.numel()
        tokens_clone = tokens.index_select(0, bbsz_idx)[:, 1: step + 2]       # without the leading <bos>
        tokens_clone[:, step] = self.eos
        attn_clone = attn.index_select(0, bbsz_idx)[:, :, 1: step + 2] if attn is not None else None
        pos_scores = scores.index_select(0, bbsz_idx)[:, : step + 1]
        pos_scores[:, step] = eos_scores
        pos_scores[:, 1:] = pos_scores[:, 1:] - pos_scores[:, :-1]           # cumulative -> per position
        if self.normalize_scores:
            eos_scores /= (step + 1) ** self.len_penalty
        # index of each live sentence in the original batch = live index + number of finished sentences before it
        cum_unfin, prev = [], 0
        for f in finished:
            if f:
                prev += 1
            else:
                cum_unfin.append(prev)
        cum_fin_tensor = torch.tensor(cum_unfin, dtype=torch.int).to(bbsz_idx)
        unfin_idx = torch.div(bbsz_idx, beam_size, rounding_mode="trunc")
        sent = unfin_idx + torch.index_select(cum_fin_tensor, 0, unfin_idx)
        unique_seen = torch.unique((sent << 32) + unfin_idx).tolist()
        if self.match_source_len:
            condition = step > torch.index_select(src_lengths, 0, unfin_idx)
            eos_scores = torch.where(condition, torch.tensor(-math.inf).to(eos_scores), eos_scores)
        sent_list = sent.tolist()
        for i in range(bbsz_idx.size(0)):
            if len(finalized[sent_list[i]]) < beam_size:
                finalized[sent_list[i]].append({
                    "tokens": tokens_clone[i], "score": eos_scores[i],
                    "attention": attn_clone[i] if attn_clone is not None else torch.empty(0),
                    "alignment": torch.empty(0), "positional_scores": pos_scores[i]})
        newly_finished = []
        for unique_s in unique_seen:
            unique_sent = unique_s >> 32
            unique_unfin_idx = unique_s - (unique_sent << 32)
            if not finished[unique_sent] and self.is_finished(step, unique_unfin_idx, max_len, len(finalized[unique_sent]), beam_size):
                finished[unique_sent] = True
                newly_finished.append(unique_unfin_idx)
        return newly_finished

    def is_finished(self, step, unfin_idx, max_len, finalized_sent_len, beam_size):
        assert finalized_sent_len <= beam_size
        return finalized_sent_len == beam_size or step == max_len


class EnsembleModel(nn.Module):
    """The generator's view of the model(s) (:819-971): encoder once, one incremental decoder step per call, log-probabilities
    of the newest position (ensembles: log of the mean probability)."""

    def __init__(self, models):
        super().__init__()
        self.models_size = len(models)
        self.single_model = models[0]
        self.models = nn.ModuleList(models)
        self.has_incremental = all(hasattr(m, "decoder") and hasattr(m.decoder, "reorder_incremental_state_scripting") for m in models)

    def forward(self):
        pass

    def has_encoder(self):
        return hasattr(self.single_model, "encoder")

    def is_t5_structure(self):
        m = self.single_model
        return hasattr(m, "text_encoder_prenet") and hasattr(m, "speech_encoder_prenet") or hasattr(m, "encoder_prenet")

    def has_incremental_states(self):
        return self.has_incremental

    def max_decoder_positions(self):
        return min([m.max_decoder_positions() for m in self.models if hasattr(m, "max_decoder_positions")] + [sys.maxsize])

    def forward_encoder(self, net_input: Dict[str, Tensor]):
        if not self.has_encoder():
            return None
        if self.is_t5_structure():
            return [m.forward_encoder_torchscript(net_input) for m in self.models]
        return [m.encoder.forward_torchscript(net_input) for m in self.models]

    def forward_decoder(self, tokens, encoder_outs, incremental_states, temperature: float = 1.0):
        log_probs, avg_attn = [], None
        for i, model in enumerate(self.models):
            encoder_out = encoder_outs[i] if self.has_encoder() else None
            if self.has_incremental_states():
                decoder_out = model.forward_decoder(tokens, encoder_out=encoder_out, incremental_state=incremental_states[i])
            elif hasattr(model, "decoder"):
                decoder_out = model.decoder.forward(tokens, encoder_out=encoder_out)
            else:
                decoder_out = model.forward(tokens)
            attn = None
            if len(decoder_out) > 1 and decoder_out[1] is not None:
                if isinstance(decoder_out[1], Tensor):
                    attn = decoder_out[1]
                else:
                    holder = decoder_out[1]["attn"]
                    attn = holder if isinstance(holder, Tensor) else (holder[0] if holder is not None else None)
                if attn is not None:
                    attn = attn[:, -1, :]
            out = (decoder_out[0][:, -1:, :].div_(temperature), None if len(decoder_out) <= 1 else decoder_out[1])
            probs = model.get_normalized_probs(out, log_probs=True, sample=None)[:, -1, :]
            if self.models_size == 1:
                return probs, attn
            log_probs.append(probs)
            if attn is not None:
                avg_attn = attn if avg_attn is None else avg_attn.add_(attn)
        avg_probs = torch.logsumexp(torch.stack(log_probs, dim=0), dim=0) - math.log(self.models_size)
        if avg_attn is not None:
            avg_attn.div_(self.models_size)
        return avg_probs, avg_attn

    def reorder_encoder_out(self, encoder_outs, new_order):
        if not self.has_encoder():
            return []
        return [m.encoder.reorder_encoder_out(encoder_outs[i], new_order) for i, m in enumerate(self.models)]

    def reorder_incremental_state(self, incremental_states, new_order):
        if not self.has_incremental_states():
            return
        for i, m in enumerate(self.models):
            m.decoder.reorder_incremental_state_scripting(incremental_states[i], new_order)

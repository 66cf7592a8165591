"""BART noise of the text pre-training items: what TextPretrainDataset.__getitem__ does to a token block before
the collater sees it -- /root/reference/SpeechT5/speecht5/data/text_dataset.py:203-226 (order of the noise steps), add_whole_word_mask
:264-397, add_insertion_noise :413-433, add_rolling_noise :405-411 -- with the recipe's defaults (tasks/speecht5.py:141-200: --mask 0.3
--mask-random 0.1 --mask-length span-poisson --poisson-lambda 3.5 --replace-length 1, no insertion / rotation / sentence permutation).

Like the HuBERT span mask (data_utils.py) the noise IS host state in the reference: its draws come from torch's global CPU generator
(Categorical.sample, randperm, uniform_, randint -- NOT re-seeded per item: fairseq's numpy_seed only seeds numpy, which the rolling
noise alone uses) with data-dependent counts, and what they select is a few dozen positions of a 512-token block.  So the draws stay
the reference's own calls, in its order, on the host; what is restated here is the arithmetic around them, in closed form where the
reference iterates: a span that starts at word-start i and covers L whole words ends at the last token t < n - 1 with
(#word starts in (i, t]) <= L - 1 -- one searchsorted over the running count of word starts -- instead of one pass over all spans
per token of the longest span.  tests/test_text_noise_cpu.py: identical items AND identical generator state afterwards, against the
verbatim reference class (tests/golden/collate_t2s_s2t_text.pt), over every replace_length / mask_length / insert combination.
The noised items then go to collate.TextPretrainCollater (ragged gathers on the GPU)."""
import contextlib
import math

import numpy as np
import torch


@contextlib.contextmanager
def numpy_seed(seed, *more):
    """fairseq.data.data_utils.numpy_seed (third party, restated from its published behaviour): numpy's global stream is seeded with
    hash((seed, *more)) % 1e6 inside the block and restored afterwards -- what __getitem__ wraps an item's noise in (:204); only the
    rolling noise draws from numpy, everything else from torch's CPU generator, which this does NOT touch."""
    if seed is None:
        yield
        return
    if len(more) > 0:
        seed = int(hash((seed, *more)) % 1e6)
    state = np.random.get_state()
    np.random.seed(seed)
    try:
        yield
    finally:
        np.random.set_state(state)


class DegenerateItem(ValueError):
    """strict mode: the reference cannot produce this item (its add_whole_word_mask returns ONE tensor where __getitem__ unpacks two
    when the masking budget is zero or only 0-length spans were drawn: text_dataset.py:270, :301 against :210) -- blocks of a few tokens
    only, never the recipe's 512-token blocks."""


class BartNoise:
    def __init__(self, vocab_size, mask_idx, *, eos=2, bos=0, mask=0.3, mask_random=0.1, insert=0.0, rotate=0.0, poisson_lambda=3.5,
                 mask_length="span-poisson", replace_length=1, mask_whole_words=None, permute_sentences=0.0, iid_noise_target=False):
        if replace_length not in (-1, 0, 1):
            raise ValueError(f"invalid arg: replace_length={replace_length}")
        if mask_length not in ("subword", "word", "span-poisson"):
            raise ValueError(f"invalid arg: mask-length={mask_length}")
        if mask_length == "subword" and replace_length not in (0, 1):
            raise ValueError("if using subwords, use replace-length=1 or 0")
        if permute_sentences > 0.0 or iid_noise_target:
            raise NotImplementedError("sentence permutation / T5-form targets are not part of the SpeechT5 pre-training recipe")
        self.strict = False       # True: raise DegenerateItem where the reference's __getitem__ raises (tests: keeps the RNG streams in step)
        self.V, self.mask_idx, self.eos, self.bos = int(vocab_size), int(mask_idx), int(eos), int(bos)
        self.mask_ratio, self.random_ratio, self.insert_ratio, self.rotate_ratio = mask, mask_random, insert, rotate
        self.replace_length = replace_length
        self.word_start_table = mask_whole_words          # ByteTensor over the vocabulary (None: every token starts a word)
        self.span_lengths = None
        if mask_length == "span-poisson":                 # (:173-189) Poisson(lambda) truncated where the mass falls below 1e-7
            lam, probs, term, k = poisson_lambda, [], math.exp(-poisson_lambda), 0
            while k < 128:
                probs.append(term)
                term = term * lam / (k + 1)
                k += 1
                if probs[-1] < 0.0000001:
                    break
            self.span_lengths = torch.distributions.Categorical(torch.FloatTensor(probs))

    # ------------------------------------------------------------------------------------------------------------------
    def __call__(self, tokens):
        """tokens: int64 [n] on the host, <s> ... </s>.  Returns (source, target) as __getitem__ does (:203-226)."""
        assert int(tokens[-1]) == self.eos
        source, target = tokens, tokens.clone()
        if self.mask_ratio > 0:
            self._degenerate = False
            source = self.whole_word_mask(source, self.mask_ratio)
            if self._degenerate and self.strict:
                raise DegenerateItem("zero masking budget / only 0-length spans: the reference raises here")
        if self.insert_ratio > 0:
            source = self.insertion(source, self.insert_ratio)
        if self.rotate_ratio > 0.0 and np.random.random() < self.rotate_ratio:
            source = self.rolling(source)
        assert int(source[0]) == self.bos and int(source[-1]) == self.eos and bool((source[1:-1] >= 1).all())
        return source, target

    def item(self, index, tokens, seed, epoch=0):
        """One dataset item (:203-226): the noise inside numpy_seed(seed, epoch, index)."""
        with numpy_seed(seed, epoch, index):
            source, target = self(tokens)
        return {"id": index, "source": source, "target": target}

    def _random_tokens(self, count):
        return torch.randint(1, self.V, size=(count,))

    def whole_word_mask(self, source, p):
        n = source.size(0)
        starts01 = self.word_start_table.gather(0, source) if self.word_start_table is not None else torch.ones(source.size())
        starts01[0] = 0
        starts01[-1] = 0
        budget = int(math.ceil(starts01.float().sum() * p))       # (fp32 product, as the reference forms it)
        if budget == 0:
            self._degenerate = True
            return source
        inserts = 0
        if self.span_lengths is not None:
            lengths = self.span_lengths.sample(sample_shape=(budget,))
            total = torch.cumsum(lengths, 0)
            while total[-1] < budget:                               # not enough words drawn yet: draw another `budget` spans
                lengths = torch.cat([lengths, self.span_lengths.sample(sample_shape=(budget,))], dim=0)
                total = torch.cumsum(lengths, 0)
            last = int(torch.searchsorted(total, torch.tensor(budget, dtype=total.dtype)))     # first span that reaches the budget
            lengths[last] = budget - (0 if last == 0 else int(total[last - 1]))                # ... trimmed to it
            lengths = lengths[: last + 1]
            drawn = last + 1
            lengths = lengths[lengths > 0]                          # 0-length spans are INSERTIONS, handled at the end
            inserts = drawn - lengths.size(0)
            n_spans = lengths.size(0)
            if n_spans == 0:
                self._degenerate = True
                return self.insertion(source, inserts / source.size(0))
        else:
            n_spans = budget
            lengths = torch.ones((n_spans,)).long()
        candidates = starts01.nonzero(as_tuple=False)
        first = candidates[torch.randperm(candidates.size(0))[:n_spans]].squeeze(1)      # span starts (word starts, random order)
        as_random = torch.FloatTensor(n_spans).uniform_() < self.random_ratio
        assert n - 1 not in first
        # span ends in closed form: running count of word starts; the final </s> is never part of a span
        count = torch.cumsum((starts01[: n - 1] != 0).long(), 0)
        if self.span_lengths is not None:
            limit = count[first] + (lengths[: first.size(0)] - 1)
        else:
            limit = count[first]                                    # (whole-word masking: to the end of the start's own word)
        last_tok = torch.searchsorted(count, limit, right=True) - 1                   # [spans]: last token of each span
        extent = last_tok - first                                   # tokens behind the start that belong to the span
        keep = torch.ones(n, dtype=torch.bool)
        if self.replace_length == 0:
            keep[first] = False
        else:
            source[first] = self.mask_idx
            source[first[as_random]] = self._random_tokens(int(as_random.sum()))
        # the tokens behind each start, offset by offset (the reference walks all live spans one token at a time; with
        # replace_length -1 it draws the random replacements of an offset's live spans in ONE randint call, in span order)
        deepest = int(extent.max()) if extent.numel() else 0
        for k in range(1, deepest + 1):
            live = extent >= k
            idx = first[live] + k
            if self.replace_length != -1:
                keep[idx] = False
            else:
                source[idx] = self.mask_idx
                rnd = as_random[live]
                source[idx[rnd]] = self._random_tokens(int(rnd.sum()))
        source = source[keep]
        if inserts > 0:
            source = self.insertion(source, inserts / source.size(0))
        return source

    def insertion(self, tokens, p):
        """(:413-433) n = ceil(len * p) extra tokens at random inner positions: the first ceil(n * random_ratio) of them random
        tokens, the rest <mask>."""
        if p == 0.0:
            return tokens
        length = len(tokens)
        n = int(math.ceil(length * p))
        where = torch.randperm(length + n - 2)[:n] + 1
        is_noise = torch.zeros(length + n, dtype=torch.bool)
        is_noise[where] = True
        out = torch.full((length + n,), -1, dtype=torch.long)
        n_random = int(math.ceil(n * self.random_ratio))
        out[where[n_random:]] = self.mask_idx
        out[where[:n_random]] = torch.randint(low=1, high=self.V, size=(n_random,))
        out[~is_noise] = tokens
        assert bool((out >= 0).all())
        return out

    def rolling(self, tokens):
        """(:405-411) rotate the inner tokens by a numpy-drawn offset (the one noise step that uses numpy's stream)."""
        offset = np.random.randint(1, max(1, tokens.size(-1) - 1) + 1)
        return torch.cat((tokens[0:1], tokens[offset:-1], tokens[1:offset], tokens[-1:]), dim=0)

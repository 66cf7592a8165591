"""The speech-pretraining collater of the reference input pipeline on the GPU ("input pipeline on GPU"):
SpeechPretrainDataset.collater of /root/reference/SpeechT5/speecht5/data/speech_dataset.py:302-386 (+ collater_audio :388-407,
crop_to_max_size :290-300, collater_frm_label :409-425) with the items resident in HBM -- raw waveforms (their log-mel targets come
from speecht5_amd.features.LogMelFilterBank, the GPU form of :142-181), frame labels, speaker embeddings.

What stays on the host is the part that IS host state in the reference: the lengths of the items (Python ints) and the crop starts,
drawn from numpy's stream with the reference's own call -- one np.random.randint(0, diff + 1) per item longer than the batch's
audio size, in item order -- so a run draws the same crops as the reference for the same seed.  Every tensor of the batch is then a
ragged gather on the device (csrc/elementwise.hip st5_ragged_rows: crop at a per-item offset, stride r for the reduction factor,
shift by one frame for the decoder input, pad to the longest) plus two tail masks (st5_tail_mask: padding_mask, stop-token labels).
No tensor passes through host memory; 9 launches per batch.  Returns the reference's batch dictionary.
tests/test_collate_gpu.py: bit-identical to batches of the verbatim reference collater (tests/golden/collate_speech_pretrain.pt)."""
import numpy as np
import torch

from . import hip


def _i32(vals, device):
    return torch.tensor([int(v) for v in vals], dtype=torch.int32).to(device, non_blocking=True)


def _ptrs(tensors, device):
    return torch.tensor([t.data_ptr() for t in tensors], dtype=torch.int64).to(device, non_blocking=True)


def _ragged(items, off, hi, B, T, w, step, tmin, dtype, pad, device):
    """out[b, t, :w] = items[b][off[b] + t*step, :w] if t >= tmin and 0 <= off[b] + t*step < hi[b] else pad."""
    out = torch.empty((B, T) if w == 1 else (B, T, w), dtype=dtype, device=device)
    es = out.element_size()
    pad_bits = int(torch.tensor([pad], dtype=dtype).view(torch.int64 if es == 8 else torch.int32 if es == 4 else torch.uint8).item()) & ((1 << (8 * es)) - 1)
    keep = (_ptrs(items, device), _i32(off, device), _i32(hi, device))
    hip.check(hip.lib().st5_ragged_rows(keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), out.data_ptr(), B, T, w, step, tmin, es,
                                        pad_bits, hip.stream()), "st5_ragged_rows")
    out._st5_keep = keep + (items,)      # the descriptor tables (and the items) live until the kernel has run on this stream
    return out


def _tail(n, B, T, as_float, device):
    out = torch.empty(B, T, dtype=torch.float32 if as_float else torch.bool, device=device)
    nd = _i32(n, device)
    hip.check(hip.lib().st5_tail_mask(nd.data_ptr(), out.data_ptr(), B, T, 1 if as_float else 0, hip.stream()), "st5_tail_mask")
    out._st5_keep = nd
    return out


class SpeechPretrainCollater:
    """collater(samples) -> batch.  samples: [{"id", "source": fp32 [S] (device), "target": fp32 [L, odim] (device) or None,
    "label_list": [int64 [n] (device)], "spkembs": fp32 [D] (device)}].  target None: computed here from the waveform on the GPU
    (features.LogMelFilterBank; what the reference's __getitem__ does per item with librosa on the host, :249-260)."""

    def __init__(self, device, *, sample_rate=16000, label_rates=(50.0,), pad_list=(1,), max_sample_size=250000, pad_audio=False,
                 random_crop=True, reduction_factor=2, single_target=False, logmel=None):
        self.device = device
        self.sample_rate, self.label_rates, self.pad_list = sample_rate, list(label_rates), list(pad_list)
        self.max_sample_size = max_sample_size if max_sample_size is not None else (1 << 62)
        self.pad_audio, self.random_crop, self.reduction_factor, self.single_target = pad_audio, random_crop, reduction_factor, single_target
        self._logmel = logmel
        if any(r == -1.0 for r in self.label_rates):
            raise NotImplementedError("sequence-level labels (label_rate -1) are not part of the pre-training recipe")

    def logmel(self, wav):
        if self._logmel is None:
            from .features import LogMelFilterBank
            self._logmel = LogMelFilterBank(self.device)
        return self._logmel(wav.view(1, -1))[0]

    def collater(self, samples):
        samples = [s for s in samples if s["source"] is not None]
        if not samples:
            return {}
        dev, B = self.device, len(samples)
        audios = [s["source"].contiguous() for s in samples]
        fbanks = [(s["target"] if s.get("target") is not None else self.logmel(s["source"])).contiguous() for s in samples]
        assert all(a.dtype == torch.float32 and a.is_cuda for a in audios) and all(f.dtype == torch.float32 for f in fbanks)
        audio_sizes = [int(a.shape[0]) for a in audios]
        fbank_sizes = [int(f.shape[0]) for f in fbanks]
        odim = int(fbanks[0].shape[1])
        audio_size = min(max(audio_sizes), self.max_sample_size) if self.pad_audio else min(min(audio_sizes), self.max_sample_size)
        # crop starts: the reference's draws, in its order (collater_audio -> crop_to_max_size)
        starts = []
        for n in audio_sizes:
            diff = n - audio_size
            assert diff >= 0 or self.pad_audio
            starts.append(int(np.random.randint(0, diff + 1)) if (diff > 0 and self.random_crop) else 0)
        source = _ragged(audios, starts, audio_sizes, B, audio_size, 1, 1, 0, torch.float32, 0.0, dev)
        padding_mask = _tail(audio_sizes, B, audio_size, False, dev)
        # log-mel targets: the frames that belong to the cropped audio (:325-331), padded to the longest
        f_lo, f_hi = [], []
        for i in range(B):
            ratio = audio_sizes[i] / fbank_sizes[i]
            fs = int(starts[i] / ratio)
            f_lo.append(fs)
            f_hi.append(min(fs + int(audio_size / ratio), fbank_sizes[i]))
        dec_len = [hi_ - lo_ for lo_, hi_ in zip(f_lo, f_hi)]
        Lmax = max(dec_len)
        dec_target = _ragged(fbanks, f_lo, f_hi, B, Lmax, odim, 1, 0, torch.float32, 0.0, dev)
        dec_lengths = torch.tensor(dec_len, dtype=torch.long).to(dev, non_blocking=True)
        r = self.reduction_factor
        # decoder input: every r-th frame (frames r-1, 2r-1, ...), shifted right by one with a zero first frame (:336-344):
        # prev[b, j] = dec_target[b, j*r - 1] for j >= 1 (padded rows of dec_target are zeros: hi = the item's cut end)
        Lin = len(range(r - 1, Lmax, r)) if r > 1 else Lmax
        prev = _ragged(fbanks, [lo_ - 1 for lo_ in f_lo], f_hi, B, Lin, odim, r, 1, torch.float32, 0.0, dev)
        tgt_lengths = torch.div(dec_lengths, r, rounding_mode="floor") if r > 1 else dec_lengths
        labels = _tail([l - 1 for l in fbank_sizes], B, Lmax, True, dev)      # (:347-349: from the item's UNCROPPED last frame on)
        spk = [s["spkembs"].contiguous() for s in samples]
        spkembs = _ragged(spk, [0] * B, [int(v.shape[0]) for v in spk], B, max(int(v.shape[0]) for v in spk), 1, 1, 0, torch.float32, 0.0, dev)
        targets_list, lengths_list, ntokens_list = [], [], []
        for li, (rate, pad) in enumerate(zip(self.label_rates, self.pad_list)):      # collater_frm_label :409-425
            labs = [s["label_list"][li].contiguous() for s in samples]
            assert all(t.dtype == torch.int64 for t in labs)
            s2f = rate / self.sample_rate
            frm_starts = [int(round(s * s2f)) for s in starts]
            frm_size = int(round(audio_size * s2f))
            if not self.pad_audio:
                frm_size = min(frm_size, *[int(t.shape[0]) - s for t, s in zip(labs, frm_starts)])
            lens = [max(0, min(int(t.shape[0]) - s, frm_size)) for t, s in zip(labs, frm_starts)]
            hi_ = [s + n for s, n in zip(frm_starts, lens)]
            targets_list.append(_ragged(labs, frm_starts, hi_, B, max(lens), 1, 1, 0, torch.int64, int(pad), dev))
            lengths_list.append(torch.tensor(lens, dtype=torch.long).to(dev, non_blocking=True))
            ntokens_list.append(int(sum(lens)))
        net_input = {"source": source, "padding_mask": padding_mask, "prev_output_tokens": prev, "spkembs": spkembs, "tgt_lengths": tgt_lengths}
        batch = {"id": torch.LongTensor([s["id"] for s in samples]), "net_input": net_input, "labels": labels, "dec_target": dec_target,
                 "dec_target_lengths": dec_lengths, "src_lengths": [audio_size] * B, "task_name": "speech_pretrain"}
        if self.single_target:
            batch["target_lengths"], batch["ntokens"], batch["target"] = lengths_list[0], ntokens_list[0], targets_list[0]
        else:
            batch["target_lengths_list"], batch["ntokens_list"], batch["target_list"] = lengths_list, ntokens_list, targets_list
        return batch

    __call__ = collater


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 6: the remaining collaters of the reference input pipeline: text-to-speech fine-tuning, speech-to-text
# fine-tuning, text pre-training (+ its BART noise, speecht5_amd/text_noise.py).  Same design: lengths are host ints, every tensor of
# the batch is a ragged gather over items resident in HBM, nothing passes through host memory.
# ---------------------------------------------------------------------------------------------------------------------------------
def _lengths(n, device):
    return torch.tensor([int(v) for v in n], dtype=torch.long).to(device, non_blocking=True)


def _pad_tokens(items, lens, B, T, pad, device, order=None, shift=False):
    """fairseq.data.data_utils.collate_tokens(values, pad, left_pad=False) as one gather: out[b, t] = items[b][t] (t < len) else pad;
    shift=True is its move_eos_to_beginning form with eos_idx=None: out[b, 0] = the item's LAST token, out[b, 1:] = items[b][:-1]."""
    if order is not None:
        items, lens = [items[i] for i in order], [lens[i] for i in order]
    if not shift:
        return _ragged(items, [0] * B, lens, B, T, 1, 1, 0, torch.int64, int(pad), device)
    out = _ragged(items, [-1] * B, [l - 1 for l in lens], B, T, 1, 1, 1, torch.int64, int(pad), device)
    first = _ragged(items, [l - 1 for l in lens], lens, B, 1, 1, 1, 0, torch.int64, int(pad), device)
    out[:, 0] = first[:, 0]       # (one strided copy: glue)
    out._st5_keep = out._st5_keep + (first,)
    return out


class TextToSpeechCollater:
    """TextToSpeechDataset.collater (data/text_to_speech_dataset.py:228-281) + _collate_frames (:27-45) + collater_label (:289-298).
    samples: [{"id", "audio_name", "source": [int64 [n] (device)], "target": fp32 [L, odim] (device), "spkembs": fp32 [D] (device)}]."""

    def __init__(self, device, *, pad_idx=1, reduction_factor=2):
        self.device, self.pad_idx, self.reduction_factor = device, pad_idx, reduction_factor

    def collater(self, samples):
        samples = [s for s in samples if s["source"] is not None]
        if not samples:
            return {}
        dev, B, r = self.device, len(samples), self.reduction_factor
        fbanks = [s["target"].contiguous() for s in samples]
        sizes = [int(f.shape[0]) for f in fbanks]
        odim, Lmax = int(fbanks[0].shape[1]), max(sizes)
        dec_target = _ragged(fbanks, [0] * B, sizes, B, Lmax, odim, 1, 0, torch.float32, 0.0, dev)
        dec_lengths = _lengths(sizes, dev)
        # decoder input (:239-248): frames r-1, 2r-1, ... of the PADDED batch, shifted right by one behind a zero frame
        Lin = len(range(r - 1, Lmax, r)) if r > 1 else Lmax
        prev = _ragged(fbanks, [-1] * B, sizes, B, Lin, odim, r, 1, torch.float32, 0.0, dev)
        tgt_lengths = _lengths([l // r for l in sizes], dev) if r > 1 else dec_lengths
        labels = _tail([l - 1 for l in sizes], B, Lmax, True, dev)                 # (:251-253: 1.0 from the last frame on)
        spk = [s["spkembs"].contiguous() for s in samples]
        spkembs = _ragged(spk, [0] * B, [int(v.shape[0]) for v in spk], B, max(int(v.shape[0]) for v in spk), 1, 1, 0, torch.float32, 0.0, dev)
        toks = [s["source"][0].contiguous() for s in samples]
        tlen = [int(t.shape[0]) for t in toks]
        src_tokens = _pad_tokens(toks, tlen, B, max(tlen), self.pad_idx, dev)
        src_lengths = _lengths(tlen, dev)
        net_input = {"src_tokens": src_tokens, "src_lengths": src_lengths, "prev_output_tokens": prev, "tgt_lengths": tgt_lengths,
                     "spkembs": spkembs, "task_name": "t2s"}
        return {"id": torch.LongTensor([s["id"] for s in samples]), "name": [s.get("audio_name") for s in samples], "net_input": net_input,
                "labels": labels, "dec_target": dec_target, "dec_target_lengths": dec_lengths, "src_lengths": src_lengths, "task_name": "t2s",
                "ntokens": int(sum(tlen)), "target": dec_target}

    __call__ = collater


class SpeechToTextCollater:
    """SpeechToTextDataset.collater (data/speech_to_text_dataset.py:150-207) + collater_audio (:209-224) + collater_label (:232-241).
    samples: [{"id", "source": fp32 [S] (device), "label_list": [int64 [n] (device)]}]."""

    def __init__(self, device, *, pad_idx=1, eos_idx=2):
        self.device, self.pad_idx, self.eos_idx = device, pad_idx, eos_idx

    def collater(self, samples):
        samples = [s for s in samples if s["source"] is not None]
        if not samples:
            return {}
        dev, B = self.device, len(samples)
        audios = [s["source"].contiguous() for s in samples]
        sizes = [int(a.shape[0]) for a in audios]
        S = max(sizes)
        source = _ragged(audios, [0] * B, sizes, B, S, 1, 1, 0, torch.float32, 0.0, dev)
        padding_mask = _tail(sizes, B, S, False, dev)
        labs = [s["label_list"][0].contiguous() for s in samples]
        n = [int(t.shape[0]) for t in labs]
        T = max(n) + 1
        rows = torch.arange(B, device=dev)
        nd = _lengths(n, dev)
        # decoder target (:167-181): label + </s>, right-padded; decoder input (:183-189): </s> moved to the front
        target = _pad_tokens(labs, n, B, T, self.pad_idx, dev)
        target[rows, nd] = self.eos_idx
        prev = _ragged(labs, [-1] * B, n, B, T, 1, 1, 1, torch.int64, self.pad_idx, dev)
        prev[:, 0] = self.eos_idx
        net_input = {"source": source, "padding_mask": padding_mask, "prev_output_tokens": prev, "task_name": "s2t"}
        return {"id": torch.LongTensor([s["id"] for s in samples]), "net_input": net_input, "target": target,
                "target_lengths": _lengths([v + 1 for v in n], dev), "task_name": "s2t", "ntokens": int(sum(n))}

    __call__ = collater


class TextPretrainCollater:
    """collate() of data/text_dataset.py:18-99 (TextPretrainDataset.collater :435-444): source / target right-padded, the batch sorted by
    descending SOURCE length (the reference's own torch sort on the host lengths: the tie order is part of the batch), the decoder
    input = target with its last token moved to the front.  samples: [{"id", "source": int64 [n] (device), "target": int64 [m] (device)}],
    e.g. from text_noise.BartNoise (the noise draws are host state in the reference: torch's CPU generator)."""

    def __init__(self, device, *, pad_idx=1):
        self.device, self.pad_idx = device, pad_idx

    def collater(self, samples):
        if not samples:
            return {}
        dev, B = self.device, len(samples)
        src = [s["source"].contiguous() for s in samples]
        slen = [int(t.numel()) for t in src]
        src_lengths, sort_order = torch.LongTensor(slen).sort(descending=True)
        order = sort_order.tolist()
        ids = torch.LongTensor([s["id"] for s in samples]).index_select(0, sort_order)
        src_tokens = _pad_tokens(src, slen, B, max(slen), self.pad_idx, dev, order=order)
        batch = {"id": ids, "net_input": {"src_tokens": src_tokens, "src_lengths": src_lengths.to(dev, non_blocking=True)},
                 "nsentences": int(samples[0]["source"].size(0)), "sort_order": sort_order, "task_name": "text_pretrain"}
        if samples[0].get("target") is not None:
            tgt = [s["target"].contiguous() for s in samples]
            tlen = [int(t.numel()) for t in tgt]
            batch["target"] = _pad_tokens(tgt, tlen, B, max(tlen), self.pad_idx, dev, order=order)
            batch["net_input"]["prev_output_tokens"] = _pad_tokens(tgt, tlen, B, max(tlen), self.pad_idx, dev, order=order, shift=True)
            batch["ntokens"] = int(sum(tlen))
        else:
            batch["target"] = None
            batch["ntokens"] = int(sum(slen))
        return batch

    __call__ = collater

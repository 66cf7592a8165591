"""CPU restatement (plain torch / numpy, TEST INFRASTRUCTURE ONLY) of the speech-pretraining collater of the reference input pipeline:
/root/reference/SpeechT5/speecht5/data/speech_dataset.py  SpeechPretrainDataset.collater :302-386, collater_audio :388-407,
crop_to_max_size :290-300, collater_frm_label :409-425 (+ fairseq.data.data_utils.collate_tokens, third party: right-pad with pad_idx).
Pinned against batches of the VERBATIM reference collater: tests/golden/collate_speech_pretrain.pt (oracle/make_golden_collate.py),
tests/test_collate_cpu.py.  The product counterpart is speecht5_amd/collate.py (ragged gathers on the GPU)."""
import numpy as np
import torch


def crop_starts(audio_sizes, audio_size, random_crop):
    """:388-407 / :290-300 -- one np.random.randint(0, diff + 1) per item that is LONGER than audio_size, in item order."""
    starts = []
    for n in audio_sizes:
        diff = n - audio_size
        starts.append(int(np.random.randint(0, diff + 1)) if (diff > 0 and random_crop) else 0)
    return starts


def collate_speech_pretrain(samples, *, pad_audio, random_crop, max_sample_size, reduction_factor, sample_rate, label_rate, pad_idx):
    audios = [s["source"] for s in samples]
    fbanks = [s["target"] for s in samples]
    audio_sizes = [len(a) for a in audios]
    fbank_sizes = [len(f) for f in fbanks]
    audio_size = min(max(audio_sizes), max_sample_size) if pad_audio else min(min(audio_sizes), max_sample_size)      # :317-320
    B = len(samples)
    starts = crop_starts(audio_sizes, audio_size, random_crop)
    source = audios[0].new_zeros(B, audio_size)
    padding_mask = torch.zeros(B, audio_size, dtype=torch.bool)
    for i, a in enumerate(audios):                                                                                    # :394-406
        n = min(audio_sizes[i] - starts[i], audio_size)
        source[i, :n] = a[starts[i]:starts[i] + n]
        if audio_sizes[i] < audio_size:
            assert pad_audio
            padding_mask[i, audio_sizes[i]:] = True
    cut = []
    for i in range(B):                                                                                                # :325-331
        ratio = audio_sizes[i] / fbank_sizes[i]
        fs = int(starts[i] / ratio)
        fe = min(fs + int(audio_size / ratio), fbank_sizes[i])
        cut.append(fbanks[i][fs:fe])
    dec_lengths = torch.tensor([len(c) for c in cut], dtype=torch.long)
    Lmax = int(dec_lengths.max())
    dec_target = fbanks[0].new_zeros(B, Lmax, fbanks[0].size(1))
    for i, c in enumerate(cut):
        dec_target[i, :len(c)] = c
    r = reduction_factor
    if r > 1:                                                                                                         # :336-340
        thin = dec_target[:, r - 1::r]
        tgt_lengths = torch.div(dec_lengths, r, rounding_mode="floor")
    else:
        thin, tgt_lengths = dec_target, dec_lengths
    prev = torch.cat([thin.new_zeros(B, 1, thin.shape[2]), thin[:, :-1]], dim=1)                                      # :342-344
    labels = dec_target.new_zeros(B, Lmax)
    for i, l in enumerate(fbank_sizes):                                                                               # :347-349 (the UNCROPPED length)
        labels[i, l - 1:] = 1.0
    spk = torch.stack([s["spkembs"] for s in samples])
    s2f = label_rate / sample_rate                                                                                    # :409-425
    frm_starts = [int(round(s * s2f)) for s in starts]
    frm_size = int(round(audio_size * s2f))
    labs = [s["label_list"][0] for s in samples]
    if not pad_audio:
        frm_size = min(frm_size, *[len(t) - s for t, s in zip(labs, frm_starts)])
    labs = [t[s:s + frm_size] for t, s in zip(labs, frm_starts)]
    tlen = torch.LongTensor([len(t) for t in labs])
    target = labs[0].new_full((B, int(tlen.max())), pad_idx)
    for i, t in enumerate(labs):
        target[i, :len(t)] = t
    return {"source": source, "padding_mask": padding_mask, "prev_output_tokens": prev, "spkembs": spk, "tgt_lengths": tgt_lengths,
            "labels": labels, "dec_target": dec_target, "dec_target_lengths": dec_lengths, "target": target, "target_lengths": tlen,
            "ntokens": int(tlen.sum()), "src_lengths": [audio_size] * B, "id": torch.LongTensor([s["id"] for s in samples])}

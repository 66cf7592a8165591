"""Host-side helpers the hot path needs from fairseq.data.data_utils (numpy, CPU): span-mask sampling
and length masks.  Same numpy RNG call sequence as the fairseq implementation the reference calls at
speech_encoder_prenet.py:237-247, so a seeded run draws the same masks."""
import numpy as np
import torch


def lengths_to_padding_mask(lens):
    bsz, max_lens = lens.size(0), int(torch.max(lens))
    ar = torch.arange(max_lens, device=lens.device).view(1, max_lens).expand(bsz, -1)
    return ar >= lens.view(bsz, 1).expand(-1, max_lens)


def compute_mask_indices(shape, padding_mask, mask_prob, mask_length, mask_type="static", mask_other=0.0, min_masks=0,
                         no_overlap=False, min_space=0, require_same_masks=True, mask_dropout=0.0):
    """Random span mask [B, T] (bool ndarray).  Span count per row = int(p*T/len + U[0,1)) (probabilistic
    rounding); span starts drawn without replacement; rows are trimmed to the common minimum count."""
    if no_overlap:
        raise NotImplementedError("no_mask_overlap is not used by the SpeechT5 recipes")
    bsz, all_sz = shape
    mask = np.full((bsz, all_sz), False)
    all_num_mask = max(min_masks, int(mask_prob * all_sz / float(mask_length) + np.random.rand()))
    per_row = []
    for i in range(bsz):
        if padding_mask is not None:
            sz = all_sz - int(padding_mask[i].long().sum().item())
            num_mask = max(min_masks, int(mask_prob * sz / float(mask_length) + np.random.rand()))
        else:
            sz, num_mask = all_sz, all_num_mask
        if mask_type == "static":
            lengths = np.full(num_mask, mask_length)
        elif mask_type == "uniform":
            lengths = np.random.randint(mask_other, mask_length * 2 + 1, size=num_mask)
        elif mask_type == "normal":
            lengths = [max(1, int(round(x))) for x in np.random.normal(mask_length, mask_other, size=num_mask)]
        elif mask_type == "poisson":
            lengths = [int(round(x)) for x in np.random.poisson(mask_length, size=num_mask)]
        else:
            raise Exception("unknown mask selection " + mask_type)
        if sum(lengths) == 0:
            lengths[0] = min(mask_length, sz - 1)
        min_len = min(lengths)
        if sz - min_len <= num_mask:
            min_len = sz - num_mask - 1
        starts = np.random.choice(sz - min_len, num_mask, replace=False)
        idc = np.asarray([starts[j] + off for j in range(len(starts)) for off in range(lengths[j])])
        per_row.append(np.unique(idc[idc < sz]))
    min_len = min(len(m) for m in per_row)
    for i, idc in enumerate(per_row):
        if len(idc) > min_len and require_same_masks:
            idc = np.random.choice(idc, min_len, replace=False)
        if mask_dropout > 0:
            holes = np.rint(len(idc) * mask_dropout).astype(int)
            idc = np.random.choice(idc, len(idc) - holes, replace=False)
        mask[i, idc] = True
    return mask

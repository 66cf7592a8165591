"""Span-mask configurations shared by oracle/make_golden_masks.py (reference run) and tests/test_masks_cpu.py (product run):
the recipes' settings (pre-training, fine-tuning, channel mask) and the other branches of compute_mask_indices.  Test infrastructure."""
import torch

# (shape, padded frames per row or None, keyword arguments) -- the recipes' settings and the other branches of the function
CASES = dict(
    pretrain=dict(shape=(8, 499), pad=None, kw=dict(mask_prob=0.8, mask_length=10, mask_type="static", mask_other=0.0, min_masks=2,
                                                    no_overlap=False, min_space=1)),
    pretrain_ragged=dict(shape=(6, 499), pad=[0, 37, 120, 0, 250, 499 - 25], kw=dict(mask_prob=0.8, mask_length=10, mask_type="static",
                                                                                  mask_other=0.0, min_masks=2, no_overlap=False, min_space=1)),
    finetune=dict(shape=(4, 199), pad=[0, 10, 0, 60], kw=dict(mask_prob=0.5, mask_length=10, mask_type="static", mask_other=0.0,
                                                            min_masks=2, no_overlap=False, min_space=1)),
    channel=dict(shape=(4, 768), pad=None, kw=dict(mask_prob=0.5, mask_length=64, mask_type="static", mask_other=0.0, no_overlap=False,
                                                   min_space=1)),
    uniform=dict(shape=(5, 300), pad=[0, 0, 33, 0, 100], kw=dict(mask_prob=0.4, mask_length=8, mask_type="uniform", mask_other=2.0,
                                                                min_masks=1, no_overlap=False, min_space=1)),
    normal=dict(shape=(5, 300), pad=None, kw=dict(mask_prob=0.4, mask_length=8, mask_type="normal", mask_other=2.0, min_masks=1,
                                                  no_overlap=False, min_space=1)),
    poisson=dict(shape=(3, 250), pad=[0, 20, 0], kw=dict(mask_prob=0.3, mask_length=6, mask_type="poisson", mask_other=0.0, min_masks=0,
                                                         no_overlap=False, min_space=1)),
    short_rows=dict(shape=(4, 40), pad=[0, 30, 35, 10], kw=dict(mask_prob=0.65, mask_length=10, mask_type="static", mask_other=0.0,
                                                                min_masks=2, no_overlap=False, min_space=1)),
)


def padding(case):
    if case["pad"] is None:
        return None
    B, T = case["shape"]
    m = torch.zeros(B, T, dtype=torch.bool)
    for b, n in enumerate(case["pad"]):
        if n:
            m[b, T - n:] = True
    return m

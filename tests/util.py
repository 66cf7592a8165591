"""Shared test helpers: golden fixtures, tiny model construction, randomness injection."""
import contextlib
import os
from argparse import Namespace
from types import SimpleNamespace

import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Dictionary(list):
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
        try:
            return list.index(self, sym)
        except ValueError:   # fairseq Dictionary.index: unknown symbol -> <unk>
            return self.unk()


class Task:
    def __init__(self, vocab=30, n_units=20):
        self.dicts = {"text": Dictionary(vocab, ["<mask>", "<ctc_blank>"]), "hubert": [Dictionary(n_units)]}
        self.t5_task = "pretrain"
        self.target_dictionary = self.dicts["text"]
        self.blank_symbol_idx = self.dicts["text"].index("<ctc_blank>")


def load_golden(name):
    m = torch.load(os.path.join(G, "tiny_model.pt"), weights_only=False)
    fx = torch.load(os.path.join(G, name), weights_only=False) if name else None
    return m, fx


def build_tiny(device, dtype=torch.float32):
    """Our T5TransformerModel with the golden tiny configuration and the reference's weights."""
    from speecht5_amd import functional as Fn
    from speecht5_amd.speecht5 import T5TransformerModel
    m, _ = load_golden(None)
    args = Namespace(**m["args"])
    Fn.set_compute_dtype(dtype)
    model = T5TransformerModel.build_model(args, Task())
    missing, unexpected = torch.nn.Module.load_state_dict(model, m["state_dict"], strict=True), None
    return model.to(device), args


def to_dev(obj, device):
    if isinstance(obj, torch.Tensor):
        return obj.to(device)
    if isinstance(obj, dict):
        return {k: to_dev(v, device) for k, v in obj.items()}
    if isinstance(obj, list):
        return [to_dev(v, device) for v in obj]
    return obj


@contextlib.contextmanager
def injected_randomness(model, mask_indices=None, mix_idx=None, gumbel_noise=None, tau=2.0):
    """Replays the random draws recorded from the reference run inside our model."""
    import speecht5_amd.modules.speech_encoder_prenet as sep
    import speecht5_amd.speecht5 as st5
    old_cmi, old_rp = sep.compute_mask_indices, torch.randperm
    qz = getattr(model, "quantizer", None)
    old_noise, old_temp = (qz.gumbel_noise, qz.curr_temp) if qz is not None else (None, None)
    if mask_indices is not None:
        sep.compute_mask_indices = lambda *a, **k: mask_indices.cpu().numpy()
    if mix_idx is not None:
        def randperm(n, *a, **k):
            rest = torch.tensor([i for i in range(n) if i not in set(mix_idx.tolist())], dtype=torch.long)
            return torch.cat([mix_idx.cpu().long(), rest])
        torch.randperm = randperm
    if gumbel_noise is not None:
        def noise(logits):
            g = gumbel_noise.to(logits.device)
            return g.expand(logits.shape).contiguous() if g.numel() == 1 else g.reshape(logits.shape)
        qz.gumbel_noise = noise
        qz.curr_temp = tau
    try:
        yield
    finally:
        sep.compute_mask_indices = old_cmi
        torch.randperm = old_rp
        if qz is not None:
            qz.gumbel_noise, qz.curr_temp = old_noise, old_temp


def close(a, b, tol, what="", floor=0.0):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    fin = torch.isfinite(b)
    assert torch.equal(torch.isfinite(a), fin), f"{what}: non-finite pattern differs"
    s = max(b[fin].abs().max().item(), 1e-6) if fin.any() else 1.0
    err = (a[fin] - b[fin]).abs().max().item() if fin.any() else 0.0
    assert err <= tol * s + floor, f"{what}: max err {err:.3e} vs scale {s:.3e} (tol {tol})"


TIED = ["text_encoder_prenet.encoder_prenet.0.weight", "text_decoder_prenet.embed_tokens.weight",
        "text_decoder_postnet.output_projection.weight"]


def check_grads(model, fx, tol):
    got = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    # named_parameters de-duplicates the tied embedding: expose it under every alias
    for n, p in model.state_dict(keep_vars=True).items():
        if isinstance(p, torch.nn.Parameter) and p.grad is not None:
            got[n] = p.grad
    ref = fx["grads"]
    total = sum(v * v for v in ref["norms"].values()) ** 0.5
    checked = 0
    for name, g in ref["full"].items():
        if ref["norms"][name] <= 1e-7 * total:
            assert name not in got or float(got[name].double().norm()) <= 1e-5 * total, name
            continue
        close(got[name], g, tol, what=f"grad {name}", floor=2e-7 * total)
        checked += 1
    for name, g in ref["rows"].items():
        close(got[name].reshape(got[name].shape[0], -1)[:8], g, tol, what=f"grad rows {name}", floor=2e-7 * total)
        rn = ref["norms"][name]
        assert abs(float(got[name].double().norm()) - rn) <= 10 * tol * rn + 1e-8, f"grad norm {name}"
        checked += 1
    assert checked > 40
    return checked


@contextlib.contextmanager
def poisoned_allocations():
    """Every GPU buffer handed out by torch.empty / torch.empty_like / Tensor.new_empty (what speecht5_amd allocates its outputs,
    scratch tensors and workspaces with) is filled with 0xFF bytes first: NaN as bf16 / fp32, -1 as an index.  A kernel that
    consumes memory it (or its producer) never wrote then yields NaN / a fault instead of a silent run-to-run difference.
    (Together with ST5_POISON=1 for the library's own hipMalloc'ed arenas, read when the library first allocates.)"""
    o_empty, o_like, o_new = torch.empty, torch.empty_like, torch.Tensor.new_empty

    def poison(t):
        if t.is_cuda and t.numel() and t.is_contiguous():
            t.view(-1).view(torch.uint8).fill_(0xFF)
        return t

    torch.empty = lambda *a, **k: poison(o_empty(*a, **k))
    torch.empty_like = lambda *a, **k: poison(o_like(*a, **k))
    torch.Tensor.new_empty = lambda self, *a, **k: poison(o_new(self, *a, **k))
    try:
        yield
    finally:
        torch.empty, torch.empty_like, torch.Tensor.new_empty = o_empty, o_like, o_new


# ONE bar for the bf16 gradients of the mel post-net's parameters (conv + BatchNorm + tanh stack) in every test that compares the
# bf16 compute mode with the fp32 oracle / reference goldens.  The BatchNorm backward cancels to ~1e-2 of its terms, so bf16
# operand rounding costs about two digits there; how much depends on the batch statistics' sample size.  Measured minima (round 4,
# MI355X): 0.9934 at the benched shape (tests/test_cfg2_shape_gpu.py, 2 x 10 s), 0.9743 on the 2 s clip of
# tests/test_fullsize_gpu.py, 0.9648 on the tiny Large-style golden (tests/test_large_gpu.py).  Everything outside the post-net is
# held to 0.98-0.999 by the tests themselves.
BF16_POST_COS = 0.96

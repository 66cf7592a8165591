"""Mel post-net block (functional.PostnetFunction: implicit-GEMM convolutions + st5_batchnorm_act_*) against plain torch
(espnet Tacotron Postnet semantics as the reference uses it, speech_decoder_postnet.py:39-51,65-70): forward `after`, running
statistics, and every gradient; fp32 parity mode at fp32 round-off, bf16 compute mode at 2e-2.  Dropout masks are taken from the
library's own counter RNG (st5_dropout on ones) so the dropped path is compared too."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from speecht5_amd import functional as Fn, hip  # noqa: E402
from speecht5_amd.modules.speech_decoder_postnet import Postnet  # noqa: E402


def _ref(before, net, masks, training):
    """fp32 torch reference on [B, L, C] input; masks: per-layer multiplicative dropout factors [B*L, C] or None."""
    x = before.transpose(1, 2)
    n = len(net.postnet)
    stats = []
    for i, blk in enumerate(net.postnet):
        conv, bn = blk[0], blk[1]
        x = F.conv1d(x, conv.weight, padding=(conv.weight.shape[2] - 1) // 2)
        if training:
            xf = x.transpose(1, 2).reshape(-1, x.shape[1])
            stats.append((xf.mean(0), xf.var(0, unbiased=True)))
        x = F.batch_norm(x, None if training else bn.running_mean, None if training else bn.running_var, bn.weight, bn.bias,
                         training=training, eps=bn.eps)
        if i < n - 1:
            x = torch.tanh(x)
        if masks is not None:
            x = x * masks[i].view(before.shape[0], before.shape[1], -1).transpose(1, 2)
    return before + x.transpose(1, 2), stats


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("training,p", [(True, 0.0), (True, 0.5), (False, 0.0)])
def test_postnet_matches_torch(cuda, dtype, training, p):
    torch.manual_seed(3)
    B, L, odim, ch = 3, 37, 80, 64
    Fn.set_compute_dtype(dtype)
    try:
        net = Postnet(0, odim, n_layers=5, n_chans=ch, n_filts=5, dropout_rate=p).to(cuda)
        with torch.no_grad():
            for blk in net.postnet:
                blk[1].weight.uniform_(0.5, 1.5); blk[1].bias.normal_(0, 0.2)
                blk[1].running_mean.normal_(0, 0.1); blk[1].running_var.uniform_(0.5, 1.5)
        net.train(training)
        rm0 = [blk[1].running_mean.clone() for blk in net.postnet]
        rv0 = [blk[1].running_var.clone() for blk in net.postnet]
        before = (torch.randn(B, L, odim) * 0.8).to(dtype).to(cuda).requires_grad_(True)
        dafter = torch.randn(B, L, odim, device=cuda)
        Fn.manual_seed(11)
        after = net(before)
        assert after.dtype == torch.float32
        after.backward(dafter)
        torch.cuda.synchronize()
        got = dict(after=after.detach(), dbefore=before.grad.float())
        for i, blk in enumerate(net.postnet):
            got[f"w{i}"] = blk[0].weight.grad.clone(); got[f"g{i}"] = blk[1].weight.grad.clone(); got[f"b{i}"] = blk[1].bias.grad.clone()
        # reference (same dropout masks: seeds are next_seed() of manual_seed(11), in layer order)
        masks = None
        if p > 0:
            Fn.manual_seed(11)
            masks = []
            for blk in net.postnet:
                C = blk[0].weight.shape[0]
                ones = torch.ones(B * L, C, device=cuda); m = torch.empty_like(ones)
                hip.check(hip.lib().st5_dropout(ones.data_ptr(), m.data_ptr(), ones.numel(), p, Fn.next_seed(), hip.F32, hip.stream()), "st5_dropout")
                masks.append(m)
        ref_net = Postnet(0, odim, n_layers=5, n_chans=ch, n_filts=5, dropout_rate=p).to(cuda)
        ref_net.load_state_dict(net.state_dict())
        with torch.no_grad():
            for i, blk in enumerate(ref_net.postnet):
                blk[1].running_mean.copy_(rm0[i]); blk[1].running_var.copy_(rv0[i])
                if dtype == torch.bfloat16:     # the kernels see bf16-rounded convolution operands
                    blk[0].weight.copy_(blk[0].weight.to(dtype).float())
        b32 = before.detach().float().clone().requires_grad_(True)
        ref_after, stats = _ref(b32, ref_net, masks, training)
        ref_after.backward(dafter)
        tol = 3e-5 if dtype == torch.float32 else 3e-2

        def close(a, b, what, t=tol):
            s = b.abs().max().item()
            e = (a.float() - b.float()).abs().max().item()
            assert e <= t * max(s, 1e-6), f"{what}: err {e:.3e} scale {s:.3e}"
        close(got["after"], ref_after.detach(), "after")
        close(got["dbefore"], b32.grad, "d before")
        for i, blk in enumerate(ref_net.postnet):
            close(got[f"w{i}"], blk[0].weight.grad, f"conv {i} weight grad")
            close(got[f"g{i}"], blk[1].weight.grad, f"bn {i} weight grad")
            close(got[f"b{i}"], blk[1].bias.grad, f"bn {i} bias grad")
        for i, blk in enumerate(net.postnet):
            if training:   # running statistics as torch.nn.BatchNorm1d updates them (momentum 0.1, unbiased variance)
                close(blk[1].running_mean, 0.9 * rm0[i] + 0.1 * stats[i][0], f"running_mean {i}", max(tol, 1e-4))
                close(blk[1].running_var, 0.9 * rv0[i] + 0.1 * stats[i][1], f"running_var {i}", max(tol, 1e-4))
                assert int(blk[1].num_batches_tracked) == 1
            else:
                assert torch.equal(blk[1].running_mean, rm0[i]) and int(blk[1].num_batches_tracked) == 0
    finally:
        Fn.set_compute_dtype(torch.float32)
        Fn.weight_cache.clear()

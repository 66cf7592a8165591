"""Fused (flash) attention kernels vs the unfused st5_gemm + st5_softmax path (same inputs, same dropout seed)
and vs plain torch fp32 math.  bf16 tolerance 2e-2 of the output scale."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from speecht5_amd import functional as Fn, hip  # noqa: E402


def _ref(q, k, v, pe, kpm, causal, maxrel):
    """fp32 torch reference on [B,H,T,hd] tensors."""
    B, H, T, hd = q.shape
    S = k.shape[2]
    s = torch.einsum("bhid,bhjd->bhij", q, k) * hd ** -0.5
    if pe is not None:
        i = torch.arange(T)[:, None]; j = torch.arange(S)[None, :]
        idx = (i - j).clamp(-maxrel, maxrel - 1) + maxrel
        qp = torch.einsum("bhid,nd->bhin", q, pe) * hd ** -0.5
        s = s + torch.gather(qp, 3, idx.expand(B, H, T, S))
    if causal:
        s = s + torch.triu(torch.full((T, S), float("-inf")), 1 + (S - T))
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :].bool(), float("-inf"))
    p = torch.softmax(s, -1)
    return torch.einsum("bhij,bhjd->bhid", p, v), torch.logsumexp(s, -1)


@pytest.mark.parametrize("T,S,causal,rel,pad", [(499, 499, 0, 1, 0), (70, 70, 0, 1, 1), (313, 313, 1, 0, 1), (130, 200, 0, 0, 1), (33, 33, 1, 0, 0)])
def test_flash_fwd_matches_unfused_and_reference(cuda, T, S, causal, rel, pad):
    torch.manual_seed(T * 3 + S)
    B, H, hd, maxrel = 2, 3, 64, 16 if T < 200 else 160
    d = H * hd
    dt = torch.bfloat16
    self_attn = T == S
    if self_attn:
        qkv = (torch.randn(B * T, 3 * d) * 1.5).to(dt).to(cuda)
        qv, kv_, vv = (qkv, 3 * d, 0), (qkv, 3 * d, d), (qkv, 3 * d, 2 * d)
        q4 = qkv.float().cpu().view(B, T, 3, H, hd)[:, :, 0].permute(0, 2, 1, 3)
        k4 = qkv.float().cpu().view(B, T, 3, H, hd)[:, :, 1].permute(0, 2, 1, 3)
        v4 = qkv.float().cpu().view(B, T, 3, H, hd)[:, :, 2].permute(0, 2, 1, 3)
    else:
        qt = (torch.randn(B * T, d) * 1.5).to(dt).to(cuda)
        kvt = (torch.randn(B * S, 2 * d) * 1.5).to(dt).to(cuda)
        qv, kv_, vv = (qt, d, 0), (kvt, 2 * d, 0), (kvt, 2 * d, d)
        q4 = qt.float().cpu().view(B, T, H, hd).permute(0, 2, 1, 3)
        k4 = kvt.float().cpu().view(B, S, 2, H, hd)[:, :, 0].permute(0, 2, 1, 3)
        v4 = kvt.float().cpu().view(B, S, 2, H, hd)[:, :, 1].permute(0, 2, 1, 3)
    pe = (torch.randn(2 * maxrel, hd)).to(dt).to(cuda) if rel else None
    kpm = torch.zeros(B, S, dtype=torch.uint8)
    if pad:
        kpm[1, S - 7:] = 1
    KP = kpm.to(cuda) if pad else None
    ref_o, ref_lse = _ref(q4, k4, v4, pe.float().cpu() if rel else None, kpm if pad else None, causal, maxrel)
    # unfused
    ctx_u, probs, _ = Fn._attn_fwd(qv, kv_, vv, B, H, T, S, hd, pe, maxrel if rel else 0, KP, causal, 0.0, 0)
    # flash
    o = torch.full((B * T, d), float("nan"), dtype=dt, device=cuda)
    lse = torch.empty(B * H, T, device=cuda)
    L = hip.lib()
    qpw = torch.empty(B * H, T, L.st5_flash_attn_qp_row(2 * maxrel), dtype=dt, device=cuda) if rel else None
    hip.check(L.st5_flash_attn_fwd_qp(qv[0].data_ptr() + qv[2] * 2, qv[1], kv_[0].data_ptr() + kv_[2] * 2, kv_[1],
                                      vv[0].data_ptr() + vv[2] * 2, vv[1], o.data_ptr(), d, lse.data_ptr(), hip.ptr(pe), hip.ptr(KP),
                                      B, H, T, S, hd, 2 * maxrel if rel else 0, maxrel if rel else 0, causal, (S + 7) // 8 * 8,
                                      hd ** -0.5, 0.0, 0, hip.ptr(qpw), hip.BF16, hip.stream()), "flash fwd")
    torch.cuda.synchronize()
    ref = ref_o.permute(0, 2, 1, 3).reshape(B * T, d)
    sc = ref.abs().max().item()
    assert (o.float().cpu() - ref).abs().max().item() <= 2e-2 * sc, "flash vs fp32 reference"
    assert (ctx_u.float().cpu() - ref).abs().max().item() <= 2e-2 * sc, "unfused vs fp32 reference"
    assert (o.float() - ctx_u.float()).abs().max().item() <= 2e-2 * sc, "flash vs unfused"
    assert (lse.cpu().view(B, H, T) - ref_lse).abs().max().item() <= 3e-2 * max(1.0, ref_lse.abs().max().item())


def test_flash_fwd_dropout_matches_unfused_mask(cuda):
    torch.manual_seed(9)
    B, H, T, hd = 2, 2, 150, 64
    d = H * hd
    qkv = torch.randn(B * T, 3 * d).to(torch.bfloat16).to(cuda)
    args = ((qkv, 3 * d, 0), (qkv, 3 * d, d), (qkv, 3 * d, 2 * d))
    ctx_u, _, _ = Fn._attn_fwd(*args, B, H, T, T, hd, None, 0, None, False, 0.2, 1234)
    o = torch.empty(B * T, d, dtype=torch.bfloat16, device=cuda)
    lse = torch.empty(B * H, T, device=cuda)
    hip.check(hip.lib().st5_flash_attn_fwd(qkv.data_ptr(), 3 * d, qkv.data_ptr() + d * 2, 3 * d, qkv.data_ptr() + 4 * d, 3 * d,
                                           o.data_ptr(), d, lse.data_ptr(), 0, 0, B, H, T, T, hd, 0, 0, 0, (T + 7) // 8 * 8,
                                           hd ** -0.5, 0.2, 1234, hip.BF16, hip.stream()), "flash fwd")
    sc = ctx_u.float().abs().max().item()
    assert (o.float() - ctx_u.float()).abs().max().item() <= 3e-2 * sc


@pytest.mark.parametrize("T,S,causal,rel,pad,pdrop", [(150, 150, 0, 1, 1, 0.0), (499, 499, 0, 1, 0, 0.1), (313, 313, 1, 0, 1, 0.1),
                                                       (70, 133, 0, 0, 1, 0.0), (40, 40, 1, 0, 0, 0.0)])
def test_flash_backward_matches_unfused(cuda, T, S, causal, rel, pad, pdrop):
    """Full forward+backward through the autograd functions: fused kernels vs the unfused path, identical dropout masks."""
    torch.manual_seed(T + S)
    B, H, hd = 2, 3, 64
    maxrel = 16 if T < 200 else 160
    d = H * hd
    dt = torch.bfloat16
    Fn.set_compute_dtype(dt)
    kpm = torch.zeros(B, S, dtype=torch.uint8)
    if pad:
        kpm[1, S - 9:] = 1
    KP = kpm.to(cuda) if pad else None
    pe0 = torch.randn(2 * maxrel, hd).to(dt).to(cuda) if rel else None
    dout = torch.randn(B * T, d).to(dt).to(cuda)
    res = {}
    try:
        for flash in (True, False):
            Fn.set_flash_attention(flash)
            Fn.manual_seed(77)
            pe = pe0.clone().requires_grad_(True) if rel else None
            if T == S:
                x = (torch.randn(B * T, 3 * d, generator=torch.Generator().manual_seed(1)) * 1.2).to(dt).to(cuda).requires_grad_(True)
                out = Fn.SelfAttentionFunction.apply(x, pe, KP, (B, H, T, hd, maxrel if rel else 0, bool(causal), pdrop))
                out.backward(dout)
                res[flash] = (out.detach().float(), x.grad.float(), pe.grad.float() if rel else None)
            else:
                g = torch.Generator().manual_seed(2)
                xq = (torch.randn(B * T, d, generator=g) * 1.2).to(dt).to(cuda).requires_grad_(True)
                xkv = (torch.randn(B * S, 2 * d, generator=g) * 1.2).to(dt).to(cuda).requires_grad_(True)
                out, _ = Fn.CrossAttentionFunction.apply(xq, xkv, KP, (B, H, T, S, hd, pdrop, False))
                out.backward(dout)
                res[flash] = (out.detach().float(), torch.cat([xq.grad.float().flatten(), xkv.grad.float().flatten()]), None)
    finally:
        Fn.set_flash_attention(True)
        Fn.set_compute_dtype(torch.float32)
    for i, name in enumerate(("out", "dinput", "dpe")):
        a, b = res[True][i], res[False][i]
        if a is None:
            continue
        sc = b.abs().max().item()
        err = (a - b).abs().max().item()
        assert err <= 3e-2 * sc, f"{name}: flash vs unfused err {err:.3e} scale {sc:.3e}"


@pytest.mark.parametrize("T,causal,rel,pdrop", [(499, 0, 1, 0.1), (313, 1, 0, 0.1), (150, 0, 1, 0.0)])
def test_flash_backward_two_streams_identical(cuda, T, causal, rel, pdrop):
    """The two-stream attention backward (D kernel, then dq and dkv side by side: st5_flash_attn_bwd_2s) and the saved
    bucket table of st5_flash_attn_fwd_qp give bit-identical gradients to the single-stream form."""
    torch.manual_seed(T)
    B, H, hd = 2, 3, 64
    maxrel = 160 if T > 200 else 16
    d = H * hd
    dt = torch.bfloat16
    Fn.set_compute_dtype(dt)
    pe0 = torch.randn(2 * maxrel, hd).to(dt).to(cuda) if rel else None
    dout = torch.randn(B * T, d).to(dt).to(cuda)
    side = torch.cuda.Stream()
    res = []
    try:
        for two in (False, True):
            Fn.set_attention_stream(side if two else None)
            Fn.manual_seed(5)
            pe = pe0.clone().requires_grad_(True) if rel else None
            x = (torch.randn(B * T, 3 * d, generator=torch.Generator().manual_seed(1)) * 1.2).to(dt).to(cuda).requires_grad_(True)
            out = Fn.SelfAttentionFunction.apply(x, pe, None, (B, H, T, hd, maxrel if rel else 0, bool(causal), pdrop))
            out.backward(dout)
            torch.cuda.synchronize()
            res.append((out.detach().clone(), x.grad.clone(), pe.grad.clone() if rel else None))
    finally:
        Fn.set_attention_stream(None)
        Fn.set_compute_dtype(torch.float32)
    for a, b in zip(res[0], res[1]):
        if a is not None:
            assert torch.equal(a, b)


def test_flash_fwd_saved_bucket_table(cuda):
    """qp_out of st5_flash_attn_fwd_qp == scale*log2(e)*q.pe^T (bf16, <= 1 ulp) with the clipped end buckets replicated in the
    8-element chunks on either side of every row (the layout the second-generation kernels window over)."""
    torch.manual_seed(3)
    B, H, T, hd, maxrel = 2, 2, 200, 64, 160
    d, nb = H * hd, 2 * maxrel
    qkv = torch.randn(B * T, 3 * d).to(torch.bfloat16).to(cuda)
    pe = torch.randn(nb, hd).to(torch.bfloat16).to(cuda)
    o = torch.empty(B * T, d, dtype=torch.bfloat16, device=cuda)
    lse = torch.empty(B * H, T, device=cuda)
    row = hip.lib().st5_flash_attn_qp_row(nb)
    assert row == nb + 16
    qp = torch.zeros(B * H, T, row, dtype=torch.bfloat16, device=cuda)
    hip.check(hip.lib().st5_flash_attn_fwd_qp(qkv.data_ptr(), 3 * d, qkv.data_ptr() + d * 2, 3 * d, qkv.data_ptr() + 4 * d, 3 * d,
                                              o.data_ptr(), d, lse.data_ptr(), pe.data_ptr(), 0, B, H, T, T, hd, nb, maxrel, 0,
                                              (T + 7) // 8 * 8, hd ** -0.5, 0.0, 0, qp.data_ptr(), hip.BF16, hip.stream()), "flash fwd qp")
    q = qkv.float().view(B, T, 3, H, hd)[:, :, 0].permute(0, 2, 1, 3)          # [B, H, T, hd]
    ref = (q @ pe.float().t()) * (hd ** -0.5 * 1.4426950408889634)
    full = qp.float().view(B, H, T, row)
    got = full[..., 8:8 + nb]
    assert (got - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()
    assert torch.equal(full[..., :8], got[..., :1].expand(-1, -1, -1, 8)) and torch.equal(full[..., 8 + nb:], got[..., -1:].expand(-1, -1, -1, 8))


@pytest.mark.parametrize("T,S,causal,rel,pad,pdrop", [(499, 499, 0, 1, 0, 0.1), (512, 512, 0, 1, 1, 0.1), (313, 313, 1, 0, 1, 0.1),
                                                       (313, 499, 0, 0, 1, 0.1), (150, 150, 0, 1, 1, 0.0), (70, 133, 0, 0, 1, 0.0),
                                                       (33, 33, 1, 0, 0, 0.0), (700, 700, 0, 1, 0, 0.0)])
def test_second_generation_kernels_bit_identical_to_first(cuda, T, S, causal, rel, pad, pdrop):
    """csrc/flash_attn2.hip (LDS-DMA staging, transpose reads, per-tile bias windows, half-tile backward) against
    csrc/flash_attn.hip on identical inputs and dropout seeds: identical MFMA chains and bf16 table values => outputs, LSE
    and every gradient are equal bit for bit.  Covers clipped / mixed / unclipped relative-position tiles (T = 700 > 2*maxrel +
    tile), key padding, causal masks, query and key tails, cross attention."""
    torch.manual_seed(T * 7 + S)
    B, H, hd = 2, 3, 64
    maxrel = 160 if T > 200 else 16
    d = H * hd
    dt = torch.bfloat16
    Fn.set_compute_dtype(dt)
    kpm = torch.zeros(B, S, dtype=torch.uint8)
    if pad:
        kpm[1, S - 9:] = 1
    KP = kpm.to(cuda) if pad else None
    pe0 = torch.randn(2 * maxrel, hd).to(dt).to(cuda) if rel else None
    dout = torch.randn(B * T, d).to(dt).to(cuda)
    res = {}
    L = hip.lib()
    try:
        for impl in (1, 2):
            hip.check(L.st5_flash_attn_set_impl(impl), "set_impl")
            Fn.manual_seed(77)
            pe = pe0.clone().requires_grad_(True) if rel else None
            if T == S:
                x = (torch.randn(B * T, 3 * d, generator=torch.Generator().manual_seed(1)) * 1.2).to(dt).to(cuda).requires_grad_(True)
                out = Fn.SelfAttentionFunction.apply(x, pe, KP, (B, H, T, hd, maxrel if rel else 0, bool(causal), pdrop))
                out.backward(dout)
                res[impl] = (out.detach().clone(), x.grad.clone(), pe.grad.clone() if rel else None)
            else:
                g = torch.Generator().manual_seed(2)
                xq = (torch.randn(B * T, d, generator=g) * 1.2).to(dt).to(cuda).requires_grad_(True)
                xkv = (torch.randn(B * S, 2 * d, generator=g) * 1.2).to(dt).to(cuda).requires_grad_(True)
                out, _ = Fn.CrossAttentionFunction.apply(xq, xkv, KP, (B, H, T, S, hd, pdrop, False))
                out.backward(dout)
                res[impl] = (out.detach().clone(), xq.grad.clone(), xkv.grad.clone())
            torch.cuda.synchronize()
    finally:
        hip.check(L.st5_flash_attn_set_impl(2), "set_impl")
        Fn.set_compute_dtype(torch.float32)
    for i, name in enumerate(("out", "grad0", "grad1")):
        a, b = res[1][i], res[2][i]
        if a is None:
            continue
        assert torch.isfinite(b.float()).all(), name
        if not torch.equal(a, b):
            diff = (a.float() - b.float()).abs()
            raise AssertionError(f"{name}: {int((diff > 0).sum())} of {diff.numel()} elements differ, max {diff.max().item():.3e} "
                                 f"(scale {a.float().abs().max().item():.3e})")

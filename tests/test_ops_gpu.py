"""Kernel-level parity of the C-ABI library (through ctypes) against plain PyTorch fp32 CPU math.

fp32 mode must agree to fp32 round-off (it backs the 1e-3 mel / bit-exact token-id parity claims);
bf16 mode is compared with the same math evaluated on bf16-rounded inputs, tolerance 2e-2 of the
output scale (bf16 has 8 mantissa bits; accumulation is fp32)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from speecht5_amd import hip  # noqa: E402
from speecht5_amd import functional as Fn  # noqa: E402

DTYPES = [torch.float32, torch.bfloat16]


def tol(dtype):
    return (2e-5, 2e-5) if dtype == torch.float32 else (2e-2, 2e-2)


def rt(x, dtype):  # round-trip through dtype (what the kernel sees); always a fresh tensor
    return x.detach().to(dtype).float().clone()


def close(got, ref, dtype, scale=None, what=""):
    rtol, atol = tol(dtype)
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    s = ref.abs().max().item() if scale is None else scale
    err = (got - ref).abs().max().item()
    assert err <= atol * max(s, 1e-6) + 1e-7, f"{what}: max err {err:.3e} vs scale {s:.3e} ({dtype})"


def dev(x, dtype, cuda):
    return x.to(dtype).to(cuda).contiguous()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(300, 200, 136), (128, 128, 64), (1, 83, 768), (257, 520, 776), (64, 48, 6144)])
def test_gemm_nt_epilogues(cuda, dtype, M, N, K):
    torch.manual_seed(M * 7 + N)
    a, w = torch.randn(M, K), torch.randn(N, K) / math.sqrt(K)
    bias, res = torch.randn(N), torch.randn(M, N)
    ldc = (N + 7) // 8 * 8
    A, W = dev(a, dtype, cuda), dev(w, dtype, cuda)
    Rr = torch.zeros(M, ldc); Rr[:, :N] = res
    Rd = dev(Rr, dtype, cuda)
    C = torch.full((M, ldc), float("nan"), dtype=dtype, device=cuda)
    Cp = torch.full((M, ldc), float("nan"), dtype=dtype, device=cuda)
    hip.gemm(hip.operand(A, K), hip.operand(W, K), hip.operand(C, ldc), M, N, K, hip.dt(dtype),
             R=hip.operand(Rd, ldc), Cpre=hip.operand(Cp, ldc), bias=bias.to(cuda), act=hip.ACT_GELU, alpha=0.5)
    pre = 0.5 * rt(a, dtype) @ rt(w, dtype).t() + bias
    ref = F.gelu(pre) + rt(res, dtype)
    close(C[:, :N], ref, dtype, what="gemm nt")
    close(Cp[:, :N], pre, dtype, what="gemm pre-activation")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("aks,bks", [(0, 1), (1, 0), (1, 1)])
def test_gemm_transposed_layouts(cuda, dtype, aks, bks):
    torch.manual_seed(3)
    M, N, K = 200, 136, 300
    a, b = torch.randn(M, K), torch.randn(N, K) / math.sqrt(K)
    ref = rt(a, dtype) @ rt(b, dtype).t()
    lda_t, ldb_t = (M + 7) // 8 * 8, (N + 7) // 8 * 8
    ldk = (K + 7) // 8 * 8  # K-major rows are 16-byte aligned; the K tail (300 % 8 != 0) is masked in-kernel
    if aks:
        At = torch.zeros(K, lda_t); At[:, :M] = a.t(); A = dev(At, dtype, cuda); opA = hip.operand(A, lda_t)
    else:
        Ap = torch.full((M, ldk), float("nan")); Ap[:, :K] = a; A = dev(Ap, dtype, cuda); opA = hip.operand(A, ldk)
    if bks:
        Bt = torch.zeros(K, ldb_t); Bt[:, :N] = b.t(); B = dev(Bt, dtype, cuda); opB = hip.operand(B, ldb_t)
    else:
        Bp = torch.full((N, ldk), float("nan")); Bp[:, :K] = b; B = dev(Bp, dtype, cuda); opB = hip.operand(B, ldk)
    # fp32 output with accumulate (the weight-gradient form)
    C = torch.ones(M, N, dtype=torch.float32, device=cuda)
    hip.gemm(opA, opB, hip.operand(C, N), M, N, K, hip.dt(dtype),
             flags=(hip.A_KSTRIDED if aks else 0) | (hip.B_KSTRIDED if bks else 0) | hip.OUT_F32, beta=1.0)
    close(C, ref + 1.0, dtype, what=f"gemm aks={aks} bks={bks}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(768, 256, 5000), (136, 72, 8192), (264, 776, 130)])
def test_gemm_weight_gradient_form(cuda, dtype, M, N, K):
    """dW[M, N] (+)= dY^T X with both operands k-strided, fp32 accumulate output: the split-K path and, for bf16 with
    8-aligned M / N, the LDS-DMA + transpose-read kernel (K tails, M/N tile tails)."""
    torch.manual_seed(M + K)
    dy, x = torch.randn(K, M), torch.randn(K, N) / math.sqrt(K)
    ref = rt(dy, dtype).t() @ rt(x, dtype)
    DY, X = dev(dy, dtype, cuda), dev(x, dtype, cuda)
    C = torch.full((M, N), 2.0, dtype=torch.float32, device=cuda)
    hip.gemm(hip.operand(DY, M), hip.operand(X, N), hip.operand(C, N), M, N, K, hip.dt(dtype),
             flags=hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32, beta=1.0)
    close(C, ref + 2.0, dtype, what="wgrad form")
    # bias gradient riding on the same kernel: asum[m] += sum_k dY[k, m]
    C3 = torch.zeros(M, N, dtype=torch.float32, device=cuda)
    db = torch.full((M,), 3.0, device=cuda)
    hip.gemm(hip.operand(DY, M), hip.operand(X, N), hip.operand(C3, N), M, N, K, hip.dt(dtype),
             flags=hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32, asum=db)
    close(C3, ref, dtype, what="wgrad form with asum")
    close(db - 3.0, rt(dy, dtype).sum(0), dtype, scale=math.sqrt(K), what="bias gradient (asum)")
    # same operands as a batched product out of a wider buffer (leading dimensions larger than the matrix)
    C2 = torch.zeros(M, N, dtype=dtype, device=cuda)
    wide = torch.zeros(K, M + 8, dtype=dtype, device=cuda); wide[:, :M] = DY
    hip.gemm(hip.operand(wide, M + 8), hip.operand(X, N), hip.operand(C2, N), M, N, K, hip.dt(dtype),
             flags=hip.A_KSTRIDED | hip.B_KSTRIDED, bias=torch.ones(N, device=cuda))
    close(C2, ref + 1.0, dtype, what="wgrad form, bf16 output + bias")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_batched_heads(cuda, dtype):
    """q.k^T per (batch, head) straight out of a fused [B*T, 3d] projection buffer."""
    torch.manual_seed(5)
    B, H, T, hd = 2, 3, 70, 64
    d = H * hd
    qkv = torch.randn(B * T, 3 * d)
    Q = dev(qkv, dtype, cuda)
    lds = (T + 7) // 8 * 8
    S = torch.zeros(B * H, T, lds, dtype=dtype, device=cuda)
    hip.gemm(hip.operand(Q, 3 * d, zs0=T * 3 * d, zs1=hd), hip.operand(Q, 3 * d, off=d, zs0=T * 3 * d, zs1=hd),
             hip.operand(S, lds, zs0=H * T * lds, zs1=T * lds), T, T, hd, hip.dt(dtype), batch=B * H, zdiv=H, alpha=0.125)
    x = rt(qkv, dtype).view(B, T, 3, H, hd)
    ref = 0.125 * torch.einsum("bihd,bjhd->bhij", x[:, :, 0], x[:, :, 1]).reshape(B * H, T, T)
    close(S[:, :, :T], ref, dtype, what="batched qk")
    # P.V with V k-strided, output back into [B*T, d]
    P = torch.softmax(ref, -1)
    Pd = torch.zeros(B * H, T, lds); Pd[:, :, :T] = P
    Pd = dev(Pd, dtype, cuda)
    O = torch.zeros(B * T, d, dtype=dtype, device=cuda)
    hip.gemm(hip.operand(Pd, lds, zs0=H * T * lds, zs1=T * lds),
             hip.operand(Q, 3 * d, off=2 * d, zs0=T * 3 * d, zs1=hd),
             hip.operand(O, d, zs0=T * d, zs1=hd), T, hd, T, hip.dt(dtype), batch=B * H, zdiv=H, flags=hip.B_KSTRIDED)
    refo = torch.einsum("bhij,bjhd->bihd", rt(P, dtype).view(B, H, T, T), x[:, :, 2]).reshape(B * T, d)
    close(O, refo, dtype, what="batched pv")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k,stride", [(3, 2), (2, 2)])
def test_gemm_conv1d_channels_last(cuda, dtype, k, stride):
    """Strided Conv1d as implicit GEMM on channels-last activations (overlapping rows) + GELU."""
    torch.manual_seed(11)
    B, L, Cin, Cout = 3, 101, 64, 72
    Lo = (L - k) // stride + 1
    x = torch.randn(B, L, Cin)
    w = torch.randn(Cout, Cin, k) / math.sqrt(Cin * k)
    X = dev(x, dtype, cuda)
    Wk = dev(w.permute(0, 2, 1).reshape(Cout, k * Cin), dtype, cuda)  # [Cout, k, Cin]
    Y = torch.zeros(B, Lo, Cout, dtype=dtype, device=cuda)
    hip.gemm(hip.operand(X, stride * Cin, rpb=Lo, bstride=L * Cin), hip.operand(Wk, k * Cin), hip.operand(Y, Cout),
             B * Lo, Cout, k * Cin, hip.dt(dtype), act=hip.ACT_GELU)
    ref = F.gelu(F.conv1d(rt(x, dtype).transpose(1, 2), rt(w, dtype), stride=stride)).transpose(1, 2)
    close(Y, ref, dtype, what="conv1d implicit gemm")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_grouped_posconv(cuda, dtype):
    """Grouped Conv1d(k=16, groups=4, pad=8) via K segmentation over a zero-padded channels-last copy."""
    torch.manual_seed(13)
    B, T, C, G, k = 2, 50, 64, 4, 16
    cg = C // G
    x = torch.randn(B, T, C)
    w = torch.randn(C, cg, k) / math.sqrt(cg * k)
    bias = torch.randn(C)
    X = dev(x, dtype, cuda)
    Tp = T + k
    Xp = torch.empty(B, Tp, C, dtype=dtype, device=cuda)
    hip.check(hip.lib().st5_pad_time(X.data_ptr(), Xp.data_ptr(), B, T, C, k // 2, k // 2, hip.dt(dtype), hip.stream()), "pad")
    # weights -> [G][cg_out][k][cg_in]
    Wg = dev(w.view(G, cg, cg, k).permute(0, 1, 3, 2).reshape(G, cg, k * cg), dtype, cuda)
    Y = torch.zeros(B, T, C, dtype=dtype, device=cuda)
    hip.gemm(hip.operand(Xp, C, rpb=T, bstride=Tp * C, seg=cg, seg_stride=C, zs0=cg),
             hip.operand(Wg, k * cg, zs0=cg * k * cg), hip.operand(Y, C, zs0=cg), B * T, cg, k * cg, hip.dt(dtype),
             batch=G, bias=bias.to(cuda), bias_zs=cg, act=hip.ACT_GELU)
    ref = F.conv1d(rt(x, dtype).transpose(1, 2), rt(w, dtype), bias, padding=k // 2, groups=G)[:, :, :T]
    ref = F.gelu(ref).transpose(1, 2)
    close(Y, ref, dtype, what="grouped pos conv")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_dact_and_dropout(cuda, dtype):
    torch.manual_seed(17)
    M, N, K = 150, 96, 64
    a, w, pre = torch.randn(M, K), torch.randn(N, K) / 8, torch.randn(M, N)
    A, W, P = dev(a, dtype, cuda), dev(w, dtype, cuda), dev(pre, dtype, cuda)
    C = torch.zeros(M, N, dtype=dtype, device=cuda)
    hip.gemm(hip.operand(A, K), hip.operand(W, K), hip.operand(C, N), M, N, K, hip.dt(dtype), P=hip.operand(P, N),
             act=hip.ACT_GELU, flags=hip.DACT)
    x = rt(pre, dtype).requires_grad_(True)
    F.gelu(x).sum().backward()
    close(C, (rt(a, dtype) @ rt(w, dtype).t()) * x.grad, dtype, what="dact")
    # dropout: deterministic in (seed, index), ~p zeros, survivors scaled
    C1 = torch.zeros(M, N, dtype=dtype, device=cuda); C2 = torch.zeros_like(C1); C3 = torch.zeros_like(C1)
    for out, seed in ((C1, 7), (C2, 7), (C3, 8)):
        hip.gemm(hip.operand(A, K), hip.operand(W, K), hip.operand(out, N), M, N, K, hip.dt(dtype), dropout_p=0.25, seed=seed)
    assert torch.equal(C1, C2) and not torch.equal(C1, C3)
    full = rt(a, dtype) @ rt(w, dtype).t()
    keep = (C1 != 0).float().cpu()
    assert abs(keep.mean().item() - 0.75) < 0.03
    close(C1.float().cpu(), full * keep / 0.75, dtype, what="dropout survivors")
    # the standalone dropout kernel uses the same generator/counter as the epilogue
    Y = torch.zeros(M, N, dtype=dtype, device=cuda)
    Fd = dev(full, dtype, cuda)
    hip.check(hip.lib().st5_dropout(Fd.data_ptr(), Y.data_ptr(), M * N, 0.25, 7, hip.dt(dtype), hip.stream()), "dropout")
    assert torch.equal((Y != 0), (C1 != 0)) or ((Y != 0) ^ (C1 != 0)).float().mean().item() < 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(2100, 1024), (333, 768), (9, 64)])
def test_layernorm_backward_with_residual_addend(cuda, dtype, rows, cols):
    """Round 6 (st5_layernorm_bwd_add: the residual gradient of a pre-LN block folded into the LayerNorm backward): dx == the plain
    backward's dx + addend up to ONE rounding of the sum (the fused form adds in fp32 before it rounds), dgamma / dbeta bit-identical."""
    torch.manual_seed(rows)
    X, DY, AD = dev(torch.randn(rows, cols) * 2 + 0.5, dtype, cuda), dev(torch.randn(rows, cols), dtype, cuda), dev(torch.randn(rows, cols), dtype, cuda)
    G = torch.randn(cols).to(cuda)
    L = hip.lib()
    mean = X.float().mean(1).contiguous(); rstd = (X.float().var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    outs = []
    for fused in (False, True):
        DX = torch.empty_like(X)
        dG = torch.ones(cols, device=cuda); dB = torch.ones(cols, device=cuda)
        ws = hip.workspace(L.st5_layernorm_bwd_ws_bytes(rows, cols), cuda)
        if fused:
            hip.check(L.st5_layernorm_bwd_add(DY.data_ptr(), X.data_ptr(), G.data_ptr(), mean.data_ptr(), rstd.data_ptr(), DX.data_ptr(), dG.data_ptr(),
                                              dB.data_ptr(), ws.data_ptr(), rows, cols, AD.data_ptr(), hip.dt(dtype), hip.stream()), "bwd add")
        else:
            hip.check(L.st5_layernorm_bwd(DY.data_ptr(), X.data_ptr(), G.data_ptr(), mean.data_ptr(), rstd.data_ptr(), DX.data_ptr(), dG.data_ptr(),
                                          dB.data_ptr(), ws.data_ptr(), rows, cols, None, 0.0, 0, hip.dt(dtype), hip.stream()), "bwd")
        torch.cuda.synchronize()
        outs.append((DX, dG, dB))
    want = outs[0][0].float() + AD.float()
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    assert float((outs[1][0].float() - want).abs().max()) <= tol * float(want.abs().max())
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(3001, 512), (70, 256), (5, 64)])
def test_layernorm_gelu_one_pass_against_torch(cuda, dtype, rows, cols):
    """Round 6 (st5_layernorm_gelu_fwd / _bwd: the layer-norm convolution extractor of t5_transformer_large): GELU(LayerNorm(x)) and its
    backward -- dx, dgamma, dbeta ACCUMULATED into non-zero buffers -- against torch fp64 autograd on the same (rounded) inputs."""
    torch.manual_seed(rows + cols)
    X = dev(torch.randn(rows, cols) * 1.7 + 0.3, dtype, cuda)
    DY = dev(torch.randn(rows, cols), dtype, cuda)
    G = (torch.randn(cols) * 0.4 + 1.0).to(cuda); Bt = (torch.randn(cols) * 0.3).to(cuda)
    L = hip.lib()
    Y = torch.empty_like(X); DX = torch.empty_like(X)
    mean = torch.empty(rows, device=cuda); rstd = torch.empty(rows, device=cuda)
    dG = torch.full((cols,), 0.5, device=cuda); dB = torch.full((cols,), -0.25, device=cuda)
    ws = hip.workspace(L.st5_layernorm_bwd_ws_bytes(rows, cols), cuda)
    hip.check(L.st5_layernorm_gelu_fwd(X.data_ptr(), G.data_ptr(), Bt.data_ptr(), Y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, cols,
                                       1e-5, hip.dt(dtype), hip.stream()), "st5_layernorm_gelu_fwd")
    hip.check(L.st5_layernorm_gelu_bwd(DY.data_ptr(), X.data_ptr(), G.data_ptr(), Bt.data_ptr(), mean.data_ptr(), rstd.data_ptr(), DX.data_ptr(),
                                       dG.data_ptr(), dB.data_ptr(), ws.data_ptr(), rows, cols, hip.dt(dtype), hip.stream()), "st5_layernorm_gelu_bwd")
    torch.cuda.synchronize()
    x = X.double().requires_grad_(True); g = G.double().requires_grad_(True); b = Bt.double().requires_grad_(True)
    y = torch.nn.functional.gelu(torch.nn.functional.layer_norm(x, (cols,), g, b, 1e-5))
    y.backward(DY.double())
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2
    def close(a, r, what, t=tol):
        err = float((a.double() - r).abs().max()) / max(float(r.abs().max()), 1e-6)
        assert err <= t, (what, err)
    close(Y, y.detach(), "y")
    close(DX, x.grad, "dx")
    close(dG - 0.5, g.grad, "dgamma", tol * 4)
    close(dB + 0.25, b.grad, "dbeta", tol * 4)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("keep", [1.0, 0.0])
def test_layernorm_gated_by_a_layerdrop_flag(cuda, dtype, keep):
    """st5_layernorm_gated_fwd / _bwd + st5_skip_grad (the LayerDrop select folded into a post-LN layer's last LayerNorm,
    functional.LayerDropGate): keep = 1 is bit for bit the plain LayerNorm and leaves the input gradient alone; keep = 0 hands the
    skip rows through, produces exact zeros in dx / dgamma / dbeta and replaces the input gradient by the output gradient."""
    rows, cols = 2100, 768
    torch.manual_seed(5)
    X, DY, SK = dev(torch.randn(rows, cols) * 2 + 0.5, dtype, cuda), dev(torch.randn(rows, cols), dtype, cuda), dev(torch.randn(rows, cols), dtype, cuda)
    G, Bt = torch.randn(cols).to(cuda), torch.randn(cols).to(cuda)
    K = torch.tensor([keep], device=cuda)
    L = hip.lib()

    def run(gated):
        Y = torch.empty_like(X); DX = torch.empty_like(X)
        mean = torch.empty(rows, device=cuda); rstd = torch.empty(rows, device=cuda)
        dG = torch.ones(cols, device=cuda); dB = torch.ones(cols, device=cuda)
        ws = hip.workspace(L.st5_layernorm_bwd_ws_bytes(rows, cols), cuda)
        if gated:
            hip.check(L.st5_layernorm_gated_fwd(X.data_ptr(), G.data_ptr(), Bt.data_ptr(), Y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                rows, cols, 1e-5, K.data_ptr(), SK.data_ptr(), hip.dt(dtype), hip.stream()), "gated fwd")
            hip.check(L.st5_layernorm_gated_bwd(DY.data_ptr(), X.data_ptr(), G.data_ptr(), mean.data_ptr(), rstd.data_ptr(), DX.data_ptr(),
                                                dG.data_ptr(), dB.data_ptr(), ws.data_ptr(), rows, cols, None, 0.0, 0, K.data_ptr(),
                                                hip.dt(dtype), hip.stream()), "gated bwd")
        else:
            hip.check(L.st5_layernorm_fwd(X.data_ptr(), G.data_ptr(), Bt.data_ptr(), Y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                          rows, cols, 1e-5, hip.dt(dtype), hip.stream()), "fwd")
            hip.check(L.st5_layernorm_bwd(DY.data_ptr(), X.data_ptr(), G.data_ptr(), mean.data_ptr(), rstd.data_ptr(), DX.data_ptr(),
                                          dG.data_ptr(), dB.data_ptr(), ws.data_ptr(), rows, cols, None, 0.0, 0, hip.dt(dtype), hip.stream()), "bwd")
        torch.cuda.synchronize()
        return Y, DX, dG, dB

    plain, gated = run(False), run(True)
    din = dev(torch.randn(rows, cols), dtype, cuda)      # the layer's own input gradient
    d0 = din.clone()
    hip.check(L.st5_skip_grad(K.data_ptr(), DY.data_ptr(), din.data_ptr(), din.numel() * din.element_size(), hip.stream()), "skip grad")
    torch.cuda.synchronize()
    if keep:
        for a, b in zip(plain, gated):
            assert torch.equal(a, b)
        assert torch.equal(din, d0)
    else:
        assert torch.equal(gated[0], SK)
        assert float(gated[1].abs().max()) == 0.0
        assert torch.equal(gated[2], torch.ones_like(gated[2])) and torch.equal(gated[3], torch.ones_like(gated[3]))
        assert torch.equal(din, DY)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,cols", [(37, 768), (130, 512), (9, 64), (5, 1024), (2100, 768), (11, 30), (6, 1536)])
def test_layernorm(cuda, dtype, rows, cols):
    torch.manual_seed(rows)
    x = torch.randn(rows, cols) * 2 + 0.5
    g, b, dy = torch.randn(cols), torch.randn(cols), torch.randn(rows, cols)
    X, DY = dev(x, dtype, cuda), dev(dy, dtype, cuda)
    G, Bt = g.to(cuda), b.to(cuda)
    Y = torch.empty_like(X)
    mean = torch.empty(rows, device=cuda); rstd = torch.empty(rows, device=cuda)
    L = hip.lib()
    hip.check(L.st5_layernorm_fwd(X.data_ptr(), G.data_ptr(), Bt.data_ptr(), Y.data_ptr(), mean.data_ptr(),
                                  rstd.data_ptr(), rows, cols, 1e-5, hip.dt(dtype), hip.stream()), "ln fwd")
    xr = rt(x, dtype).requires_grad_(True); gr = g.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (cols,), gr, br, 1e-5)
    close(Y, ref, dtype, what="ln fwd")
    ref.backward(rt(dy, dtype))
    DX = torch.empty_like(X)
    dG = torch.ones(cols, device=cuda); dB = torch.ones(cols, device=cuda)  # accumulate semantics
    ws = hip.workspace(L.st5_layernorm_bwd_ws_bytes(rows, cols), cuda)
    hip.check(L.st5_layernorm_bwd(DY.data_ptr(), X.data_ptr(), G.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                  DX.data_ptr(), dG.data_ptr(), dB.data_ptr(), ws.data_ptr(), rows, cols,
                                  None, 0.0, 0, hip.dt(dtype), hip.stream()), "ln bwd")
    close(DX, xr.grad, dtype, what="ln dx")
    close(dG - 1, gr.grad, dtype, what="ln dgamma")
    close(dB - 1, br.grad, dtype, what="ln dbeta")
    if cols % 4 == 0:
        # second output: dX under the dropout mask of the Linear in front (same counter stream as st5_dropout / the
        # GEMM epilogue: element index row * cols + col) -- bit-identical to masking dX afterwards
        DX2 = torch.empty_like(X); DXD = torch.empty_like(X)
        hip.check(L.st5_layernorm_bwd(DY.data_ptr(), X.data_ptr(), G.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                      DX2.data_ptr(), dG.data_ptr(), dB.data_ptr(), ws.data_ptr(), rows, cols,
                                      DXD.data_ptr(), 0.25, 991, hip.dt(dtype), hip.stream()), "ln bwd + dropped")
        assert torch.equal(DX2, DX)
        want = torch.empty_like(X)
        hip.check(L.st5_dropout(DX.data_ptr(), want.data_ptr(), rows * cols, 0.25, 991, hip.dt(dtype), hip.stream()), "dropout")
        assert torch.equal(DXD, want)
    else:
        assert L.st5_layernorm_bwd(DY.data_ptr(), X.data_ptr(), G.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                   DX.data_ptr(), dG.data_ptr(), dB.data_ptr(), ws.data_ptr(), rows, cols,
                                   DX.data_ptr(), 0.25, 991, hip.dt(dtype), hip.stream()) != 0


def _ref_attn_probs(scores, qp, kpm, H, causal, maxrel):
    """multihead_attention.py:343-385 with the bias gathered from q.pe^T."""
    BH, T, S = scores.shape
    s = scores.clone()
    if qp is not None:
        i = torch.arange(T)[:, None]; j = torch.arange(S)[None, :]
        idx = (i - j).clamp(-maxrel, maxrel - 1) + maxrel
        s = s + torch.gather(qp, 2, idx.expand(BH, T, S))
    if causal:
        s = s + torch.triu(torch.full((T, S), float("-inf")), 1 + (S - T))
    if kpm is not None:
        s = s.view(-1, H, T, S).masked_fill(kpm[:, None, None, :].bool(), float("-inf")).view(BH, T, S)
    return torch.softmax(s, -1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,S,causal,rel,pad", [(70, 70, 0, 1, 1), (33, 33, 1, 0, 1), (20, 75, 0, 0, 1), (200, 200, 0, 1, 0)])
def test_softmax_relpos(cuda, dtype, T, S, causal, rel, pad):
    torch.manual_seed(T + S)
    B, H, maxrel = 2, 3, 16
    BH, nb = B * H, 2 * maxrel
    lds = (S + 7) // 8 * 8
    scores = torch.randn(BH, T, S) * 2
    qp = torch.randn(BH, T, nb) if rel else None
    kpm = torch.zeros(B, S, dtype=torch.uint8)
    if pad:
        kpm[1, S - 5:] = 1
    sc = torch.zeros(BH, T, lds); sc[:, :, :S] = scores
    SC = dev(sc, dtype, cuda)
    QP = dev(qp, dtype, cuda) if rel else None
    KP = kpm.to(cuda) if pad else None
    P = torch.full((BH, T, lds), float("nan"), dtype=dtype, device=cuda)
    L = hip.lib()
    hip.check(L.st5_softmax_fwd(SC.data_ptr(), hip.ptr(QP), hip.ptr(KP), P.data_ptr(), 0, BH, H, T, S, lds, nb, maxrel,
                                causal, 0.0, 0, hip.dt(dtype), hip.stream()), "softmax fwd")
    sr = rt(scores, dtype).requires_grad_(True)
    qr = rt(qp, dtype).requires_grad_(True) if rel else None
    ref = _ref_attn_probs(sr, qr, kpm if pad else None, H, causal, maxrel)
    close(P[:, :, :S], ref, dtype, what="softmax fwd")
    assert float(P[:, :, S:].abs().sum()) == 0.0
    # backward: dS and the bucket-scattered dQP
    dP = torch.randn(BH, T, S)
    extra = torch.randn(BH, T, S) * 0.1
    ref.backward(rt(dP, dtype) + extra)
    dp = torch.zeros(BH, T, lds); dp[:, :, :S] = dP
    DP = dev(dp, dtype, cuda)
    DQP = torch.full((BH, T, nb), float("nan"), dtype=dtype, device=cuda) if rel else None
    Pin = dev(torch.cat([ref.detach(), torch.zeros(BH, T, lds - S)], -1), dtype, cuda)
    EX = extra.to(cuda)
    hip.check(L.st5_softmax_bwd(DP.data_ptr(), Pin.data_ptr(), EX.data_ptr(), hip.ptr(DQP), BH, T, S, lds,
                                nb, maxrel, 0.0, 0, hip.dt(dtype), hip.stream()), "softmax bwd")
    close(DP[:, :, :S], sr.grad, dtype, scale=1.0, what="softmax dS")
    if rel:
        close(DQP, qr.grad, dtype, scale=max(1.0, qr.grad.abs().max().item()), what="softmax dQP")


@pytest.mark.parametrize("dtype", DTYPES)
def test_softmax_dropout_consistency(cuda, dtype):
    torch.manual_seed(2)
    BH, H, T, S = 4, 2, 40, 40
    SC = dev(torch.randn(BH, T, S), dtype, cuda)
    P = torch.empty_like(SC); PD = torch.empty_like(SC)
    L = hip.lib()
    hip.check(L.st5_softmax_fwd(SC.data_ptr(), 0, 0, P.data_ptr(), PD.data_ptr(), BH, H, T, S, S, 0, 0, 0, 0.3, 99,
                                hip.dt(dtype), hip.stream()), "softmax fwd drop")
    keep = (PD != 0)
    assert abs(keep.float().mean().item() - 0.7) < 0.03
    close(PD.float().cpu(), P.float().cpu() * keep.float().cpu() / 0.7, dtype, what="dropped probs")
    # backward regenerates the same mask: dS must equal P*(m*dP/keep - sum(...))
    dP = torch.randn(BH, T, S)
    DP = dev(dP, dtype, cuda)
    hip.check(L.st5_softmax_bwd(DP.data_ptr(), P.data_ptr(), 0, 0, BH, T, S, S, 0, 0, 0.3, 99, hip.dt(dtype),
                                hip.stream()), "softmax bwd drop")
    p = P.float().cpu(); g = rt(dP, dtype) * keep.float().cpu() / 0.7
    ref = p * (g - (g * p).sum(-1, keepdim=True))
    close(DP, ref, dtype, scale=1.0, what="softmax bwd with dropout")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,C", [(2, 3000, 64), (1, 16000, 512), (1, 4000, 1024)])
def test_conv0_groupnorm_gelu(cuda, dtype, B, S, C):
    torch.manual_seed(S)
    k, stride = 10, 5
    wav = torch.randn(B, S)
    w = torch.randn(C, 1, k) * math.sqrt(2.0 / k)
    g, b = torch.rand(C) + 0.5, torch.randn(C) * 0.1
    Lo = (S - k) // stride + 1
    Ld = hip.lib()
    WAV, W, G, Bt = wav.to(cuda), w.view(C, k).contiguous().to(cuda), g.to(cuda), b.to(cuda)
    out = torch.empty(B, Lo, C, dtype=dtype, device=cuda)
    stats = torch.empty(B, C, 2, device=cuda)
    ws = hip.workspace(Ld.st5_conv0_ws_bytes(B, S, C, k, stride), cuda)
    hip.check(Ld.st5_conv0_gn_gelu_fwd(WAV.data_ptr(), W.data_ptr(), G.data_ptr(), Bt.data_ptr(), out.data_ptr(),
                                       stats.data_ptr(), ws.data_ptr(), B, S, C, k, stride, 1e-5, hip.dt(dtype),
                                       hip.stream()), "conv0 fwd")
    wr = w.clone().requires_grad_(True); gr = g.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    y = F.gelu(F.group_norm(F.conv1d(wav[:, None], wr, stride=stride), C, gr, br, 1e-5))  # [B,C,L]
    close(out, y.transpose(1, 2), dtype, what="conv0 fwd")
    dy = torch.randn(B, Lo, C)
    y.backward(rt(dy, dtype).transpose(1, 2) * 0.1)  # gscale = 0.1 (feature_grad_mult)
    DY = dev(dy, dtype, cuda)
    dW = torch.zeros(C, k, device=cuda); dG = torch.zeros(C, device=cuda); dB = torch.zeros(C, device=cuda)
    hip.check(Ld.st5_conv0_gn_gelu_bwd(WAV.data_ptr(), W.data_ptr(), G.data_ptr(), Bt.data_ptr(), stats.data_ptr(),
                                       DY.data_ptr(), dW.data_ptr(), dG.data_ptr(), dB.data_ptr(), ws.data_ptr(), B, S,
                                       C, k, stride, 0.1, hip.dt(dtype), hip.stream()), "conv0 bwd")
    # bf16: the kernel recomputes x_hat in fp32 while dY is bf16-rounded -> same tolerance class
    close(dW, wr.grad.view(C, k), dtype, what="conv0 dW")
    close(dG, gr.grad, dtype, what="conv0 dgamma")
    close(dB, br.grad, dtype, what="conv0 dbeta")


@pytest.mark.parametrize("B,S,C", [(2, 3007, 64), (1, 16000, 512), (3, 5125, 256)])
def test_conv0_matrix_core_apply_equals_the_valu_apply(cuda, B, S, C):
    """bf16 forward of conv layer 0: the matrix-core form (split-bf16 operands, two MFMAs per 32 channels x 32 steps, permuted channel
    rows so that a lane stores 16 consecutive channels) against the VALU form (fp32 FMAs): the pre-activations agree to ~2^-16, so the
    bf16 outputs are equal except where that flips a rounding -- never by more than one bf16 ulp; ragged last time tile included."""
    torch.manual_seed(S + C)
    k, stride = 10, 5
    Lo = (S - k) // stride + 1
    wav = torch.randn(B, S, device=cuda)
    w = (torch.randn(C, k) * math.sqrt(2.0 / k)).to(cuda)
    g, b = (torch.rand(C) + 0.5).to(cuda), (torch.randn(C) * 0.1).to(cuda)
    Ld = hip.lib()
    ws = hip.workspace(Ld.st5_conv0_ws_bytes(B, S, C, k, stride), cuda)
    outs = []
    try:
        for mode in (0, 1):
            hip.check(Ld.st5_conv0_set_mfma(mode), "set_mfma")
            out = torch.full((B, Lo, C), float("nan"), dtype=torch.bfloat16, device=cuda)
            stats = torch.empty(B, C, 2, device=cuda)
            hip.check(Ld.st5_conv0_gn_gelu_fwd(wav.data_ptr(), w.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(), stats.data_ptr(),
                                               ws.data_ptr(), B, S, C, k, stride, 1e-5, hip.BF16, hip.stream()), "conv0 fwd")
            torch.cuda.synchronize()
            outs.append(out.float())
    finally:
        hip.check(Ld.st5_conv0_set_mfma(1), "set_mfma")
    valu, mfma = outs
    assert torch.isfinite(mfma).all()
    diff = (valu - mfma).abs()
    # one bf16 ulp of the larger value, plus the split-operand error of the pre-activation itself (2^-16 sum |w_j x_j| ~ 5e-5) where
    # the output is tiny
    ulp = torch.maximum(valu.abs(), mfma.abs()) * 2.0 ** -7 + 6e-5
    assert bool((diff <= ulp).all()), float((diff / ulp).max())
    assert float((diff > 0).float().mean()) < 0.03, float((diff > 0).float().mean())


@pytest.mark.parametrize("B,S,C,k", [(2, 3007, 64, 10), (8, 160000, 512, 10), (3, 5125, 256, 7), (1, 4000, 1024, 10), (2, 300, 512, 10)])
def test_conv0_folded_statistics_equal_the_separate_launches_bit_for_bit(cuda, B, S, C, k):
    """The matrix-core forward with the statistics and weight fragments from ONE launch (moments, statistics + fragments, apply) against
    the four-launch form (moments, statistics, fragments, apply): the same device functions in the same order, so the output AND the
    saved statistics are the same bits -- at the benched shape, ragged tiles, k < 10, one time chunk, C = 1024."""
    torch.manual_seed(S + C)
    stride = 5
    Lo = (S - k) // stride + 1
    wav = torch.randn(B, S, device=cuda) * 0.7 + 0.05
    w = (torch.randn(C, k) * math.sqrt(2.0 / k)).to(cuda)
    g, b = (torch.rand(C) + 0.5).to(cuda), (torch.randn(C) * 0.1).to(cuda)
    dY = (torch.randn(B, Lo, C, device=cuda) * 0.3).to(torch.bfloat16)
    Ld = hip.lib()
    ws = hip.workspace(Ld.st5_conv0_ws_bytes(B, S, C, k, stride), cuda)
    res = []
    try:
        for mode in (0, 1):
            hip.check(Ld.st5_conv0_set_fold(mode), "set_fold")
            out = torch.full((B, Lo, C), float("nan"), dtype=torch.bfloat16, device=cuda)
            stats = torch.full((B, C, 2), float("nan"), device=cuda)
            hip.check(Ld.st5_conv0_gn_gelu_fwd(wav.data_ptr(), w.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(), stats.data_ptr(),
                                               ws.data_ptr(), B, S, C, k, stride, 1e-5, hip.BF16, hip.stream()), "conv0 fwd")
            dW = torch.zeros(C, k, device=cuda); dG = torch.zeros(C, device=cuda); dB = torch.zeros(C, device=cuda)
            hip.check(Ld.st5_conv0_gn_gelu_bwd(wav.data_ptr(), w.data_ptr(), g.data_ptr(), b.data_ptr(), stats.data_ptr(), dY.data_ptr(),
                                               dW.data_ptr(), dG.data_ptr(), dB.data_ptr(), ws.data_ptr(), B, S, C, k, stride, 1.0, hip.BF16,
                                               hip.stream()), "conv0 bwd")
            torch.cuda.synchronize()
            res.append((out, stats, dW, dG, dB))
    finally:
        hip.check(Ld.st5_conv0_set_fold(1), "set_fold")
    assert torch.isfinite(res[1][0].float()).all() and torch.isfinite(res[1][1]).all()
    assert torch.equal(res[0][1], res[1][1]), "statistics"
    assert torch.equal(res[0][0].view(torch.int16), res[1][0].view(torch.int16)), "output"
    # the backward: the fragment launch folds and publishes the waveform moments itself (no statistics launch) -- same bits
    for i, nm in ((2, "dW"), (3, "dgamma"), (4, "dbeta")):
        assert torch.isfinite(res[1][i]).all() and torch.equal(res[0][i], res[1][i]), nm
    # ... and with the moments carried over from the forward (st5_conv0_gn_gelu_fwd_m / _bwd_m: no pass over the waveform at all)
    out = torch.full((B, Lo, C), float("nan"), dtype=torch.bfloat16, device=cuda)
    stats = torch.full((B, C, 2), float("nan"), device=cuda)
    mom = torch.full((B, Ld.st5_conv0_mom_count(k)), float("nan"), dtype=torch.float64, device=cuda)
    assert mom.shape[1] == k + k * (k + 1) // 2
    hip.check(Ld.st5_conv0_gn_gelu_fwd_m(wav.data_ptr(), w.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(), stats.data_ptr(),
                                         mom.data_ptr(), ws.data_ptr(), B, S, C, k, stride, 1e-5, hip.BF16, hip.stream()), "conv0 fwd_m")
    ws.view(torch.uint8).fill_(0xFF)        # (nothing of the forward's workspace may be needed)
    dW = torch.zeros(C, k, device=cuda); dG = torch.zeros(C, device=cuda); dB = torch.zeros(C, device=cuda)
    hip.check(Ld.st5_conv0_gn_gelu_bwd_m(wav.data_ptr(), w.data_ptr(), g.data_ptr(), b.data_ptr(), stats.data_ptr(), mom.data_ptr(),
                                         dY.data_ptr(), dW.data_ptr(), dG.data_ptr(), dB.data_ptr(), ws.data_ptr(), B, S, C, k, stride, 1.0,
                                         hip.BF16, hip.stream()), "conv0 bwd_m")
    torch.cuda.synchronize()
    assert torch.isfinite(mom).all()
    assert torch.equal(out.view(torch.int16), res[1][0].view(torch.int16)) and torch.equal(stats, res[1][1])
    for t, i, nm in ((dW, 2, "dW"), (dG, 3, "dgamma"), (dB, 4, "dbeta")):
        assert torch.equal(t, res[1][i]), nm


@pytest.mark.parametrize("B,S,C,k", [(2, 6407, 128, 10), (1, 16000, 512, 10), (2, 4000, 256, 7), (1, 4000, 1024, 10)])   # (C = 1024: beyond the
# four channel tiles a wave of the matrix-core backward holds -- the dispatcher must take the VALU kernel there)
def test_conv0_matrix_core_backward_equals_the_valu_backward(cuda, B, S, C, k):
    """bf16 backward of conv layer 0 on the matrix cores (recomputed convolution and the dz . x products as split-bf16 MFMAs, dz
    transposed through LDS with the gfx950 transpose read, S2 derived from A and S1) against the VALU form on the same inputs: dW,
    dgamma, dbeta agree to 4e-3 of their scale (dz enters the matrix cores as ONE bf16 operand, as dY does in every weight-gradient GEMM of
    the bf16 mode; the waveform side is split and keeps ~16 bits)."""
    torch.manual_seed(S + C)
    stride = 5
    Lo = (S - k) // stride + 1
    wav = torch.randn(B, S, device=cuda)
    w = (torch.randn(C, k) * math.sqrt(2.0 / k)).to(cuda)
    g, b = (torch.rand(C) + 0.5).to(cuda), (torch.randn(C) * 0.1).to(cuda)
    dY = (torch.randn(B, Lo, C, device=cuda) * 0.3).to(torch.bfloat16)
    Ld = hip.lib()
    ws = hip.workspace(Ld.st5_conv0_ws_bytes(B, S, C, k, stride), cuda)
    res = []
    try:
        for mode in (0, 1):
            hip.check(Ld.st5_conv0_set_mfma(mode), "set_mfma")
            out = torch.empty(B, Lo, C, dtype=torch.bfloat16, device=cuda)
            stats = torch.empty(B, C, 2, device=cuda)
            hip.check(Ld.st5_conv0_gn_gelu_fwd(wav.data_ptr(), w.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(), stats.data_ptr(),
                                               ws.data_ptr(), B, S, C, k, stride, 1e-5, hip.BF16, hip.stream()), "conv0 fwd")
            dW = torch.zeros(C, k, device=cuda); dG = torch.zeros(C, device=cuda); dB = torch.zeros(C, device=cuda)
            hip.check(Ld.st5_conv0_gn_gelu_bwd(wav.data_ptr(), w.data_ptr(), g.data_ptr(), b.data_ptr(), stats.data_ptr(), dY.data_ptr(),
                                               dW.data_ptr(), dG.data_ptr(), dB.data_ptr(), ws.data_ptr(), B, S, C, k, stride, 1.0, hip.BF16,
                                               hip.stream()), "conv0 bwd")
            torch.cuda.synchronize()
            res.append((dW.clone(), dG.clone(), dB.clone()))
    finally:
        hip.check(Ld.st5_conv0_set_mfma(1), "set_mfma")
    for (a, m, nm) in zip(res[0], res[1], ("dW", "dgamma", "dbeta")):
        assert torch.isfinite(m).all(), nm
        err = float((a - m).abs().max()) / max(float(a.abs().max()), 1e-6)
        assert err <= 4e-3, (nm, err)


@pytest.mark.parametrize("dtype", DTYPES)
def test_elementwise_and_losses(cuda, dtype):
    torch.manual_seed(23)
    L = hip.lib()
    d, s = hip.dt(dtype), hip.stream()
    rows, cols = 77, 96
    x = torch.randn(rows, cols)
    X = dev(x, dtype, cuda)
    # casts (+ transpose)
    Wt = torch.empty(cols, rows, dtype=dtype, device=cuda)
    XF = x.to(cuda)
    hip.check(L.st5_cast_from_f32(XF.data_ptr(), Wt.data_ptr(), rows, cols, 1, d, s), "cast t")
    close(Wt, rt(x, dtype).t(), dtype, what="cast transpose")
    back = torch.empty(rows * cols, device=cuda)
    hip.check(L.st5_cast_to_f32(X.data_ptr(), back.data_ptr(), rows * cols, d, s), "cast back")
    assert torch.equal(back.cpu().view(rows, cols), rt(x, dtype))
    # colsum / sumsq
    out = torch.ones(cols, device=cuda)
    ws = hip.workspace(L.st5_colsum_ws_bytes(rows, cols), cuda)
    hip.check(L.st5_colsum_ws(X.data_ptr(), out.data_ptr(), ws.data_ptr(), rows, cols, cols, 2.0, 1, d, s), "colsum")
    close(out - 1, 2 * rt(x, dtype).sum(0), dtype, what="colsum")
    sq = torch.zeros(1, device=cuda)
    hip.check(L.st5_sumsq(X.data_ptr(), sq.data_ptr(), rows * cols, 1.0 / (rows * cols), 0, d, s), "sumsq")
    close(sq, rt(x, dtype).pow(2).mean().view(1), dtype, what="sumsq")
    # activation fwd/bwd
    for act, fn in ((hip.ACT_GELU, F.gelu), (hip.ACT_RELU, F.relu), (hip.ACT_TANH, torch.tanh)):
        Y = torch.empty_like(X)
        hip.check(L.st5_act_fwd(X.data_ptr(), Y.data_ptr(), rows * cols, act, d, s), "act")
        xr = rt(x, dtype).requires_grad_(True)
        yr = fn(xr)
        close(Y, yr, dtype, what=f"act {act}")
        yr.sum().backward()
        DX = torch.empty_like(X)
        ones = torch.ones_like(X)
        hip.check(L.st5_act_bwd(ones.data_ptr(), X.data_ptr(), DX.data_ptr(), rows * cols, act, d, s), "act bwd")
        close(DX, xr.grad, dtype, what=f"act bwd {act}")
    # masked fill rows fwd/bwd
    mask = (torch.rand(rows) < 0.4).to(torch.uint8)
    v = torch.randn(cols)
    X2 = X.clone()
    MK, Vd = mask.to(cuda), v.to(cuda)  # keep device temporaries alive across the async launches
    hip.check(L.st5_masked_fill_rows(X2.data_ptr(), MK.data_ptr(), Vd.data_ptr(), rows, cols, d, s), "mfill")
    ref = rt(x, dtype).clone(); ref[mask.bool()] = rt(v, dtype)
    close(X2, ref, dtype, what="masked fill")
    DXm = X.clone(); dv = torch.zeros(cols, device=cuda)
    hip.check(L.st5_masked_fill_rows_bwd(DXm.data_ptr(), MK.data_ptr(), dv.data_ptr(), rows, cols, d, s), "mfill bwd")
    close(dv, rt(x, dtype)[mask.bool()].sum(0), dtype, what="mask_emb grad")
    assert float(DXm[mask.bool().to(cuda)].abs().sum()) == 0
    # table add / embedding
    table = torch.randn(50, cols)
    idx = torch.randint(0, 50, (rows,), dtype=torch.int32)
    Y = torch.empty_like(X)
    TB, IX = table.to(cuda), idx.to(cuda)
    hip.check(L.st5_add_table_rows(X.data_ptr(), TB.data_ptr(), IX.data_ptr(), Y.data_ptr(), rows, cols, 0.5, d, s), "addtab")
    close(Y, rt(x, dtype) + 0.5 * table[idx.long()], dtype, what="add table rows")
    hip.check(L.st5_embed_rows(TB.data_ptr(), IX.data_ptr(), TB.data_ptr(), IX.data_ptr(), Y.data_ptr(), rows, cols, 2.0, 1.0, d, s), "embed")
    close(Y, 3.0 * table[idx.long()], dtype, what="embed rows")
    dT = torch.zeros(50, cols, device=cuda)
    hip.check(L.st5_embed_rows_bwd(X.data_ptr(), IX.data_ptr(), dT.data_ptr(), rows, cols, 1.0, d, s), "embed bwd")
    refT = torch.zeros(50, cols).index_add_(0, idx.long(), rt(x, dtype))
    close(dT, refT, dtype, what="embed bwd")
    # cross entropy (label smoothing, ignore index, -inf logits)
    V, ld = 83, 88
    logits = torch.randn(rows, V) * 3
    logits[:, 5] = float("-inf")
    tgt = torch.randint(0, V, (rows,), dtype=torch.int32); tgt[tgt == 5] = 6; tgt[:7] = 1
    Lg = torch.zeros(rows, ld); Lg[:, :V] = logits
    LG = dev(Lg, dtype, cuda)
    loss = torch.zeros(1, device=cuda); nll = torch.zeros(1, device=cuda)
    DL = torch.empty_like(LG)
    TG = tgt.to(cuda)
    hip.check(L.st5_cross_entropy(LG.data_ptr(), TG.data_ptr(), loss.data_ptr(), nll.data_ptr(), DL.data_ptr(), rows, V, ld, 0.0, 1, 0.5, d, s), "ce")
    lr = rt(logits, dtype).requires_grad_(True)
    ref = F.nll_loss(F.log_softmax(lr, -1), tgt.long(), ignore_index=1, reduction="sum")
    close(loss, ref.view(1), dtype, what="ce loss")
    (0.5 * ref).backward()
    close(DL[:, :V], lr.grad, dtype, scale=1.0, what="ce grad")


@pytest.mark.parametrize("dtype", DTYPES)
def test_channel_mask(cuda, dtype):
    """apply_hubert_mask's channel masking (speech_encoder_prenet.py:253-270): x[b, :, c] = 0, gradient likewise."""
    from speecht5_amd import functional as Fn
    torch.manual_seed(23)
    B, T, C = 3, 17, 64
    x = torch.randn(B, T, C)
    m = torch.rand(B, C) < 0.3
    X = dev(x, dtype, cuda).requires_grad_(True)
    old = Fn.get_compute_dtype() if hasattr(Fn, "get_compute_dtype") else None
    y = Fn.mask_channels(X, m.to(cuda))
    ref = rt(x, dtype).masked_fill(m.unsqueeze(1).expand(-1, T, -1), 0.0)
    close(y, ref, dtype, what="channel mask fwd")
    g = torch.randn(B, T, C)
    y.backward(dev(g, dtype, cuda))
    close(X.grad, rt(g, dtype).masked_fill(m.unsqueeze(1).expand(-1, T, -1), 0.0), dtype, what="channel mask bwd")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,V,eps", [(37, 505, 0.0), (130, 83, 0.1), (8, 10003, 0.1), (0, 7, 0.0)])
def test_cross_entropy_sum_function(cuda, dtype, rows, V, eps):
    """functional.cross_entropy_sum (st5_cross_entropy_rows: per-row losses, caller-side sum) vs the reference's
    label_smoothed_nll_loss on log_softmax (speech_to_text_loss.py:93-110), with -inf logits and ignored rows."""
    torch.manual_seed(rows + V)
    x = torch.randn(rows, V) * 3
    if rows > 2:
        x[1, 3] = float("-inf")
    t = torch.randint(0, V, (rows,))
    pad = 1
    if rows > 4:
        t[4] = pad
    X = dev(x, dtype, cuda).requires_grad_(True)
    loss, nll = Fn.cross_entropy_sum(X, t.to(cuda), eps, pad)
    (loss * 0.5).backward()
    xr = rt(x, dtype).requires_grad_(True)
    lp = F.log_softmax(xr, -1)
    nl = -lp.gather(1, t[:, None])
    finite = torch.isfinite(lp)
    sm = -(torch.where(finite, lp, torch.zeros_like(lp))).sum(-1, keepdim=True)
    keep = (t != pad)[:, None]
    nl, sm = nl * keep, sm * keep
    eps_i = eps / (V - 1)
    ref = (1 - eps - eps_i) * nl.sum() + eps_i * sm.sum()
    if rows:
        (ref * 0.5).backward()
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-4
    assert abs(loss.item() - ref.item()) <= tol * max(1.0, abs(ref.item()))
    assert abs(nll.item() - nl.sum().item()) <= tol * max(1.0, abs(nl.sum().item()))
    if rows:
        g = X.grad.float().cpu()
        gr = torch.nan_to_num(xr.grad, nan=0.0)
        assert (g - gr).abs().max().item() <= tol * max(1e-3, gr.abs().max().item()) + (1e-2 if dtype == torch.bfloat16 else 1e-6)


@pytest.mark.parametrize("block", ["linear", "ffn"])
def test_layernorm_backward_hands_dropped_gradient_to_the_linear_in_front(cuda, block, monkeypatch):
    """Post-LN block  y = LN(x + drop(Linear(x)))  (transformer_layer.py:122-134): the LayerNorm backward kernel's second
    output dX * mask replaces the dropout kernel of the Linear's backward.  Same seed with and without the hand-off must
    give bit-identical gradients, and the hand-off must actually have been used."""
    dt = torch.bfloat16
    Fn.set_compute_dtype(dt)
    try:
        torch.manual_seed(11)
        M, d, Fd = 300, 256, 512
        x0 = torch.randn(M, d).to(dt).to(cuda)
        lin = torch.nn.Linear(d, d).to(cuda)
        fc1, fc2 = torch.nn.Linear(d, Fd).to(cuda), torch.nn.Linear(Fd, d).to(cuda)
        lw, lb = torch.randn(d, device=cuda).requires_grad_(True), torch.randn(d, device=cuda).requires_grad_(True)
        gout = torch.randn(M, d).to(dt).to(cuda)
        params = [lw, lb] + list(lin.parameters()) + list(fc1.parameters()) + list(fc2.parameters())

        def run():
            Fn.manual_seed(99)
            Fn.weight_cache.clear()
            for p in params:
                p.grad = None
            x = x0.clone().requires_grad_(True)
            if block == "linear":
                y = Fn.linear(x, lin.weight, lin.bias, residual=x, dropout_p=0.3)
            else:
                y = Fn.ffn(x, x, fc1, fc2, p_act=0.0, p_out=0.3)
            out = Fn.layer_norm(y, lw, lb)
            out.backward(gout)
            torch.cuda.synchronize()
            return [x.grad.clone()] + [p.grad.clone() for p in params if p.grad is not None]

        calls = []
        orig = Fn._dropout
        monkeypatch.setattr(Fn, "_dropout", lambda x, p, s: (calls.append(1), orig(x, p, s))[1])
        with_handoff = run()
        n_with = len(calls)
        monkeypatch.setattr(Fn, "_tag_dropout_output", lambda *a: None)      # no tag -> LayerNorm emits no second output
        without = run()
        assert n_with == 0 and len(calls) == 1
        assert len(with_handoff) == len(without) >= 4
        for a, b in zip(with_handoff, without):
            assert torch.equal(a, b)
    finally:
        Fn.set_compute_dtype(torch.float32)
        Fn.weight_cache.clear()


def test_conv0_statistics_are_run_to_run_deterministic(cuda):
    """The GroupNorm statistics of conv layer 0 feed every later activation through bf16 rounding, so they must not depend
    on wave arrival order (a fixed-order reduction, no fp32 atomics): repeated launches, interleaved with other work that
    shifts the timing, give bit-identical (mean, rstd) and outputs."""
    torch.manual_seed(3)
    B, S, C, k, stride = 4, 48000, 512, 10, 5
    wav = (torch.randn(B, S) * 0.3).to(cuda)
    w = (torch.randn(C, k) * math.sqrt(2.0 / k)).to(cuda)
    g, b = (torch.rand(C) + 0.5).to(cuda), (torch.randn(C) * 0.1).to(cuda)
    Lo = (S - k) // stride + 1
    Ld = hip.lib()
    ws = hip.workspace(Ld.st5_conv0_ws_bytes(B, S, C, k, stride), cuda)
    first = None
    for i in range(25):
        if i % 3:
            torch.randn(1 << (14 + i % 5), device=cuda).sum()      # unrelated kernels in between
        out = torch.empty(B, Lo, C, dtype=torch.bfloat16, device=cuda)
        stats = torch.empty(B, C, 2, device=cuda)
        hip.check(Ld.st5_conv0_gn_gelu_fwd(wav.data_ptr(), w.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(),
                                           stats.data_ptr(), ws.data_ptr(), B, S, C, k, stride, 1e-5, hip.BF16, hip.stream()),
                  "conv0 fwd")
        torch.cuda.synchronize()
        if first is None:
            first = (stats.clone(), out.clone())
        else:
            assert torch.equal(stats, first[0]) and torch.equal(out, first[1]), f"launch {i} differs"


def test_layernorm_deferred_parameter_gradients(cuda):
    """st5_layernorm_defer / st5_layernorm_flush: block partials of several LayerNorm backwards folded by one launch, the
    same parameter twice in a batch (two micro-batches) included -- equal to the immediate per-call reduction."""
    torch.manual_seed(17)
    L = hip.lib()
    dt = torch.bfloat16
    shapes = [(500, 768), (2100, 768), (64, 256), (500, 768)]      # the last one reuses the first one's parameters
    xs = [dev(torch.randn(r, c), dt, cuda) for r, c in shapes]
    dys = [dev(torch.randn(r, c), dt, cuda) for r, c in shapes]
    gam = [torch.randn(c, device=cuda) for _, c in shapes]
    means = [torch.randn(r, device=cuda) * 0.1 for r, _ in shapes]
    rstds = [torch.rand(r, device=cuda) + 0.5 for r, _ in shapes]

    def run(defer):
        dG = [torch.zeros(c, device=cuda) for _, c in shapes[:3]]
        dB = [torch.zeros(c, device=cuda) for _, c in shapes[:3]]
        hip.check(L.st5_layernorm_defer(1 if defer else 0, hip.stream()), "defer")
        try:
            for i, (r, c) in enumerate(shapes):
                p = 0 if i == 3 else i
                dx = torch.empty_like(xs[i])
                ws = hip.workspace(L.st5_layernorm_bwd_ws_bytes(r, c), cuda)
                hip.check(L.st5_layernorm_bwd(dys[i].data_ptr(), xs[i].data_ptr(), gam[p].data_ptr(), means[i].data_ptr(),
                                              rstds[i].data_ptr(), dx.data_ptr(), dG[p].data_ptr(), dB[p].data_ptr(), ws.data_ptr(),
                                              r, c, None, 0.0, 0, hip.BF16, hip.stream()), "ln bwd")
            hip.check(L.st5_layernorm_flush(hip.stream()), "flush")
        finally:
            hip.check(L.st5_layernorm_defer(0, hip.stream()), "defer off")
        torch.cuda.synchronize()
        return dG, dB

    a, b = run(False), run(True)
    for u, v in zip(a[0] + a[1], b[0] + b[1]):
        assert torch.equal(u, v)
    assert a[0][0].abs().max().item() > 0


@pytest.mark.parametrize("k,stride,L", [(3, 2, 401), (2, 2, 300)])
def test_conv_weight_gradient_and_data_gradient_with_row_split_operands(cuda, k, stride, L):
    """The haloed per-utterance layouts of the convolution backward (operands split every `rpb` rows):
    weight gradient = TN-form GEMM on the LDS-DMA kernel with row-split k-strided operands and split-K; data gradient with the
    act' epilogue writing a row-split output (vectorised epilogue path).  Against fp32 autograd of the same convolution."""
    torch.manual_seed(5)
    dtype = torch.bfloat16
    B, Cin, Cout = 3, 64, 128
    Lo = (L - k) // stride + 1
    x = torch.randn(B, L, Cin)
    w = torch.randn(Cout, Cin, k) / math.sqrt(Cin * k)
    dy = torch.randn(B, Lo, Cout)
    xr, wr, dyr = rt(x, dtype).requires_grad_(True), rt(w, dtype).requires_grad_(True), rt(dy, dtype)
    y = F.conv1d(xr.transpose(1, 2), wr, stride=stride).transpose(1, 2)
    y.backward(dyr)
    # ---- weight gradient: dW[co, j*Cin + ci] = sum_{b,t} dY[b,t,co] * X[b, t*s + j, ci]; dY kept with one halo row per side
    dpre = torch.zeros(B, Lo + 2, Cout, dtype=dtype, device=cuda)
    dpre[:, 1:-1] = dy.to(dtype)
    X = dev(x, dtype, cuda)
    gw = torch.zeros(Cout, k * Cin, device=cuda)
    hip.gemm(hip.operand(dpre, Cout, off=Cout, rpb=Lo, bstride=(Lo + 2) * Cout), hip.operand(X, stride * Cin, rpb=Lo, bstride=L * Cin),
             hip.operand(gw, k * Cin), Cout, k * Cin, B * Lo, hip.BF16, flags=hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32)
    ref_w = wr.grad.permute(0, 2, 1).reshape(Cout, k * Cin)
    close(gw, ref_w, dtype, what="conv weight gradient (row-split TN)")
    if k != 2:
        return
    # ---- data gradient (k = 2, stride 2: one NT GEMM) times gelu'(P), written into a haloed [B, L + 2, Cin] buffer
    P = torch.randn(B, L, Cin)
    Wd = dev(w.permute(2, 1, 0).reshape(k * Cin, Cout), dtype, cuda)            # [k*Cin, Cout]
    nxt = torch.zeros(B, L + 2, Cin, dtype=dtype, device=cuda)
    Pd = dev(P, dtype, cuda)
    hip.gemm(hip.operand(dpre, Cout, off=Cout, rpb=Lo, bstride=(Lo + 2) * Cout), hip.operand(Wd, Cout),
             hip.operand(nxt, 2 * Cin, off=Cin, rpb=Lo, bstride=(L + 2) * Cin), B * Lo, 2 * Cin, Cout, hip.BF16,
             P=hip.operand(Pd, 2 * Cin, rpb=Lo, bstride=L * Cin), act=hip.ACT_GELU, flags=hip.DACT)
    Pr = rt(P, dtype).requires_grad_(True)
    F.gelu(Pr).backward(xr.grad)                         # gelu'(P) * dX
    close(nxt[:, 1:-1][:, :2 * Lo], Pr.grad[:, :2 * Lo], dtype, what="conv data gradient * gelu' (row-split epilogue)")
    assert float(nxt[:, 0].abs().max()) == 0 and float(nxt[:, -1].abs().max()) == 0   # halo rows untouched


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,vocab,cols", [(8192, 83, 768), (5000, 10000, 96), (37, 5, 8)])
def test_deterministic_row_scatter(cuda, dtype, rows, vocab, cols):
    """st5_embed_rows_bwd_det(_w): dtable[tok[r]] += scale * w[r] * dy[r] without atomics -- equals index_add to fp32 round-off,
    is bit-identical from run to run, leaves rows of out-of-range ids alone, honours the per-row weight."""
    torch.manual_seed(3)
    tok = torch.randint(0, vocab, (rows,), dtype=torch.int32)
    tok[::7] = 1                                   # one frequent id (a long run for a single block)
    tok[5] = -1
    dy = torch.randn(rows, cols)
    T, div = 11, 2
    w = torch.rand(T)
    L = hip.lib()
    base = torch.randn(vocab, cols)
    outs = []
    dy_d, tok_d, w_d = dev(dy, dtype, cuda), tok.to(cuda), w.to(cuda)
    for _ in range(2):
        table = base.clone().to(cuda)
        hip.check(L.st5_embed_rows_bwd_det_w(dy_d.data_ptr(), tok_d.data_ptr(), table.data_ptr(), rows, cols, vocab, 0.5, w_d.data_ptr(), div, T,
                                             hip.dt(dtype), hip.stream()), "st5_embed_rows_bwd_det_w")
        outs.append(table.cpu())
    assert torch.equal(outs[0], outs[1])
    ok = tok >= 0
    rw = w[(torch.arange(rows) // div) % T]
    ref = base.clone().index_add_(0, tok[ok].long(), 0.5 * rw[ok, None] * rt(dy, dtype)[ok])
    close(outs[0], ref, torch.float32, what="row scatter")


def test_ctc_loss_matches_torch(cuda):
    """st5_ctc_loss_fwd / _bwd against torch's own CTC recursion on the host (what the reference runs with cuDNN off,
    speech_to_text_loss.py:333-337): ragged frame and label counts, repeated labels (the l'_s != l'_{s-2} rule), an empty
    target, a sentence with more labels than frames (impossible: zeroed by zero_infinity), frames past a sentence's length."""
    from speecht5_amd import functional as Fn
    torch.manual_seed(3)
    T, B, V, blank = 37, 6, 11, 4
    logits = torch.randn(T, B, V) * 2.0
    in_len = torch.tensor([37, 30, 12, 37, 5, 1])
    tgs = [[1, 1, 2, 3, 3, 3, 5], [7, 8, 7, 8, 7], [], [9] * 18, [1, 2, 3, 1, 2, 3, 1, 2], [6]]
    tg_len = torch.tensor([len(t) for t in tgs])
    flat = torch.tensor([c for t in tgs for c in t], dtype=torch.long)
    for scale in (1.0, 0.37):
        ref_in = logits.clone().double().requires_grad_(True)
        ref = F.ctc_loss(ref_in.log_softmax(-1), flat, in_len, tg_len, blank=blank, reduction="sum", zero_infinity=True)
        (ref * scale).backward()
        x = logits.to(cuda).requires_grad_(True)
        got = Fn.ctc_loss_sum(x.log_softmax(-1), flat.to(cuda), in_len.to(cuda), tg_len.to(cuda), blank, True, max_target_len=20)
        (got * scale).backward()
        assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref)), (float(got), float(ref))
        err = float((x.grad.cpu().double() - ref_in.grad).abs().max())
        assert err <= 1e-4 * float(ref_in.grad.abs().max()), err   # (fp32 log-space recursions vs fp64)
    # without zero_infinity the impossible sentence makes the sum infinite, as in torch
    got = Fn.ctc_loss_sum(logits.to(cuda).log_softmax(-1), flat.to(cuda), in_len.to(cuda), tg_len.to(cuda), blank, False, max_target_len=20)
    assert torch.isinf(got)


def test_guided_attention_loss_matches_formula(cuda):
    """st5_guided_attn_fwd / _bwd against the reference formula (text_to_speech_loss.py:370-427) in fp64."""
    from speecht5_amd import functional as Fn
    torch.manual_seed(4)
    B, H, To, Ti = 3, 4, 29, 17
    att = torch.rand(B, H, To, Ti)
    ilens, olens = torch.tensor([17, 9, 13]), torch.tensor([29, 20, 5])
    a = att.double().requires_grad_(True)
    gx = torch.arange(To).double()[None, :, None] / olens[:, None, None]
    gy = torch.arange(Ti).double()[None, None, :] / ilens[:, None, None]
    w = 1.0 - torch.exp(-((gy - gx) ** 2) / (2 * 0.4 ** 2))
    mask = (torch.arange(To)[None, :, None] < olens[:, None, None]) & (torch.arange(Ti)[None, None, :] < ilens[:, None, None])
    ref = 10.0 * torch.mean((w.unsqueeze(1) * a).masked_select(mask.unsqueeze(1).expand(B, H, To, Ti)))
    (ref * 0.5).backward()
    x = att.to(cuda).requires_grad_(True)
    got = Fn.guided_attention_loss(x, ilens, olens, 0.4, 10.0)
    (got * 0.5).backward()
    assert abs(float(got) - float(ref)) <= 2e-6 * abs(float(ref)), (float(got), float(ref))
    assert float((x.grad.cpu().double() - a.grad).abs().max()) <= 2e-6 * float(a.grad.abs().max())


def test_pos_conv_grouped_bf16_matches_torch(cuda):
    """Positional convolution (speech_encoder_prenet.py:107-118: grouped Conv1d k = 128, 16 groups, SamePad, GELU, residual) in bf16:
    forward, data gradient and -- through the LDS-DMA TN kernel with a SEGMENTED operand (the haloed activations read as
    [tap][channel-in-group] windows) -- the weight gradient, against torch on the same bf16-rounded inputs."""
    from speecht5_amd import functional as Fn
    torch.manual_seed(11)
    B, T, d, groups, k = 3, 200, 768, 16, 128
    cg = d // groups
    Fn.set_compute_dtype(torch.bfloat16)
    try:
        x = (torch.randn(B, T, d, device=cuda) * 0.5).to(torch.bfloat16).requires_grad_(True)
        w = (torch.randn(d, cg, k, device=cuda) * 0.02).requires_grad_(True)
        bias = (torch.randn(d, device=cuda) * 0.1).requires_grad_(True)
        dy = torch.randn(B, T, d, device=cuda).to(torch.bfloat16)
        y = Fn.pos_conv(x, w, bias, groups)
        y.backward(dy)
        xr = x.detach().float().requires_grad_(True)
        wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)
        br = bias.detach().clone().requires_grad_(True)
        conv = F.conv1d(xr.transpose(1, 2), wr, br, padding=k // 2, groups=groups)[:, :, :T]
        yr = xr + F.gelu(conv).transpose(1, 2)
        yr.backward(dy.float())
        def rel(a, b):
            return float((a.detach().float() - b.detach()).norm() / b.detach().norm())
        assert rel(y, yr) < 1e-2, rel(y, yr)
        assert rel(x.grad, xr.grad) < 2e-2, rel(x.grad, xr.grad)
        assert rel(w.grad, wr.grad) < 2e-2, rel(w.grad, wr.grad)
        assert rel(bias.grad, br.grad) < 2e-2, rel(bias.grad, br.grad)
    finally:
        Fn.set_compute_dtype(torch.float32)
        Fn.weight_cache.clear()

"""MX-fp8 path (BASELINE.json configs[4]: SpeechT5-Large with fp8 MFMA GEMMs; csrc/gemm.hip gemm_nt_mx8_kernel) through the C ABI:
  * st5_quant_mxfp8 against a plain-torch restatement of the OCP MX rule (scale = floor(log2(amax)) - 8, e4m3 round-to-nearest-even,
    saturation): scale bytes and fp8 bytes BIT-EXACT;
  * st5_gemm_mxfp8 against fp32 torch matmul of the DEQUANTISED operands (so only the MFMA's own fp32 accumulation differs: 2e-3 of
    the output scale), every fused epilogue class the training step uses, odd M / N tails;
  * end to end: the fp8 Linear against the bf16 Linear on the same inputs (the quantisation error itself: cosine >= 0.995)."""
import pytest
import torch

from speecht5_amd import hip

pytestmark = pytest.mark.gpu


def _mx_ref(x):
    """(q uint8, s uint8, dequantised fp32) of a bf16 matrix, MX blocks of 32 along rows -- plain torch."""
    rows, cols = x.shape
    xb = x.float().view(rows, cols // 32, 32)
    amax = xb.abs().amax(-1)
    e_b = (amax.view(torch.int32) >> 23) & 0xff                  # biased exponent of amax
    E = (e_b - 8).clamp(0, 254)
    inv = ((254 - E) << 23).view(torch.float32)                  # 2^(127 - E)
    y = (xb * inv.unsqueeze(-1)).clamp(-448.0, 448.0)
    q8 = y.to(torch.float8_e4m3fn)
    deq = q8.float() * torch.pow(torch.tensor(2.0, device=x.device), (E - 127).float()).unsqueeze(-1)
    return q8.view(torch.uint8).view(rows, cols), E.to(torch.uint8), deq.view(rows, cols)


def _deq(q, s):
    rows, cols = q.shape
    v = q.view(torch.float8_e4m3fn).float().view(rows, cols // 32, 32)
    return (v * torch.pow(torch.tensor(2.0, device=q.device), s.float() - 127.0).unsqueeze(-1)).view(rows, cols)


@pytest.mark.parametrize("rows,cols,scale", [(257, 1024, 1.0), (64, 4096, 37.0), (5, 128, 1e-3)])
def test_quantiser_is_the_mx_rule_bit_for_bit(cuda, rows, cols, scale):
    torch.manual_seed(rows)
    x = (torch.randn(rows, cols, device=cuda) * scale)
    x[0, :32] = 0.0                                  # an all-zero block
    x[1, 5] = 3e4 * scale                            # an outlier that sets its block's scale
    x = x.to(torch.bfloat16)
    q, s = hip.quant_mxfp8(x)
    rq, rs, _ = _mx_ref(x)
    torch.cuda.synchronize()
    assert torch.equal(s, rs), f"{int((s != rs).sum())} scale bytes differ"
    assert torch.equal(q, rq), f"{int((q != rq).sum())} of {q.numel()} fp8 bytes differ"


@pytest.mark.parametrize("M,N,K", [(256, 256, 1024), (4096, 1024, 1024), (1000, 4096, 1024), (333, 520, 4096), (8192, 1024, 4096)])
def test_mx_gemm_matches_fp32_matmul_of_the_dequantised_operands(cuda, M, N, K):
    torch.manual_seed(M + N)
    A = torch.randn(M, K, device=cuda).to(torch.bfloat16)
    B = (torch.randn(N, K, device=cuda) * 0.05).to(torch.bfloat16)
    B[3] *= 40.0                                                         # rows with very different scales
    Aq, As = hip.quant_mxfp8(A)
    Bq, Bs = hip.quant_mxfp8(B)
    ref = _deq(Aq, As) @ _deq(Bq, Bs).t()
    ldn = (N + 7) // 8 * 8
    C = torch.zeros(M, ldn, dtype=torch.bfloat16, device=cuda)
    hip.gemm_mxfp8(Aq, As, Bq, Bs, hip.operand(C, ldn), M, N, K)
    torch.cuda.synchronize()
    got = C[:, :N].float()
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    assert torch.isfinite(got).all() and err <= 6e-3, err                # bf16 output rounding (2^-8) dominates
    # the fused epilogues of the training step: bias + GELU + pre-activation copy; bias + dropout 0 + residual; x act'(P); beta
    if N % 8 == 0:
        bias = torch.randn(N, device=cuda)
        R = torch.randn(M, N, device=cuda).to(torch.bfloat16)
        pre = torch.empty_like(C)
        y = torch.empty_like(C)
        hip.gemm_mxfp8(Aq, As, Bq, Bs, hip.operand(y, ldn), M, N, K, bias=bias, act=hip.ACT_GELU, Cpre=hip.operand(pre, ldn))
        torch.cuda.synchronize()
        want_pre = ref + bias
        assert float((pre.float() - want_pre).abs().max()) <= 1e-2 * float(want_pre.abs().max())
        assert float((y.float() - torch.nn.functional.gelu(pre.float())).abs().max()) <= 1e-2 * float(want_pre.abs().max())
        y2 = torch.empty_like(C)
        hip.gemm_mxfp8(Aq, As, Bq, Bs, hip.operand(y2, ldn), M, N, K, bias=bias, R=hip.operand(R, N))
        torch.cuda.synchronize()
        assert float((y2.float() - (want_pre + R.float())).abs().max()) <= 1e-2 * float(want_pre.abs().max())


def test_fp8_linear_against_bf16_linear(cuda):
    """The quantisation error of one Linear at Large's shapes: cosine with the bf16 result."""
    torch.manual_seed(0)
    M, N, K = 2048, 4096, 1024
    X = torch.randn(M, K, device=cuda).to(torch.bfloat16)
    W = (torch.randn(N, K, device=cuda) * K ** -0.5).to(torch.bfloat16)
    Y = torch.empty(M, N, dtype=torch.bfloat16, device=cuda)
    hip.gemm(hip.operand(X, K), hip.operand(W, K), hip.operand(Y, N), M, N, K, hip.BF16)
    Xq, Xs = hip.quant_mxfp8(X)
    Wq, Ws = hip.quant_mxfp8(W)
    Y8 = torch.empty_like(Y)
    hip.gemm_mxfp8(Xq, Xs, Wq, Ws, hip.operand(Y8, N), M, N, K)
    torch.cuda.synchronize()
    cos = float(torch.nn.functional.cosine_similarity(Y.float().flatten(), Y8.float().flatten(), dim=0))
    print("fp8 vs bf16 Linear cosine", cos)
    assert cos >= 0.995

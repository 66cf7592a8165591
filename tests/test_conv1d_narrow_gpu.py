"""csrc/conv1d_narrow.hip (few-output-channel Conv1d on MFMA, the HiFi-GAN late stages) through speecht5_amd.hifigan's conv1d /
conv_transpose1d against torch's own convolutions in fp32 on the same bf16-rounded operands: dilations, kernel sizes, ragged
lengths (not a multiple of the 1024-step block tile, shorter than one wave tile), residual, running sum, output written into a
padded buffer, the transposed convolution's interleaved phases, and the one-output-channel kernel.  Tolerance: bf16 output rounding
(2^-8 relative) plus fp32 summation order."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _r(t):
    return t.to(torch.bfloat16).float()


@pytest.fixture
def bf16_mode():
    from speecht5_amd import functional as Fn
    Fn.set_compute_dtype(torch.bfloat16)
    yield
    Fn.set_compute_dtype(torch.float32)
    Fn.weight_cache.clear()


@pytest.mark.parametrize("cin,cout,k,d,L,B", [(32, 32, 3, 1, 1500, 2), (32, 32, 11, 5, 2049, 2), (64, 64, 7, 3, 1024, 3), (64, 64, 11, 1, 70, 1),
                                             (32, 32, 7, 1, 37, 2), (64, 64, 3, 5, 4100, 1)])
def test_narrow_conv1d_matches_torch(cuda, bf16_mode, cin, cout, k, d, L, B):
    from speecht5_amd import hifigan as H, hip
    torch.manual_seed(cin + k + d + L)
    conv = nn.Conv1d(cin, cout, k, dilation=d, padding=(k * d - d) // 2).to(cuda)
    with torch.no_grad():
        conv.weight.copy_(_r(conv.weight * 3))
    x = _r(torch.randn(B, L, cin, device=cuda))
    res = _r(torch.randn(B, L, cout, device=cuda))
    ref = F.conv1d(x.transpose(1, 2), conv.weight, conv.bias, dilation=d, padding=(k * d - d) // 2).transpose(1, 2)     # fp32
    p = H.conv_pad(conv)
    xp = H._pad_act(x.to(torch.bfloat16), p, p)
    scale = float(ref.abs().max())

    def check(got, want, what):
        err = float((got.float() - want).abs().max())
        assert err <= 6e-3 * max(float(want.abs().max()), scale), f"{what}: {err:.3e} (scale {scale:.3e})"
    # plain, alpha, LeakyReLU
    y = H.conv1d(xp, L, conv, alpha=0.5, act=hip.ACT_LRELU_01)
    check(y, F.leaky_relu(0.5 * (ref - conv.bias) + conv.bias, 0.1), "alpha + LeakyReLU")
    # residual, then the running sum into the same buffer
    y = H.conv1d(xp, L, conv, residual=res.to(torch.bfloat16))
    check(y, ref + res, "residual")
    y0 = _r(y.float())
    H.conv1d(xp, L, conv, residual=res.to(torch.bfloat16), out=y, beta=1.0)
    check(y, ref + res + y0, "running sum")
    # output written into the interior of a padded buffer with zero halo
    yp = H.conv1d(xp, L, conv, act=hip.ACT_LRELU_01, out_pad=5)
    assert yp.shape == (B, L + 10, cout)
    assert not yp[:, :5].any() and not yp[:, L + 5:].any()
    check(yp[:, 5:L + 5], F.leaky_relu(ref, 0.1), "padded output")


@pytest.mark.parametrize("cin,cout,L,B", [(128, 64, 300, 2), (64, 32, 1030, 2)])
def test_narrow_transposed_conv_matches_torch(cuda, bf16_mode, cin, cout, L, B):
    from speecht5_amd import hifigan as H, hip
    torch.manual_seed(cin + L)
    ct = nn.ConvTranspose1d(cin, cout, kernel_size=8, stride=4, padding=2).to(cuda)
    with torch.no_grad():
        ct.weight.copy_(_r(ct.weight * 3))
    x = _r(torch.randn(B, L, cin, device=cuda))
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1).to(torch.bfloat16).float().transpose(1, 2), ct.weight, ct.bias, stride=4, padding=2).transpose(1, 2)
    y = H.conv_transpose1d(H._pad_act(x.to(torch.bfloat16), 1, 1, hip.ACT_LRELU_01), L, ct)
    assert y.shape == (B, 4 * L, cout)
    err = float((y.float() - ref).abs().max())
    assert err <= 6e-3 * float(ref.abs().max()), err


def test_single_output_channel_conv_matches_torch(cuda, bf16_mode):
    from speecht5_amd import hifigan as H, hip
    torch.manual_seed(5)
    conv = nn.Conv1d(32, 1, 7, padding=3).to(cuda)
    with torch.no_grad():
        conv.weight.copy_(_r(conv.weight * 2))
    B, L = 3, 3001
    x = _r(torch.randn(B, L, 32, device=cuda))
    ref = torch.tanh(0.25 * (F.conv1d(x.transpose(1, 2), conv.weight, None, padding=3)) + conv.bias.view(1, 1, 1)).transpose(1, 2)
    y = H.conv1d(H._pad_act(x.to(torch.bfloat16), 3, 3), L, conv, alpha=0.25, act=hip.ACT_TANH)
    assert y.shape == (B, L, 1)
    assert float((y.float() - ref).abs().max()) <= 6e-3


def test_narrow_conv_rejects_what_it_cannot_address(cuda):
    from speecht5_amd import hip
    x = torch.zeros(1, 64, 32, dtype=torch.bfloat16, device=cuda)
    w = torch.zeros(32, 96, dtype=torch.bfloat16, device=cuda)
    y = torch.zeros(1, 62, 32, dtype=torch.bfloat16, device=cuda)
    L = hip.lib()
    args = lambda cin, cout, x_ts: (x.data_ptr(), 64 * 32, x_ts, w.data_ptr(), None, y.data_ptr(), 62 * 32, 32, None, 0, 0, 1, 62, cin, cout, 3, 32,
                                    1.0, 0.0, 0, hip.stream())
    assert L.st5_conv1d_narrow(*args(32, 48, 32)) == 1       # ST5_ERR_ARG: 48 output channels
    assert L.st5_conv1d_narrow(*args(24, 32, 32)) == 1       # ST5_ERR_ARG: 24 input channels
    assert L.st5_conv1d_narrow(*args(32, 32, 30)) == 2       # ST5_ERR_ALIGN: time stride not a multiple of 8 elements
    assert L.st5_conv1d_narrow(*args(32, 32, 32)) == 0
    torch.cuda.synchronize()

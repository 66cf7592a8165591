"""HiFi-GAN generator on the gfx950 GEMM kernel, inference only.

The reference tree has no vocoder (`SpeechT5/README.md:250` links an external one); the only concrete spec is
HuggingFace `SpeechT5HifiGan` (transformers/models/speecht5/modeling_speecht5.py:2887-3066).  This module keeps
that class's parameter names (`conv_pre`, `upsampler.{i}`, `resblocks.{j}.convs1/convs2.{k}`, `conv_post`, `mean`,
`scale`) so `microsoft/speecht5_hifigan` state dicts load, and computes on channels-last activations:

* Conv1d (dilated, "same" padding)  -> implicit GEMM over a zero-padded input, K segmented per tap
  (`seg = C, seg_stride = dilation*C`), bias + LeakyReLU / residual / running-sum fused in the epilogue; the padded input is
  made together with the LeakyReLU that precedes every convolution (`st5_pad_time_act`, one vectorised pass), or written
  directly by the producing convolution's epilogue (`out_pad`);
* ConvTranspose1d(k = 2*stride, pad = stride/2) -> `stride` phase GEMMs, each reading two adjacent input rows
  ([x[t-1], x[t]] . [W[:,:,j+stride]; W[:,:,j]]) and writing every stride-th output row;
* the mean over the resblocks is folded into the next convolution's alpha (LeakyReLU is positively homogeneous).
"""
import torch
import torch.nn as nn

from . import functional as Fn
from . import hip


class HifiGanResidualBlock(nn.Module):
    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5), leaky_relu_slope=0.1):
        super().__init__()
        assert abs(leaky_relu_slope - 0.1) < 1e-12
        self.dilation = dilation
        pad = lambda k, d: (k * d - d) // 2  # noqa: E731
        self.convs1 = nn.ModuleList([nn.Conv1d(channels, channels, kernel_size, 1, dilation=d, padding=pad(kernel_size, d))
                                     for d in dilation])
        self.convs2 = nn.ModuleList([nn.Conv1d(channels, channels, kernel_size, 1, dilation=1, padding=pad(kernel_size, 1))
                                     for _ in dilation])


def _pad_act(x, pl, pr, act=hip.ACT_NONE):
    """[B, L, C] -> zero-padded act(x) [B, pl + L + pr, C] in one pass (every convolution of the generator reads LeakyReLU of
    its input, zero "same"-padded: the activation and the halo are made together)."""
    B, L, C = x.shape
    out = torch.empty(B, L + pl + pr, C, dtype=x.dtype, device=x.device)
    hip.check(hip.lib().st5_pad_time_act(x.data_ptr(), out.data_ptr(), B, L, C, pl, pr, act, hip.dt(x), hip.stream()), "st5_pad_time_act")
    return out


def _conv_w(conv, dtype):
    """[Cout, Cin, k] -> [Cout, k*Cin] in the compute dtype (cached)."""
    return Fn._conv_w_fwd(conv.weight, dtype)


def conv_pad(conv):
    k, d = conv.weight.shape[2], conv.dilation[0]
    return (k * d - d) // 2


def conv1d(xp, L, conv, *, alpha=1.0, act=hip.ACT_NONE, residual=None, out=None, beta=0.0, out_pad=0):
    """Channels-last dilated 'same' Conv1d as an implicit GEMM (K segmented per tap: seg = Cin, seg_stride = dilation * Cin) over
    the PADDED input xp [B, L + 2 * conv_pad(conv), Cin]:  y = act(alpha * conv(x) + bias) + residual + beta * out.
    out_pad > 0: y is written into the interior of a fresh [B, L + 2 * out_pad, Cout] buffer with zeroed halo rows -- the layout
    the next convolution reads, no copy in between -- and that buffer is returned."""
    B, Lp, Cin = xp.shape
    Cout, _, k = conv.weight.shape
    d = conv.dilation[0]
    p = conv_pad(conv)
    assert Lp == L + 2 * p
    if out_pad:
        assert out is None and residual is None
        y = torch.empty(B, L + 2 * out_pad, Cout, dtype=xp.dtype, device=xp.device)
        hip.check(hip.lib().st5_zero_halo(y.data_ptr(), B, L, Cout, out_pad, out_pad, hip.dt(y), hip.stream()), "st5_zero_halo")
        c_op = hip.operand(y, Cout, off=out_pad * Cout, rpb=L, bstride=(L + 2 * out_pad) * Cout)
    else:
        y = out if out is not None else torch.empty(B, L, Cout, dtype=xp.dtype, device=xp.device)
        c_op = hip.operand(y, Cout)
    if xp.dtype == torch.bfloat16 and Cout in (32, 64) and Cin in (32, 64, 128):
        # few output channels: the MFMA rows are the output channels (csrc/conv1d_narrow.hip), not a 128-wide GEMM tile
        esz = 2
        hip.check(hip.lib().st5_conv1d_narrow(
            xp.data_ptr(), Lp * Cin, Cin, _conv_w(conv, xp.dtype).data_ptr(), conv.bias.detach().data_ptr(),
            y.data_ptr() + out_pad * Cout * esz, (L + 2 * out_pad) * Cout, Cout,
            hip.ptr(residual), L * Cout, Cout, B, L, Cin, Cout, k, d * Cin, alpha, beta, act, hip.stream()), "st5_conv1d_narrow")
        return y
    if xp.dtype == torch.bfloat16 and Cout == 1 and Cin % 8 == 0 and residual is None and out is None and not out_pad:
        hip.check(hip.lib().st5_conv1d_cout1(xp.data_ptr(), Lp * Cin, Cin, _conv_w(conv, xp.dtype).data_ptr(), conv.bias.detach().data_ptr(),
                                             y.data_ptr(), B, L, Cin, k, d * Cin, alpha, act, hip.stream()), "st5_conv1d_cout1")
        return y
    hip.gemm(hip.operand(xp, Cin, rpb=L, bstride=Lp * Cin, seg=Cin, seg_stride=d * Cin),
             hip.operand(_conv_w(conv, xp.dtype), k * Cin), c_op, B * L, Cout, k * Cin, hip.dt(xp),
             R=hip.operand(residual, Cout) if residual is not None else None, bias=conv.bias.detach(), act=act,
             alpha=alpha, beta=beta)
    return y


def conv_transpose1d(xp, L, convt, *, alpha=1.0):
    """Channels-last ConvTranspose1d with kernel = 2*stride, padding = stride/2 (HiFi-GAN upsampler) over the input padded by one
    row on each side, xp [B, L + 2, Cin] -> [B, s*L, Cout]."""
    B, Lp, Cin = xp.shape
    _, Cout, k = convt.weight.shape
    s = convt.stride[0]
    assert k == 2 * s and convt.padding[0] == s // 2 and Lp == L + 2
    y = torch.empty(B, L * s, Cout, dtype=xp.dtype, device=xp.device)

    def build():
        w = convt.weight.detach()  # [Cin, Cout, k]
        mats = []
        for r in range(s):
            j1 = (r + s // 2) % s
            # y[s*t' + r] = x[t1] W[:,:,j1] + x[t1-1] W[:,:,j1+s]; rows are [x[t1-1] ; x[t1]]
            mats.append(torch.cat([w[:, :, j1 + s], w[:, :, j1]], 0).t().contiguous())  # [Cout, 2*Cin]
        src = torch.stack(mats, 0)
        outw = torch.empty(src.shape, dtype=xp.dtype, device=xp.device)
        Fn._cast_into(src.view(-1, 2 * Cin), outw.view(-1, 2 * Cin))
        return outw
    Wp = Fn.weight_cache.get(("convT", xp.dtype, id(convt.weight)), [convt.weight], build)
    narrow = xp.dtype == torch.bfloat16 and Cout in (32, 64) and Cin in (32, 64, 128)
    for r in range(s):
        t_off = 0 if r < s - s // 2 else 1  # t1 = t' + t_off ; padded row index of x[t1-1] is t1
        if narrow:      # one phase = a 2-tap convolution over adjacent input rows, written to every s-th output row
            hip.check(hip.lib().st5_conv1d_narrow(
                xp.data_ptr() + t_off * Cin * 2, (L + 2) * Cin, Cin, Wp[r].data_ptr(), convt.bias.detach().data_ptr(),
                y.data_ptr() + r * Cout * 2, L * s * Cout, s * Cout, None, 0, 0, B, L, Cin, Cout, 2, Cin, alpha, 0.0, hip.ACT_NONE,
                hip.stream()), "st5_conv1d_narrow")
            continue
        hip.gemm(hip.operand(xp, Cin, off=t_off * Cin, rpb=L, bstride=(L + 2) * Cin), hip.operand(Wp[r], 2 * Cin),
                 hip.operand(y, s * Cout, off=r * Cout, rpb=L, bstride=L * s * Cout), B * L, Cout, 2 * Cin, hip.dt(xp),
                 bias=convt.bias.detach(), alpha=alpha)
    return y


class SpeechT5HifiGan(nn.Module):
    def __init__(self, model_in_dim=80, upsample_initial_channel=512, upsample_rates=(4, 4, 4, 4), upsample_kernel_sizes=(8, 8, 8, 8),
                 resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5)), leaky_relu_slope=0.1,
                 normalize_before=True):
        super().__init__()
        self.num_kernels = len(resblock_kernel_sizes)
        self.num_upsamples = len(upsample_rates)
        self.normalize_before = normalize_before
        self.conv_pre = nn.Conv1d(model_in_dim, upsample_initial_channel, kernel_size=7, stride=1, padding=3)
        self.upsampler = nn.ModuleList()
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            self.upsampler.append(nn.ConvTranspose1d(upsample_initial_channel // (2 ** i), upsample_initial_channel // (2 ** (i + 1)),
                                                     kernel_size=k, stride=u, padding=(k - u) // 2))
        self.resblocks = nn.ModuleList()
        for i in range(len(self.upsampler)):
            channels = upsample_initial_channel // (2 ** (i + 1))
            for k, d in zip(resblock_kernel_sizes, resblock_dilation_sizes):
                self.resblocks.append(HifiGanResidualBlock(channels, k, d, leaky_relu_slope))
        self.conv_post = nn.Conv1d(channels, 1, kernel_size=7, stride=1, padding=3)
        self.register_buffer("mean", torch.zeros(model_in_dim))
        self.register_buffer("scale", torch.ones(model_in_dim))

    @torch.no_grad()
    def forward(self, spectrogram):
        """[B, L, 80] (or [L, 80]) log-mel -> waveform [B, prod(rates)*L] (or 1-D), fp32."""
        is_batched = spectrogram.dim() == 3
        x = spectrogram if is_batched else spectrogram.unsqueeze(0)
        x = Fn.to_compute(x.float().contiguous())
        B, L, C = x.shape
        if self.normalize_before:
            a = (1.0 / self.scale).float().contiguous()
            b = (-self.mean / self.scale).float().contiguous()
            y = torch.empty_like(x)
            hip.check(hip.lib().st5_channel_affine(x.data_ptr(), a.data_ptr(), b.data_ptr(), y.data_ptr(), B * L, C, hip.ACT_NONE,
                                                   hip.dt(x), hip.stream()), "st5_channel_affine")
            x = y
        h = conv1d(_pad_act(x, 3, 3), L, self.conv_pre)
        alpha = 1.0
        LR = hip.ACT_LRELU_01
        for i in range(self.num_upsamples):
            h = conv_transpose1d(_pad_act(h, 1, 1, LR), L, self.upsampler[i], alpha=alpha)
            L = h.shape[1]
            acc = None
            for j in range(self.num_kernels):
                blk = self.resblocks[i * self.num_kernels + j]
                r = h
                n = len(blk.convs1)
                for q, (c1, c2) in enumerate(zip(blk.convs1, blk.convs2)):
                    # r -> [LeakyReLU + halo, one pass] -> c1 (+ LeakyReLU in its epilogue, written straight into c2's padded
                    # input) -> c2 (+ r in its epilogue): 7 passes over the activation per pair (round 2: 11)
                    t = conv1d(_pad_act(r, conv_pad(c1), conv_pad(c1), LR), L, c1, act=LR, out_pad=conv_pad(c2))
                    if q == n - 1:   # last pair: accumulate the block output into the running sum of the resblocks
                        if acc is None:
                            acc = conv1d(t, L, c2, residual=r)
                        else:
                            conv1d(t, L, c2, residual=r, out=acc, beta=1.0)
                    else:
                        r = conv1d(t, L, c2, residual=r)
            h = acc
            alpha = 1.0 / self.num_kernels  # mean over resblocks, applied by the next convolution
        w = conv1d(_pad_act(h, 3, 3, hip.ACT_LRELU_001), L, self.conv_post, alpha=alpha, act=hip.ACT_TANH)  # [B, L_out, 1]
        wav = Fn.to_float(w).view(B, -1)
        return wav if is_batched else wav.view(-1)

"""Shared helpers of the module mirrors: row layout conversion, fp32 LayerNorm module, positional tables."""
import math

import torch
import torch.nn as nn

from .. import functional as Fn


class LayerNorm(nn.Module):
    """fairseq LayerNorm(normalized_shape, eps) with the HIP kernel behind it; fp32 affine parameters."""

    def __init__(self, dim, eps=1e-5, export=False):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.eps = eps
        self.normalized_shape = (dim,)

    def forward(self, x, gate=None, q8=False, relay=None):
        return Fn.layer_norm(x, self.weight, self.bias, self.eps, gate=gate, q8=q8, relay=relay)


def tbc_to_rows(x):
    """[T,B,C] -> contiguous batch-major [B,T,C] (zero-copy when x is a transposed view of one)."""
    return x.transpose(0, 1).contiguous()


def rows_to_tbc(x_btc):
    return x_btc.transpose(0, 1)


def fairseq_sinusoidal_table(num_embeddings, dim, padding_idx, device):
    """fairseq SinusoidalPositionalEmbedding.get_embedding: [sin | cos] halves, zero pad row."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    ang = torch.arange(num_embeddings, dtype=torch.float).unsqueeze(1) * freq.unsqueeze(0)
    emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    if dim % 2 == 1:
        emb = torch.cat([emb, torch.zeros(num_embeddings, 1)], dim=1)
    if padding_idx is not None:
        emb[padding_idx, :] = 0
    return emb.to(device)


def espnet_pe_table(length, dim, device):
    """espnet PositionalEncoding table: interleaved sin/cos."""
    pe = torch.zeros(length, dim)
    pos = torch.arange(0, length, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(10000.0) / dim))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.to(device)


class SinusoidalPositionalEmbedding(nn.Module):
    """fairseq PositionalEmbedding(learned=False): constant table, positions = cumsum(non-pad)+pad.
    Keeps the `_float_tensor` buffer so that reference checkpoints load."""

    def __init__(self, embedding_dim, padding_idx, init_size=1024):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.padding_idx = padding_idx if padding_idx is not None else 0
        self.register_buffer("_float_tensor", torch.FloatTensor(1))
        self._table = None

    def table(self, n, device):
        if self._table is None or self._table.shape[0] < n or self._table.device != device:
            self._table = fairseq_sinusoidal_table(max(n, 64), self.embedding_dim, self.padding_idx, device)
        return self._table

    def positions(self, non_pad):
        m = non_pad.int()
        return (torch.cumsum(m, dim=1) * m).long() + self.padding_idx


class ScaledPositionalEncoding(nn.Module):
    """espnet ScaledPositionalEncoding: x + alpha * pe, then dropout."""

    def __init__(self, d_model, dropout_rate, max_len=5000):
        super().__init__()
        self.d_model = d_model
        self.alpha = nn.Parameter(torch.tensor(1.0))
        self.dropout_rate = dropout_rate
        self._pe = None

    def pe(self, n, device):
        if self._pe is None or self._pe.shape[0] < n or self._pe.device != device:
            self._pe = espnet_pe_table(max(n, 64), self.d_model, device)
        return self._pe


class ScaledPEAdd(torch.autograd.Function):
    """y = x + alpha * pe[:T] (pe constant fp32 table, alpha fp32 scalar parameter)."""

    @staticmethod
    def forward(ctx, x, alpha, pe):
        B, T, d = x.shape
        idx = torch.arange(T, device=x.device, dtype=torch.int32).repeat(B)
        y = Fn.add_table_rows_dev(x, pe, idx, alpha.detach())  # alpha stays on the device (no host sync)
        ctx.save_for_backward(pe)
        ctx.alpha = alpha
        ctx.T = T
        return y

    @staticmethod
    def backward(ctx, dy):
        (pe,) = ctx.saved_tensors
        dalpha = None
        if ctx.alpha.requires_grad:
            # d alpha = sum(dy * pe): tiny reduction (glue)
            dalpha = (dy.float().sum(0) * pe[:ctx.T]).sum()
        return dy, dalpha, None

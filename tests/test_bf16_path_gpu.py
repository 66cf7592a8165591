"""Parity of the kernels the BENCHED bf16 step actually runs: the 256x256 NT tile, the multi-tile
LDS-DMA 128x128 NT kernel at the model's shapes with every epilogue class, the fused attention backward against torch
autograd directly, and the fused Adam step against an fp64 restatement of fairseq's adam.

References are plain torch fp32 math on the same (dtype-rounded) inputs; the big GEMM references run through torch.matmul on
the GPU in fp32 (a CPU matmul of 10^11 FLOP would dominate the suite).  bf16 tolerance: 2e-2 of the output scale (8 mantissa
bits in, fp32 accumulation), fp32: 3e-5 (MFMA fp32 chains vs rocBLAS summation order at K up to 3072)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from speecht5_amd import functional as Fn, hip  # noqa: E402

DTYPES = [torch.bfloat16, torch.float32]


def _tol(dtype):
    return 2e-2 if dtype == torch.bfloat16 else 3e-5


def _close(got, ref, dtype, what):
    got, ref = got.float(), ref.float()
    s = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    assert math.isfinite(err) and err <= _tol(dtype) * max(s, 1e-6), f"{what}: max err {err:.3e} vs scale {s:.3e} ({dtype})"


def _drop_scale(M, N, p, seed, cuda):
    """0 / (1/keep) factors of the library's counter RNG for element counters row*N + col (st5_dropout on ones)."""
    ones = torch.ones(M, N, dtype=torch.float32, device=cuda)
    out = torch.empty_like(ones)
    hip.check(hip.lib().st5_dropout(ones.data_ptr(), out.data_ptr(), M * N, p, seed, hip.F32, hip.stream()), "st5_dropout")
    return out


def _run_nt(cuda, dtype, M, N, K, kind, seed=0):
    """One NT GEMM of epilogue class `kind` + its torch reference (fp32 on the rounded inputs)."""
    g = torch.Generator(device="cpu").manual_seed(1000 * seed + M + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    A, W = a.to(dtype).to(cuda), w.to(dtype).to(cuda)
    Af, Wf = A.float(), W.float()
    bias = torch.randn(N, generator=g).to(cuda)
    C = torch.full((M, N), float("nan"), dtype=dtype, device=cuda)
    kw, extra = {}, {}
    base = Af @ Wf.t()
    if kind == "plain":
        ref = base
    elif kind == "fc1":            # bias + GELU + pre-activation second output (+ activation dropout)
        Cp = torch.full((M, N), float("nan"), dtype=dtype, device=cuda)
        kw = dict(bias=bias, act=hip.ACT_GELU, Cpre=hip.operand(Cp, N), dropout_p=0.1, seed=4242)
        pre = base + bias
        ref = F.gelu(pre) * _drop_scale(M, N, 0.1, 4242, cuda)
        extra = dict(pre=(Cp, pre))
    elif kind == "fc2":            # bias + dropout + residual
        r = torch.randn(M, N, generator=g).to(dtype).to(cuda)
        kw = dict(bias=bias, R=hip.operand(r, N), dropout_p=0.1, seed=77)
        ref = (base + bias) * _drop_scale(M, N, 0.1, 77, cuda) + r.float()
    elif kind == "dact":           # data gradient with act'(P) * dropout mask, scaled
        P = torch.randn(M, N, generator=g).to(dtype).to(cuda)
        kw = dict(P=hip.operand(P, N), act=hip.ACT_GELU, flags=hip.DACT, dropout_p=0.1, seed=99, alpha=0.5)
        x = P.float()
        dgelu = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
        ref = 0.5 * base * dgelu * _drop_scale(M, N, 0.1, 99, cuda)
    elif kind == "relay":          # data gradient + relayed residual gradient
        r = torch.randn(M, N, generator=g).to(dtype).to(cuda)
        kw = dict(R=hip.operand(r, N))
        ref = base + r.float()
    elif kind == "beta":           # accumulate into the existing output
        c0 = torch.randn(M, N, generator=g).to(dtype).to(cuda)
        C = c0.clone()
        kw = dict(beta=1.0, act=hip.ACT_RELU, bias=bias)
        ref = torch.relu(base + bias) + c0.float()
    else:
        raise AssertionError(kind)
    hip.gemm(hip.operand(A, K), hip.operand(W, K), hip.operand(C, N), M, N, K, hip.dt(dtype), **kw)
    torch.cuda.synchronize()
    return C, ref, extra


EPI = ["plain", "fc1", "fc2", "dact", "relay", "beta"]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", EPI)
def test_nt256_tile_forced(cuda, dtype, kind):
    """gemm_nt256_kernel through st5_gemm_set_nt_tile(2): M and N tails (520 = 2*256 + 8, 776 = 3*256 + 8), every epilogue."""
    L = hip.lib()
    hip.check(L.st5_gemm_set_nt_tile(2), "set_nt_tile")
    try:
        C, ref, extra = _run_nt(cuda, dtype, 520, 776, 768, kind, seed=1)
    finally:
        hip.check(L.st5_gemm_set_nt_tile(0), "set_nt_tile")
    _close(C, ref, dtype, f"nt256 forced / {kind}")
    for k, (got, r) in extra.items():
        _close(got, r, dtype, f"nt256 forced / {kind} / {k}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_nt256_tile_natural_and_bit_equal_to_128(cuda, dtype):
    """A problem the dispatcher itself sends to the 256^2 kernel (>= 448 tiles: the conv feature-extractor shape class,
    M = 57 608 rows incl. a tail, N = 512, K = 1536) with the conv epilogue (GELU + pre-activation output); result equals the
    128^2 kernel's bit for bit (same MFMA chains per output element) and torch within tolerance."""
    M, N, K = 57608, 512, 1536
    L = hip.lib()
    outs = []
    for mode in (0, 1):
        hip.check(L.st5_gemm_set_nt_tile(mode), "set_nt_tile")
        try:
            g = torch.Generator().manual_seed(5)
            a = torch.randn(M, K, generator=g).to(dtype).to(cuda)
            w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype).to(cuda)
            C = torch.full((M, N), float("nan"), dtype=dtype, device=cuda)
            Cp = torch.full((M, N), float("nan"), dtype=dtype, device=cuda)
            hip.gemm(hip.operand(a, K), hip.operand(w, K), hip.operand(C, N), M, N, K, hip.dt(dtype), Cpre=hip.operand(Cp, N),
                     act=hip.ACT_GELU)
            torch.cuda.synchronize()
            outs.append((C, Cp))
        finally:
            hip.check(L.st5_gemm_set_nt_tile(0), "set_nt_tile")
    pre = a.float() @ w.float().t()
    _close(outs[0][1], pre, dtype, "nt256 natural / pre-activation")
    _close(outs[0][0], F.gelu(pre), dtype, "nt256 natural / gelu")
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][0], outs[1][0]), "256^2 and 128^2 tiles differ bitwise"


@pytest.mark.parametrize("kind", EPI)
@pytest.mark.parametrize("M,N,K", [(3992, 3072, 768), (8192, 768, 3072), (3992, 2304, 768), (2504, 768, 768)])
def test_nt_glds_multi_tile_model_shapes(cuda, M, N, K, kind):
    """gemm_nt_glds_kernel<bf16> on the transformer GEMM shapes of cfg 2 (XCD-remapped multi-tile grids, M tails 3992 =
    31*128 + 24 and 2504 = 19*128 + 72) for every epilogue class of the step."""
    C, ref, extra = _run_nt(cuda, torch.bfloat16, M, N, K, kind, seed=2)
    _close(C, ref, torch.bfloat16, f"nt glds {M}x{N}x{K} / {kind}")
    for k, (got, r) in extra.items():
        _close(got, r, torch.bfloat16, f"nt glds {M}x{N}x{K} / {kind} / {k}")


def test_tn_glds_model_shapes_with_bias_column(cuda):
    """Weight-gradient form at the model's shapes (split-K slabs + batched reduction off): dW = dY^T X, db = dY^T 1."""
    for (M, N, K) in [(3072, 768, 3992), (768, 3072, 8192), (2304, 768, 3992)]:
        g = torch.Generator().manual_seed(M + K)
        dy = torch.randn(K, M, generator=g).to(torch.bfloat16).to(cuda)
        x = (torch.randn(K, N, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(cuda)
        C = torch.full((M, N), 1.0, dtype=torch.float32, device=cuda)
        db = torch.full((M,), 2.0, dtype=torch.float32, device=cuda)
        hip.gemm(hip.operand(dy, M), hip.operand(x, N), hip.operand(C, N), M, N, K, hip.BF16,
                 flags=hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32, beta=1.0, asum=db)
        torch.cuda.synchronize()
        ref = dy.float().t() @ x.float()
        _close(C - 1.0, ref, torch.bfloat16, f"tn glds {M}x{N}x{K}")
        rb = dy.float().sum(0)
        assert (db - 2.0 - rb).abs().max().item() <= 2e-2 * max(rb.abs().max().item(), math.sqrt(K))


@pytest.mark.parametrize("K", [3992, 2504, 8192])
def test_weight_gradient_group_launch(cuda, K):
    """st5_gemm_tn_group: the four weight gradients of an encoder layer (QKV 2304 x 768, out 768 x 768, fc1 3072 x 768, fc2 768 x 3072;
    reduction over K tokens incl. a partial k-tile) as ONE launch without split-K, accumulating into non-zero gradient buffers with
    the bias-gradient columns -- against fp32 matmul, and against the same problems through st5_gemm one by one (split-K: another
    summation order, so a tolerance), and a problem launched alone against its result inside the group (bit-equal: replayed and eager
    updates, or two micro-batch schedules, queue different groups and still have to agree)."""
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
    fl = hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32
    g = torch.Generator().manual_seed(K)
    probs, keep = [], []
    for (M, N) in shapes:
        dy = torch.randn(K, M, generator=g).to(torch.bfloat16).to(cuda)
        x = (torch.randn(K, N, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(cuda)
        C = torch.full((M, N), 1.0, dtype=torch.float32, device=cuda)
        db = torch.full((M,), 2.0, dtype=torch.float32, device=cuda)
        keep.append((dy, x, C, db))
        probs.append((hip.operand(dy, M), hip.operand(x, N), hip.operand(C, N), M, N, K, fl, 1.0, db))
    hip.gemm_tn_group(probs, hip.BF16)
    torch.cuda.synchronize()
    for (dy, x, C, db), (M, N) in zip(keep, shapes):
        ref = dy.float().t() @ x.float()
        _close(C - 1.0, ref, torch.bfloat16, f"tn group {M}x{N}x{K}")
        rb = dy.float().sum(0)
        assert (db - 2.0 - rb).abs().max().item() <= 2e-2 * max(rb.abs().max().item(), math.sqrt(K))
        C1 = torch.full((M, N), 1.0, dtype=torch.float32, device=cuda)
        d1 = torch.full((M,), 2.0, dtype=torch.float32, device=cuda)
        hip.gemm(hip.operand(dy, M), hip.operand(x, N), hip.operand(C1, N), M, N, K, hip.BF16, flags=fl, beta=1.0, asum=d1)
        torch.cuda.synchronize()
        assert (C1 - C).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item())
    # a problem's result does not depend on its group: the out-projection gradient alone == the one computed inside the group of four
    dy, x, Cg, dbg = keep[1]
    M, N = shapes[1]
    C1 = torch.full((M, N), 1.0, dtype=torch.float32, device=cuda)
    d1 = torch.full((M,), 2.0, dtype=torch.float32, device=cuda)
    hip.gemm_tn_group([(hip.operand(dy, M), hip.operand(x, N), hip.operand(C1, N), M, N, K, fl, 1.0, d1)], hip.BF16)
    torch.cuda.synchronize()
    assert torch.equal(C1, Cg) and torch.equal(d1, dbg)


@pytest.mark.parametrize("K", [3992, 8192, 640])
def test_phased_weight_gradient_group_equals_the_128_tile_group_bit_for_bit(cuda, K):
    """Round 6 (gemm_tn8p_group_kernel): the grouped launch on the phased 256 x 256 schedule -- the default for problems whose M, N are
    multiples of 256 -- gives every WEIGHT-gradient element the MFMA chain of the 128 x 128 group: equal bits (a partial last k-tile
    included); the bias-gradient column (summed on the VALU in its own fixed order) within rounding of it; two layers' problems in one
    launch; a problem alone == inside the group."""
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072), (1024, 1024), (256, 512)]
    fl = hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32
    L = hip.lib()
    g = torch.Generator().manual_seed(K + 1)
    ops = []
    for (M, N) in shapes:
        dy = torch.randn(K, M, generator=g).to(torch.bfloat16).to(cuda)
        x = (torch.randn(K, N, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(cuda)
        ops.append((dy, x))
    assert all(L.st5_gemm_tn_group_is_phased(M, N, K) == 1 for (M, N) in shapes)

    def run(mode, which):
        hip.check(L.st5_gemm_set_tn_group_tile(mode), "st5_gemm_set_tn_group_tile")
        try:
            outs, probs = [], []
            for i in which:
                (M, N), (dy, x) = shapes[i], ops[i]
                C = torch.full((M, N), 1.0, dtype=torch.float32, device=cuda)
                db = torch.full((M,), 2.0, dtype=torch.float32, device=cuda)
                outs.append((C, db))
                probs.append((hip.operand(dy, M), hip.operand(x, N), hip.operand(C, N), M, N, K, fl, 1.0, db))
            hip.gemm_tn_group(probs, hip.BF16)
            torch.cuda.synchronize()
            return outs
        finally:
            hip.check(L.st5_gemm_set_tn_group_tile(0), "st5_gemm_set_tn_group_tile")

    small, big = run(1, range(6)), run(0, range(6))
    for i, ((C0, d0), (C1, d1)) in enumerate(zip(small, big)):
        dy, x = ops[i]
        _close(C1 - 1.0, dy.float().t() @ x.float(), torch.bfloat16, f"phased tn group {shapes[i]} K={K}")
        assert torch.equal(C0, C1), f"{shapes[i]}: {int((C0 != C1).sum())} weight-gradient elements differ from the 128^2 group"
        rb = dy.float().sum(0)
        assert (d1 - 2.0 - rb).abs().max().item() <= 2e-2 * max(rb.abs().max().item(), math.sqrt(K))
        assert (d1 - d0).abs().max().item() <= 1e-3 * max(rb.abs().max().item(), math.sqrt(K))
    alone = run(0, [1])[0]
    assert torch.equal(alone[0], big[1][0]) and torch.equal(alone[1], big[1][1])
    # twelve problems in ONE phased launch (compact records: up to sixteen per launch; a Base decoder layer has six)
    which = [0, 1, 2, 3, 4, 5, 1, 4, 5, 1, 4, 5]
    for i, (C, d) in zip(which, run(0, which)):
        assert torch.equal(C, big[i][0]) and torch.equal(d, big[i][1]), f"problem {shapes[i]} inside a group of twelve"


@pytest.mark.parametrize("mode", [3, 4])
@pytest.mark.parametrize("kind", EPI)
def test_nt_phased_256_tile_forced(cuda, mode, kind):
    """gemm_nt8p_kernel (phased 256^2 schedule; 3 = staggered wave halves, 4 = not) through st5_gemm_set_nt_tile: M / N tails, k-tile
    counts 1, 2 (prologue-only paths) and 12, every epilogue; bit-equal to the 128^2 kernel (same MFMA chain per output element)."""
    L = hip.lib()
    for (M, N, K) in [(520, 776, 768), (264, 256, 64), (256, 520, 128)]:
        hip.check(L.st5_gemm_set_nt_tile(mode), "set_nt_tile")
        try:
            C, ref, extra = _run_nt(cuda, torch.bfloat16, M, N, K, kind, seed=3)
        finally:
            hip.check(L.st5_gemm_set_nt_tile(0), "set_nt_tile")
        _close(C, ref, torch.bfloat16, f"nt8p mode {mode} {M}x{N}x{K} / {kind}")
        for k, (got, r) in extra.items():
            _close(got, r, torch.bfloat16, f"nt8p mode {mode} / {kind} / {k}")
        C0, _, extra0 = _run_nt(cuda, torch.bfloat16, M, N, K, kind, seed=3)
        assert torch.equal(C, C0), f"phased 256^2 (mode {mode}) and 128^2 tiles differ bitwise at {M}x{N}x{K} / {kind}"


@pytest.mark.parametrize("kind", EPI)
def test_nt_half_height_tile_bit_equal_to_128(cuda, kind):
    """gemm_nt_m64_kernel (64 x 128 tiles, three blocks per CU; round 6) forced through st5_gemm_set_nt_tile(5) against the 128^2 kernel
    (mode 1): M tails inside a 64-row tile (3992 = 62 * 64 + 24, 520, 72), an N tail (776), k-tile counts 1, 2, 3, 12, 48, every epilogue:
    torch within tolerance and bit-equal (same MFMA chain per output element); and the dispatcher's own choice (mode 0) equals both."""
    L = hip.lib()
    for (M, N, K) in [(3992, 768, 768), (520, 776, 128), (72, 128, 64), (2504, 768, 192), (1024, 768, 3072)]:
        outs = []
        for mode in (5, 1, 0):
            hip.check(L.st5_gemm_set_nt_tile(mode), "set_nt_tile")
            try:
                outs.append(_run_nt(cuda, torch.bfloat16, M, N, K, kind, seed=6))
            finally:
                hip.check(L.st5_gemm_set_nt_tile(0), "set_nt_tile")
        (C5, ref, extra5), (C1, _, extra1), (C0, _, extra0) = outs
        _close(C5, ref, torch.bfloat16, f"64x128 tiles {M}x{N}x{K} / {kind}")
        assert torch.equal(C5, C1), f"64x128 and 128^2 tiles differ bitwise at {M}x{N}x{K} / {kind}"
        assert torch.equal(C0, C1), f"the dispatcher's choice differs bitwise from 128^2 at {M}x{N}x{K} / {kind}"
        for k in extra5:
            _close(extra5[k][0], extra5[k][1], torch.bfloat16, f"64x128 tiles / {kind} / {k}")
            assert torch.equal(extra5[k][0], extra1[k][0]) and torch.equal(extra0[k][0], extra1[k][0])


@pytest.mark.parametrize("kind", EPI)
def test_nt_five_slot_ring_bit_equal_to_two_stages(cuda, kind):
    """gemm_nt_glds_kernel<.., 5, ..> (five 16 KB operand slots, the default for grids of more than one block per CU) against the
    two-stage ring (st5_gemm_set_nt_slots(4)): same MFMA chain per element -> bit-equal; k-tile counts 2, 3, 5, 12 walk the slot ring
    through its wrap-arounds, M tail 3992."""
    L = hip.lib()
    for (M, N, K) in [(3992, 3072, 768), (2304, 1024, 128), (2304, 768, 192), (4096, 768, 320)]:
        C5, ref, extra5 = _run_nt(cuda, torch.bfloat16, M, N, K, kind, seed=4)
        hip.check(L.st5_gemm_set_nt_slots(4), "set_nt_slots")
        try:
            C4, _, extra4 = _run_nt(cuda, torch.bfloat16, M, N, K, kind, seed=4)
        finally:
            hip.check(L.st5_gemm_set_nt_slots(5), "set_nt_slots")
        _close(C5, ref, torch.bfloat16, f"five-slot ring {M}x{N}x{K} / {kind}")
        assert torch.equal(C5, C4), f"five-slot ring differs from two stages at {M}x{N}x{K} / {kind}"
        for k in extra5:
            assert torch.equal(extra5[k][0], extra4[k][0])


@pytest.mark.parametrize("mode", [1, 2])
def test_tn_phased_256_tile(cuda, mode):
    """gemm_tn8p_kernel through st5_gemm_set_tn_phased (off by default): weight-gradient shapes with a K tail (3992 = 62 * 64 + 24),
    accumulate-into-C (beta = 1 after the slab reduction) and the VALU bias-gradient column."""
    L = hip.lib()
    for (M, N, K) in [(768, 3072, 3992), (3072, 768, 8192), (256, 256, 512), (512, 1536, 20000)]:
        g = torch.Generator().manual_seed(M + K)
        dy = torch.randn(K, M, generator=g).to(torch.bfloat16).to(cuda)
        x = (torch.randn(K, N, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(cuda)
        outs = []
        for m in (mode, 0):
            C = torch.full((M, N), 1.0, dtype=torch.float32, device=cuda)
            db = torch.full((M,), 2.0, dtype=torch.float32, device=cuda)
            hip.check(L.st5_gemm_set_tn_phased(m), "set_tn_phased")
            try:
                hip.gemm(hip.operand(dy, M), hip.operand(x, N), hip.operand(C, N), M, N, K, hip.BF16,
                         flags=hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32, beta=1.0, asum=db)
                torch.cuda.synchronize()
            finally:
                hip.check(L.st5_gemm_set_tn_phased(0), "set_tn_phased")
            outs.append((C, db))
        ref = dy.float().t() @ x.float()
        rb = dy.float().sum(0)
        for tag, (C, db) in zip((f"tn8p mode {mode}", "tn 128^2"), outs):
            _close(C - 1.0, ref, torch.bfloat16, f"{tag} {M}x{N}x{K}")
            assert (db - 2.0 - rb).abs().max().item() <= 2e-2 * max(rb.abs().max().item(), math.sqrt(K)), tag
        # the two kernels split K differently, so they agree to fp32 summation order, not bitwise
        assert (outs[0][0] - outs[1][0]).abs().max().item() <= 1e-3 * ref.abs().max().item()


# ----------------------------------------------------------------------------------------------------------------------
# fused attention backward vs torch autograd (no dropout: torch cannot reproduce the counter RNG; the dropout masks of the
# fused kernels are pinned against the unfused path, which test_ops_gpu.py pins against torch, in test_flash_gpu.py)
# ----------------------------------------------------------------------------------------------------------------------
def _torch_attention(q, k, v, pe, kpm, causal, maxrel):
    """fp32 reference on [B,H,T,hd] leaf tensors (multihead_attention.py:340-389 incl. the Shaw bias of encoder.py:52-59)."""
    B, H, T, hd = q.shape
    S = k.shape[2]
    s = torch.einsum("bhid,bhjd->bhij", q, k) * hd ** -0.5
    if pe is not None:
        i = torch.arange(T, device=q.device)[:, None]
        j = torch.arange(S, device=q.device)[None, :]
        idx = (i - j).clamp(-maxrel, maxrel - 1) + maxrel
        s = s + torch.einsum("bhid,ijd->bhij", q * hd ** -0.5, pe[idx])
    if causal:
        s = s + torch.triu(torch.full((T, S), float("-inf"), device=q.device), 1 + (S - T))
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :].bool(), float("-inf"))
    return torch.einsum("bhij,bhjd->bhid", torch.softmax(s, -1), v)


@pytest.mark.parametrize("T,S,causal,rel,pad", [(499, 499, 0, 1, 0), (512, 512, 0, 1, 1), (313, 313, 1, 0, 1), (313, 499, 0, 0, 1),
                                                 (70, 70, 0, 1, 1), (33, 200, 0, 0, 0)])
def test_flash_backward_matches_torch_autograd(cuda, T, S, causal, rel, pad):
    torch.manual_seed(T * 5 + S)
    B, H, hd = 2, 3, 64
    maxrel = 160 if T > 200 else 16
    d = H * hd
    dt = torch.bfloat16
    Fn.set_compute_dtype(dt)
    try:
        kpm = torch.zeros(B, S, dtype=torch.uint8)
        if pad:
            kpm[1, S - 11:] = 1
        KP = kpm.to(cuda) if pad else None
        pe = (torch.randn(2 * maxrel, hd) * 0.7).to(dt).to(cuda).requires_grad_(True) if rel else None
        dout = torch.randn(B * T, d).to(dt).to(cuda)
        if T == S:
            x = (torch.randn(B * T, 3 * d) * 1.2).to(dt).to(cuda).requires_grad_(True)
            out = Fn.SelfAttentionFunction.apply(x, pe, KP, (B, H, T, hd, maxrel if rel else 0, bool(causal), 0.0))
            out.backward(dout)
            x32 = x.detach().float().view(B, T, 3, H, hd)
            q, k, v = (x32[:, :, i].permute(0, 2, 1, 3).clone().requires_grad_(True) for i in range(3))
        else:
            xq = (torch.randn(B * T, d) * 1.2).to(dt).to(cuda).requires_grad_(True)
            xkv = (torch.randn(B * S, 2 * d) * 1.2).to(dt).to(cuda).requires_grad_(True)
            out, _ = Fn.CrossAttentionFunction.apply(xq, xkv, KP, (B, H, T, S, hd, 0.0, False))
            out.backward(dout)
            q = xq.detach().float().view(B, T, H, hd).permute(0, 2, 1, 3).clone().requires_grad_(True)
            kv32 = xkv.detach().float().view(B, S, 2, H, hd)
            k, v = (kv32[:, :, i].permute(0, 2, 1, 3).clone().requires_grad_(True) for i in range(2))
        pe32 = pe.detach().float().clone().requires_grad_(True) if rel else None
        ref = _torch_attention(q, k, v, pe32, KP, causal, maxrel)
        ref.backward(dout.float().view(B, T, H, hd).permute(0, 2, 1, 3))
        torch.cuda.synchronize()
        ro = ref.detach().permute(0, 2, 1, 3).reshape(B * T, d)
        _close(out.detach(), ro, dt, "flash forward")
        if T == S:
            gref = torch.stack([t.grad.permute(0, 2, 1, 3) for t in (q, k, v)], 2).reshape(B * T, 3 * d)
            for i, nm in enumerate(("dq", "dk", "dv")):
                _close(x.grad.view(B * T, 3, d)[:, i], gref.view(B * T, 3, d)[:, i], dt, f"flash backward {nm}")
        else:
            _close(xq.grad, q.grad.permute(0, 2, 1, 3).reshape(B * T, d), dt, "flash backward dq (cross)")
            gkv = torch.stack([t.grad.permute(0, 2, 1, 3) for t in (k, v)], 2).reshape(B * S, 2 * d)
            _close(xkv.grad, gkv, dt, "flash backward dkv (cross)")
        if rel:
            _close(pe.grad, pe32.grad, dt, "flash backward dpe")
    finally:
        Fn.set_compute_dtype(torch.float32)


# ----------------------------------------------------------------------------------------------------------------------
# fused Adam vs an fp64 restatement of fairseq/optim/adam.py + fairseq's clip_grad_norm_ (un-vendored third party:
# algorithm restated from its published source; SURVEY.md App. A "trainer/DDP gradient semantics")
# ----------------------------------------------------------------------------------------------------------------------
def _adam_ref(p, g, m, v, lr, b1, b2, eps, wd, step, max_norm, gscale):
    g = g.double() * gscale
    if max_norm > 0:
        norm = g.norm()
        g = g * min(1.0, max_norm / (float(norm) + 1e-6))
    m = b1 * m.double() + (1 - b1) * g
    v = b2 * v.double() + (1 - b2) * g * g
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    step_size = lr * math.sqrt(bc2) / bc1
    p = p.double() * (1 - lr * wd) - step_size * m / (v.sqrt() + eps)
    return p, m, v


@pytest.mark.parametrize("clip,wd,step", [(0.0, 0.0, 1), (5.0, 0.01, 1), (0.05, 0.01, 7), (25.0, 0.1, 1000)])
def test_adam_step_matches_fp64_restatement(cuda, clip, wd, step):
    torch.manual_seed(step)
    n = 100003                                  # not a multiple of 4: vector body + scalar tail
    p = torch.randn(n, device=cuda)
    g = torch.randn(n, device=cuda) * torch.logspace(-6, 0, n, device=cuda)   # tiny-gradient entries exercise eps
    m = torch.randn(n, device=cuda) * 0.1
    v = torch.rand(n, device=cuda) * 1e-3
    lr, b1, b2, eps, gscale = 2e-4, 0.9, 0.98, 1e-6, 0.5
    rp, rm, rv = _adam_ref(p, g, m, v, lr, b1, b2, eps, wd, step, clip, gscale)
    gn = torch.zeros(1, device=cuda)
    L = hip.lib()
    hip.check(L.st5_sumsq(g.data_ptr(), gn.data_ptr(), n, 1.0, 0, hip.F32, hip.stream()), "st5_sumsq")
    assert abs(float(gn) - float(g.double().pow(2).sum())) <= 1e-5 * float(g.double().pow(2).sum())
    mirror = torch.zeros(n, dtype=torch.bfloat16, device=cuda)
    hip.check(L.st5_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, b1, b2, eps, wd, step,
                              gn.data_ptr() if clip > 0 else 0, clip, gscale, mirror.data_ptr(), hip.stream()), "st5_adam_step")
    torch.cuda.synchronize()
    assert (p.double() - rp).abs().max().item() <= 2e-6 * rp.abs().max().item() + 1e-9, "parameters"
    assert (m.double() - rm).abs().max().item() <= 2e-6 * rm.abs().max().item(), "first moment"
    assert (v.double() - rv).abs().max().item() <= 2e-6 * rv.abs().max().item() + 1e-12, "second moment"
    assert torch.equal(mirror, p.to(torch.bfloat16)), "bf16 parameter image"

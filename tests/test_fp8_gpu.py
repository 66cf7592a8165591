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


def test_quantiser_propagates_non_finite_inputs(cuda):
    """a NaN or Inf must reach the GEMM output (and with it the loss / the gradient norm) instead of being clamped to
    +-448: the element becomes the e4m3 NaN code 0x7f, its block's scale the e8m0 NaN 0xff, the OTHER elements of the block keep the
    codes the finite block maximum gives them, and other blocks are untouched."""
    torch.manual_seed(3)
    x = torch.randn(16, 256, device=cuda).to(torch.bfloat16)
    clean_q, clean_s = hip.quant_mxfp8(x)
    bad = x.clone()
    bad[2, 7] = float("nan")
    bad[5, 40] = float("inf")
    bad[5, 41] = float("-inf")
    q, s = hip.quant_mxfp8(bad)
    torch.cuda.synchronize()
    assert int(s[2, 0]) == 0xff and int(s[5, 1]) == 0xff
    assert int(q[2, 7]) == 0x7f and int(q[5, 40]) == 0x7f and int(q[5, 41]) == 0x7f
    keep = torch.ones_like(s, dtype=torch.bool); keep[2, 0] = False; keep[5, 1] = False
    assert torch.equal(s[keep], clean_s[keep])
    other = torch.ones_like(q, dtype=torch.bool); other[2, :32] = False; other[5, 32:64] = False
    assert torch.equal(q[other], clean_q[other])
    # the finite neighbours of the Inf are NOT flushed to zero by an Inf-sized scale
    assert int((q[5, 32:64] & 0x7f).ne(0).sum()) >= 28
    # and the GEMM carries it to the output rows that consume the block
    W = (torch.randn(64, 256, device=cuda) * 0.05).to(torch.bfloat16)
    Wq, Ws = hip.quant_mxfp8(W)
    C = torch.zeros(16, 64, dtype=torch.bfloat16, device=cuda)
    hip.gemm_mxfp8(q, s, Wq, Ws, hip.operand(C, 64), 16, 64, 256)
    torch.cuda.synchronize()
    fin = torch.isfinite(C.float()).all(dim=1)
    assert not bool(fin[2]) and not bool(fin[5]) and bool(fin[[0, 1, 3, 4] + list(range(6, 16))].all())


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


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 256), (300, 520, 384), (1000, 1024, 1024), (257, 264, 640), (4096, 1024, 4096)])
def test_phased_256_tile_mx_kernel_is_bit_identical_to_the_128_tile_kernel(cuda, M, N, K):
    """Round 6: gemm_nt8p_mx8_kernel (st5_gemm_set_mx8_tile(2): phased 256 x 256 schedule on fp8 bytes, scales through buffer loads one
    k-tile ahead, last two k-tiles peeled) runs every output element through the MFMA chain of the 128 x 128 kernel: equal bits, for
    1 / 2 / 3 / many k-tiles, M / N tails inside a tile, and every fused epilogue class of the training step."""
    torch.manual_seed(M * 7 + N)
    L = hip.lib()
    A = torch.randn(M, K, device=cuda).to(torch.bfloat16)
    B = (torch.randn(N, K, device=cuda) * 0.05).to(torch.bfloat16)
    A[5] *= 300.0; B[3] *= 40.0; A[M - 1] *= 1e-3                         # rows with very different scale bytes
    Aq, As = hip.quant_mxfp8(A)
    Bq, Bs = hip.quant_mxfp8(B)
    bias = torch.randn(N, device=cuda)
    R = torch.randn(M, N, device=cuda).to(torch.bfloat16)
    P = torch.randn(M, N, device=cuda).to(torch.bfloat16)

    def run(mode):
        hip.check(L.st5_gemm_set_mx8_tile(mode), "st5_gemm_set_mx8_tile")
        outs = []
        try:
            C = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=cuda)
            hip.gemm_mxfp8(Aq, As, Bq, Bs, hip.operand(C, N), M, N, K)
            outs.append(C)
            y, pre = torch.empty_like(C), torch.empty_like(C)
            hip.gemm_mxfp8(Aq, As, Bq, Bs, hip.operand(y, N), M, N, K, bias=bias, act=hip.ACT_GELU, Cpre=hip.operand(pre, N))
            outs += [y, pre]
            y2 = torch.empty_like(C)
            hip.gemm_mxfp8(Aq, As, Bq, Bs, hip.operand(y2, N), M, N, K, bias=bias, R=hip.operand(R, N), dropout_p=0.1, seed=11)
            outs.append(y2)
            y3 = torch.empty_like(C)
            hip.gemm_mxfp8(Aq, As, Bq, Bs, hip.operand(y3, N), M, N, K, P=hip.operand(P, N), act=hip.ACT_GELU, flags=hip.DACT)
            outs.append(y3)
            y4 = R.clone()
            hip.gemm_mxfp8(Aq, As, Bq, Bs, hip.operand(y4, N), M, N, K, beta=1.0)
            outs.append(y4)
            y5 = torch.empty_like(C)
            hip.gemm_mxfp8(Aq, As, Bq, Bs, hip.operand(y5, N), M, N, K, R=hip.operand(R, N))
            outs.append(y5)
            torch.cuda.synchronize()
        finally:
            hip.check(L.st5_gemm_set_mx8_tile(0), "st5_gemm_set_mx8_tile")
        return outs

    small, big = run(1), run(2)
    for i, (a, b) in enumerate(zip(small, big)):
        assert torch.isfinite(a.float()).all()
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), \
            f"output {i}: {int((a.view(torch.int16) != b.view(torch.int16)).sum())} of {a.numel()} elements differ"
    ref = _deq(Aq, As) @ _deq(Bq, Bs).t()
    assert float((big[0].float() - ref).abs().max()) <= 6e-3 * float(ref.abs().max())


@pytest.mark.parametrize("tile", [1, 2])
@pytest.mark.parametrize("M,N,K", [(512, 1024, 256), (1000, 4096, 1024), (300, 256, 128)])
def test_epilogue_written_fp8_image_is_the_quantiser_applied_to_the_output(cuda, tile, M, N, K):
    """Round 6 (st5_gemm_mxfp8_q): the fc1 forward (bias + GELU + pre-activation copy) and the data gradient through the GELU write the
    MX-fp8 image of their bf16 output from the epilogue; it must be byte for byte what st5_quant_mxfp8 makes of that output, and the bf16
    outputs must not change -- on both block tiles, with an M tail."""
    torch.manual_seed(M + N + tile)
    L = hip.lib()
    A = torch.randn(M, K, device=cuda).to(torch.bfloat16)
    B = (torch.randn(N, K, device=cuda) * 0.08).to(torch.bfloat16)
    Aq, As = hip.quant_mxfp8(A)
    Bq, Bs = hip.quant_mxfp8(B)
    bias = torch.randn(N, device=cuda)
    P = torch.randn(M, N, device=cuda).to(torch.bfloat16)
    hip.check(L.st5_gemm_set_mx8_tile(tile), "st5_gemm_set_mx8_tile")
    try:
        for kind in ("gelu", "dact"):
            kw = dict(bias=bias, act=hip.ACT_GELU) if kind == "gelu" else dict(P=hip.operand(P, N), act=hip.ACT_GELU, flags=hip.DACT)
            y0, pre0 = torch.empty(M, N, dtype=torch.bfloat16, device=cuda), torch.empty(M, N, dtype=torch.bfloat16, device=cuda)
            y1, pre1 = torch.empty_like(y0), torch.empty_like(pre0)
            if kind == "gelu":
                hip.gemm_mxfp8(Aq, As, Bq, Bs, hip.operand(y0, N), M, N, K, Cpre=hip.operand(pre0, N), **kw)
            else:
                hip.gemm_mxfp8(Aq, As, Bq, Bs, hip.operand(y0, N), M, N, K, **kw)
            q = torch.full((M, N), 0xAA, dtype=torch.uint8, device=cuda)
            sc = torch.full((M, N // 32), 0xAA, dtype=torch.uint8, device=cuda)
            if kind == "gelu":
                hip.gemm_mxfp8(Aq, As, Bq, Bs, hip.operand(y1, N), M, N, K, Cpre=hip.operand(pre1, N), out_q=(q, sc), **kw)
            else:
                hip.gemm_mxfp8(Aq, As, Bq, Bs, hip.operand(y1, N), M, N, K, out_q=(q, sc), **kw)
            wq, ws = hip.quant_mxfp8(y0)
            torch.cuda.synchronize()
            assert torch.equal(y0.view(torch.int16), y1.view(torch.int16)), kind
            if kind == "gelu":
                assert torch.equal(pre0.view(torch.int16), pre1.view(torch.int16))
            assert torch.equal(sc, ws), f"{kind}: {int((sc != ws).sum())} scale bytes differ"
            assert torch.equal(q, wq), f"{kind}: {int((q != wq).sum())} of {q.numel()} fp8 bytes differ"
        # any other epilogue combination is refused, not silently left unquantised
        y = torch.empty(M, N, dtype=torch.bfloat16, device=cuda)
        with pytest.raises(hip.HipKernelError):
            hip.gemm_mxfp8(Aq, As, Bq, Bs, hip.operand(y, N), M, N, K, bias=bias, out_q=(q, sc))
    finally:
        hip.check(L.st5_gemm_set_mx8_tile(0), "st5_gemm_set_mx8_tile")


@pytest.mark.parametrize("rows,cols", [(1000, 1024), (37, 768), (513, 256), (16, 2048)])
def test_layernorm_written_fp8_image_is_the_quantiser_applied_to_its_output(cuda, rows, cols):
    """Round 6 (st5_layernorm_fwd_q8): same output, mean and rstd as st5_layernorm_fwd, and the fp8 image == st5_quant_mxfp8(y)."""
    torch.manual_seed(rows)
    L = hip.lib()
    x = (torch.randn(rows, cols, device=cuda) * 3.0 + 0.5).to(torch.bfloat16)
    x[1] *= 1e-3
    g = torch.randn(cols, device=cuda) * 0.5 + 1.0
    b = torch.randn(cols, device=cuda) * 0.1
    outs = []
    for fused in (False, True):
        y = torch.empty_like(x)
        mean = torch.empty(rows, device=cuda); rstd = torch.empty(rows, device=cuda)
        if fused:
            q = torch.full((rows, cols), 0xAA, dtype=torch.uint8, device=cuda)
            sc = torch.full((rows, cols // 32), 0xAA, dtype=torch.uint8, device=cuda)
            hip.check(L.st5_layernorm_fwd_q8(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                             q.data_ptr(), sc.data_ptr(), rows, cols, 1e-5, hip.stream()), "st5_layernorm_fwd_q8")
        else:
            hip.check(L.st5_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                          rows, cols, 1e-5, hip.BF16, hip.stream()), "st5_layernorm_fwd")
        outs.append((y, mean, rstd))
    wq, ws = hip.quant_mxfp8(outs[0][0])
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16))
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert torch.equal(sc, ws) and torch.equal(q, wq), (int((sc != ws).sum()), int((q != wq).sum()))


def test_batched_weight_quantiser_matches_the_single_matrix_one(cuda):
    """Round 6 (st5_multi_quant_mxfp8): many contiguous matrices in one launch == st5_quant_mxfp8 on each."""
    import struct
    torch.manual_seed(5)
    L = hip.lib()
    mats = [torch.randn(r, c, device=cuda).mul_(sc).to(torch.bfloat16) for (r, c, sc) in
            ((1024, 1024, 0.03), (4096, 1024, 1.0), (96, 32, 50.0), (1024, 4096, 0.02), (7, 128, 1.0), (3072, 1024, 0.05))]
    outs, recs, blk0 = [], [], 0
    for m in mats:
        q = torch.full(m.shape, 0xAA, dtype=torch.uint8, device=cuda)
        sc = torch.full((m.shape[0], m.shape[1] // 32), 0xAA, dtype=torch.uint8, device=cuda)
        outs.append((q, sc))
        recs.append(struct.pack("<QQQqii", m.data_ptr(), q.data_ptr(), sc.data_ptr(), m.numel(), m.shape[1], blk0))
        blk0 += (m.numel() + 2047) // 2048
    jobs = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(cuda)
    hip.check(L.st5_multi_quant_mxfp8(jobs.data_ptr(), len(mats), blk0, hip.stream()), "st5_multi_quant_mxfp8")
    torch.cuda.synchronize()
    for m, (q, sc) in zip(mats, outs):
        wq, ws = hip.quant_mxfp8(m)
        torch.cuda.synchronize()
        assert torch.equal(sc, ws) and torch.equal(q, wq), m.shape


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


def _large_grads(cuda, fp8):
    """Loss and flat gradient of one pre-training update's two micro-batches on t5_transformer_large (speech 2 x 2 s + text 2 x 128)."""
    import bench
    from speecht5_amd import functional as Fn
    upd = None
    old_rows = Fn._FP8.min_rows
    try:
        Fn.set_fp8(fp8)
        Fn._FP8.min_rows, Fn._FP8.launches = 128, 0      # (the test's micro-batches are small: 198 / 256 rows)
        _, _, model, upd = bench.make_update(cuda, torch.bfloat16, "large", 2, 0, graph=False, micro="in_turn", layerdrop=0.0,
                                             prefetch_host=False, text_batch=2, text_len=128, seconds=2.0)
        upd.advance()
        losses = []
        with torch.cuda.stream(upd.stream):
            upd.ddp.zero_grad()
            for mb in upd.micro:
                loss = upd.task.forward_loss(mb, upd.model, upd.crit, upd.n)
                losses.append(loss.detach().float())
                loss.backward()
            upd.ddp.finish()
        torch.cuda.synchronize()
        names = {id(p): n for n, p in model.named_parameters()}
        grads = {names[id(p)]: upd.ddp.flat[o:o + p.numel()].clone() for p, o in zip(upd.ddp.params, upd.ddp.offsets)}
        return [float(l) for l in losses], grads, Fn._FP8.launches
    finally:
        Fn.set_fp8(False)
        Fn._FP8.min_rows = old_rows
        if upd is not None:
            upd.close()
        Fn.bf16_mirror.__init__()
        Fn.weight_cache.clear()
        Fn.set_layer_boundary_hook(None)
        Fn.set_compute_dtype(torch.float32)


def test_large_update_in_fp8_mode_stays_close_to_bf16_mode(cuda):
    """t5_transformer_large (24 + 6 layers, d = 1024, FFN 4096: models/speecht5.py:1402-1425) with the forward / data-gradient
    GEMMs of its Linears on the MX-fp8 kernel against the same update in bf16 (itself pinned to the reference by
    tests/test_large_gpu.py and the full-size tests): both micro-batch losses within 5e-2 relative, every parameter gradient of
    norm > 1e-6 (key biases excepted: their exact gradient is zero) at cosine >= 0.95 and the norm-weighted mean cosine >= 0.99 (the tolerance VERDICT r3 item 7 states)."""
    l16, g16, n16 = _large_grads(cuda, False)
    l8, g8, n8 = _large_grads(cuda, True)
    assert n16 == 0 and n8 >= 300, (n16, n8)          # the fp8 kernel really ran: ~9 GEMMs per layer x 30 layers x fwd + bwd
    for a, b in zip(l16, l8):
        assert abs(a - b) <= 5e-2 * abs(a), (l16, l8)
    num = den = 0.0
    worst = []
    for k, a in g16.items():
        b = g8[k]
        na, nb = float(a.norm()), float(b.norm())
        assert torch.isfinite(b).all(), k
        if na <= 1e-6 or k.endswith("k_proj.bias") or "norm_k" in k:
            continue        # (key biases: softmax is invariant to a shift of the keys, their exact gradient is 0 -- only rounding noise)
        c = float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm()).clamp_min(1e-30))
        worst.append((c, k))
        num += c * na; den += na
    worst.sort()
    print("losses bf16", l16, "fp8", l8, "worst cosines", worst[:5], "weighted mean", num / den)
    # (the mel post-net's BatchNorm backward cancels to ~1e-2 of its terms and amplifies any operand rounding -- bf16 itself reaches
    # 0.98-0.99 there against the fp32 oracle, tests/test_cfg2_shape_gpu.py; fp8 inputs upstream of it: 0.92 measured, bar 0.90)
    post = [w for w in worst if "speech_decoder_postnet.postnet" in w[1]]
    rest = [w for w in worst if "speech_decoder_postnet.postnet" not in w[1]]
    assert rest[0][0] >= 0.95, rest[:5]
    assert not post or post[0][0] >= 0.90, post[:5]
    assert num / den >= 0.99


def test_large_fp8_mode_against_the_oracle(cuda):
    """the fp8 compute mode was only ever compared with the bf16 mode.  Here the FULL t5_transformer_large
    architecture (24 + 6 layers, d = 1024, FFN 4096, pre-LN, layer-norm extractor: models/speecht5.py:1402-1425) runs one
    speech-pretraining forward + backward on 2 x 2 s with the forward / data-gradient GEMMs of its Linears on the MX-fp8 kernel, dropout
    off and every random draw injected, against the fp32 CPU ORACLE on the same weights and draws: loss within 5e-2, norm-weighted mean
    gradient cosine >= 0.99 (the bars VERDICT r3 item 7 / r4 item 6 state), every matrix gradient outside the mel post-net >= 0.95."""
    from argparse import Namespace
    from types import SimpleNamespace
    from oracle import speecht5_oracle as O
    from speecht5_amd import functional as Fn
    from speecht5_amd.criterions import SpeechPretrainCriterion
    from speecht5_amd.speecht5 import t5_transformer_large
    from speecht5_amd.synthetic import speech_pretrain_sample
    from speecht5_amd.task import SpeechT5Task
    from tests.util import injected_randomness, to_dev
    old_rows = Fn._FP8.min_rows
    try:
        Fn.set_compute_dtype(torch.bfloat16)
        Fn.set_fp8(True)
        Fn._FP8.min_rows, Fn._FP8.launches = 64, 0
        args = Namespace(label_rates=50, sample_rate=16000, speech_odim=80, bert_init=True, use_codebook=True,
                         share_input_output_embed=True, encoder_layerdrop=0.0, decoder_layerdrop=0.0)
        t5_transformer_large(args)
        for k, v in list(vars(args).items()):
            if "dropout" in k and isinstance(v, float):
                setattr(args, k, 0.0)
        args.encoder_layerdrop = args.decoder_layerdrop = 0.0
        task = SpeechT5Task.synthetic(args)
        torch.manual_seed(4243)
        model = task.build_model(args).to(cuda).train()
        sample = speech_pretrain_sample(B=2, seconds=2.0, device="cpu", seed=11)
        T = int(2.0 * 50) - 1
        mask = torch.zeros(2, T, dtype=torch.bool)
        mask[0, 10:70] = True
        mask[1, 25:85] = True
        mix_idx = torch.arange(0, T, 2)[: int(T * getattr(args, "codebook_prob", 0.5))]
        noise = torch.zeros(1)
        crit = SpeechPretrainCriterion(task, False, 1.0, 0.0, loss_weights=[10, 0.1], sync_logging=False)
        with injected_randomness(model, mask, mix_idx, noise, 2.0):
            loss, ss, _ = crit(model, to_dev(sample, cuda))
        (loss / ss).backward()
        torch.cuda.synchronize()
        assert Fn._FP8.launches >= 200, Fn._FP8.launches            # the fp8 kernel really ran (8 GEMMs per layer x 30 layers)
        got = {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None}
        sd = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
        cfg = SimpleNamespace(**vars(args))
        ref = O.forward_speech_pretrain(sd, cfg, sample, mask_indices=mask, mix_idx=mix_idx, gumbel_noise=noise)
        rl, rs, _ = O.speech_pretrain_loss(ref, sample, cfg, loss_weights=(10, 0.1))
        (rl / rs).backward()
        assert ss == rs
        lv, rv = float(loss.detach()), float(rl.detach())
        assert abs(lv - rv) <= 5e-2 * abs(rv), (lv, rv)
        rnorm = sum(float(v.grad.double().pow(2).sum()) for v in sd.values() if v.requires_grad and v.grad is not None) ** 0.5
        num = den = 0.0
        worst = []
        for n, g in got.items():
            if n not in sd or sd[n].grad is None:
                continue
            r = sd[n].grad.double()
            if float(r.norm()) <= 1e-5 * rnorm or n == "quantizer.vars":      # (structurally ~zero; the one discontinuous gradient)
                continue
            cos = float((g * r).sum() / (g.norm() * r.norm()).clamp_min(1e-30))
            w = float(r.norm())
            num += cos * w
            den += w
            if g.dim() >= 2:
                worst.append((cos, n))
        worst.sort()
        print("fp8 mode vs the fp32 oracle: norm-weighted mean gradient cosine", num / den, "worst matrices", worst[:6])
        assert num / den >= 0.99, num / den
        # measured on MI355X: 0.9908 weighted; every matrix outside the mel post-net >= 0.97; the post-net's convolutions 0.83-0.93 --
        # BatchNorm's backward removes the per-channel mean of a nearly constant incoming gradient, so what survives is a few % of what
        # the rounded operands carried (tests/test_fullsize_gpu.py: 0.98 already in bf16 mode; tests/util.py BF16_POST_COS)
        rest = [w for w in worst if "speech_decoder_postnet.postnet" not in w[1]]
        post = [w for w in worst if "speech_decoder_postnet.postnet" in w[1]]
        assert rest[0][0] >= 0.95, rest[:6]
        assert not post or post[0][0] >= 0.8, post[:6]
    finally:
        Fn.set_fp8(False)
        Fn._FP8.min_rows = old_rows
        Fn.bf16_mirror.__init__()
        Fn.weight_cache.clear()
        Fn.set_compute_dtype(torch.float32)


def test_producer_side_quantisation_changes_no_bit_of_the_update(cuda):
    """Round 6: fp8 images written by the producers (LayerNorm forward, GELU epilogue, data gradient through the GELU) and the batched
    weight images are the bytes the stand-alone quantiser makes -- so the update's losses and EVERY gradient element must be identical
    with the fusion on and off."""
    from speecht5_amd import functional as Fn
    old = Fn._FP8.fuse
    try:
        Fn._FP8.fuse = 0
        l0, g0, n0 = _large_grads(cuda, True)
        Fn._FP8.fuse = 7
        l1, g1, n1 = _large_grads(cuda, True)
    finally:
        Fn._FP8.fuse = old
    assert n0 == n1 and l0 == l1, (n0, n1, l0, l1)
    bad = [k for k in g0 if not torch.equal(g0[k], g1[k])]
    assert not bad, bad[:8]


def _fp8_updates(cuda, graph, micro, n_updates, fuse):
    """State after n updates of t5_transformer_large in fp8 mode (speech 4 x 4 s + text 4 x 256: 796 / 1024 rows, above the fp8 row floor)."""
    import bench
    from speecht5_amd import functional as Fn
    upd = None
    old = Fn._FP8.fuse
    try:
        Fn._FP8.fuse = fuse
        Fn.set_fp8(True)
        Fn._FP8.launches = 0
        _, _, model, upd = bench.make_update(cuda, torch.bfloat16, "large", 4, 0, graph=graph, micro=micro, layerdrop=0.05,
                                             text_batch=4, text_len=256, seconds=4.0)
        if graph:
            upd.prepare_graph()
            for _ in range(n_updates - 2):
                upd.update()
            upd.finish()
        else:
            Fn._S.force_static = True
            for _ in range(n_updates):
                upd.eager_update()
        return upd.state(), Fn._FP8.launches
    finally:
        Fn._S.force_static = False
        Fn._FP8.fuse = old
        Fn.set_fp8(False)
        if upd is not None:
            upd.close()
        Fn.bf16_mirror.__init__()
        Fn.weight_cache.clear()
        Fn.set_layer_boundary_hook(None)
        Fn.set_compute_dtype(torch.float32)


def test_fp8_updates_with_mirrored_weight_images_replayed_equal_the_unfused_eager_ones(cuda):
    """Round 6: 4 fp8-mode updates (2 recorded + 2 replayed, micro-batches side by side; weight images from the batched refresh behind
    every optimizer step, activation images from the producers) == the same 4 updates enqueued eagerly in turn with every image made
    by the stand-alone quantiser at its consumer -- parameters and both Adam moments bit for bit."""
    ref, n_ref = _fp8_updates(cuda, False, "in_turn", 4, fuse=0)
    got, n_got = _fp8_updates(cuda, True, "side_by_side", 4, fuse=7)
    assert n_ref > 400 and n_got > 200, (n_ref, n_got)
    assert ref[3] == got[3] == 4 and torch.isfinite(got[0]).all()
    for x, y, name in zip(ref[:3], got[:3], ("parameters", "first moment", "second moment")):
        assert torch.equal(x, y), f"{name} differ, max {float((x - y).abs().max()):.3e}"

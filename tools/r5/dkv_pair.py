"""Two-kernel reproducer for the side-by-side replay difference: the relative-position attention backward
(fa2::bwd_dq_kernel<true,true> + fa2::bwd_dkv_kernel<true,true> + the dQP.PE GEMM) of the speech micro-batch's shape on one stream,
compared bit for bit with its own solo output, while an aggressor loops on a second stream:
  none | nt (8192 x 3072 x 768 NT GEMM) | nt_s (3992 x 3072 x 768) | tn (3072 x 768 x 8192 weight gradient) | attn (the text shape's
  attention forward + backward, BH 192, T 512) | self (a second copy of the victim)
  dkv_pair.py <aggressor> [--reps 2000] [--graph 0|1] [--per 4]
Prints one line per arm: launches, wrong launches, and for the first few wrong ones which of dQ / dK / dV differ and where."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speecht5_amd import functional as Fn, hip

ap = argparse.ArgumentParser()
ap.add_argument("aggr"); ap.add_argument("--reps", type=int, default=2000); ap.add_argument("--graph", type=int, default=0)
ap.add_argument("--per", type=int, default=4, help="aggressor launches per victim launch")
ap.add_argument("--pool", type=int, default=16)
ap.add_argument("--bias", type=int, default=1, help="victim with (1) / without (0) the relative-position bias")
ap.add_argument("--pdrop", type=float, default=0.1, help="victim's attention dropout")
ap.add_argument("--abias", type=int, default=1); ap.add_argument("--apdrop", type=float, default=0.1)
a = ap.parse_args()
cuda = torch.device("cuda:0")
torch.manual_seed(7)
H, hd, d, nb, maxrel = 12, 64, 768, 320, 160


class Attn:
    def __init__(self, B, T, seed, bias=1, p_drop=0.1):
        self.p_drop = p_drop
        self.B, self.T = B, T
        # the seed the way a replayed step passes it: a tagged pointer to a device slot (csrc/common.h resolve_seed)
        self.slot = torch.tensor([seed], dtype=torch.int64, device=cuda)
        self.seed = (1 << 63) | self.slot.data_ptr()
        seed = self.seed
        self.qkv = (torch.randn(B * T, 3 * d, device=cuda) * 0.5).to(torch.bfloat16)
        self.pe = (torch.randn(nb, hd, device=cuda) * 0.1).to(torch.bfloat16) if bias else None
        self.dctx = (torch.randn(B * T, d, device=cuda) * 0.1).to(torch.bfloat16)
        self.q, self.k, self.v = (self.qkv, 3 * d, 0), (self.qkv, 3 * d, d), (self.qkv, 3 * d, 2 * d)
        self.ctx, self.lse, self.qp = Fn._flash_fwd(self.q, self.k, self.v, B, H, T, T, hd, self.pe, maxrel, None, False, p_drop, seed, save_qp=True)

    def fwd(self):
        Fn._flash_fwd(self.q, self.k, self.v, self.B, H, self.T, self.T, hd, self.pe, maxrel, None, False, self.p_drop, self.seed, save_qp=True)

    def bwd(self, out):
        Fn._flash_bwd(self.dctx, self.ctx, self.lse, self.q, self.k, self.v, (out, 3 * d, 0), (out, 3 * d, d), (out, 3 * d, 2 * d),
                      self.B, H, self.T, self.T, hd, self.pe, False, maxrel, None, False, self.p_drop, self.seed, qp=self.qp)


    def qpt(self):     # only the QP table kernel
        hip.check(hip.lib().st5_flash_attn_qp_table(Fn._eptr(self.q), self.q[1], self.pe.data_ptr(), self.qp.data_ptr(), self.B, H, self.T, nb,
                                                    hd ** -0.5, hip.BF16, hip.stream()), "qp_table")

    def bwdk(self, out):   # only the dq + dkv kernels (no dQP.PE GEMM behind them)
        B, T = self.B, self.T
        if not hasattr(self, "dvec"):
            self.dvec = torch.empty(B * H * T, dtype=torch.float32, device=cuda)
            self.dqp = torch.empty(B * H, T, nb, dtype=torch.bfloat16, device=cuda) if self.pe is not None else None
        es = 2
        hip.check(hip.lib().st5_flash_attn_bwd_2s(Fn._eptr(self.q), 3 * d, Fn._eptr(self.k), 3 * d, Fn._eptr(self.v), 3 * d, self.ctx.data_ptr(), d,
                                                  self.dctx.data_ptr(), d, out.data_ptr(), 3 * d, out.data_ptr() + d * es, 3 * d,
                                                  out.data_ptr() + 2 * d * es, 3 * d, self.lse.data_ptr(), self.dvec.data_ptr(), hip.ptr(self.pe),
                                                  hip.ptr(self.qp), hip.ptr(self.dqp), 0, B, H, T, T, hd, nb if self.pe is not None else 0, maxrel, 0,
                                                  (T + 7) // 8 * 8, hd ** -0.5, self.p_drop, self.seed, hip.BF16, hip.stream(), None), "bwd_2s")

    def dqpgemm(self, out):   # only the GEMM behind them: dQ += alpha * dQP . PE
        B, T = self.B, self.T
        hip.gemm(hip.operand(self.dqp, nb, zs0=H * T * nb, zs1=T * nb), hip.operand(self.pe, hd),
                 hip.operand(out, 3 * d, off=0, zs0=T * 3 * d, zs1=hd), T, hd, nb, hip.BF16, batch=B * H, zdiv=H,
                 flags=hip.B_KSTRIDED, alpha=hd ** -0.5, beta=1.0)


vic = Attn(8, 499, 1234567, a.bias, a.pdrop)
ref = torch.empty_like(vic.qkv)
vic.bwd(ref)
torch.cuda.synchronize()
ref2 = torch.empty_like(vic.qkv)
vic.bwd(ref2)
torch.cuda.synchronize()
assert torch.equal(ref, ref2), "the solo attention backward does not reproduce itself"
pool = [torch.empty_like(vic.qkv) for _ in range(a.pool)]

# ---- aggressors ----------------------------------------------------------------------------------------------------------------
if a.aggr in ("nt", "nt_s"):
    M, N, K = (8192 if a.aggr == "nt" else 3992), 3072, 768
    X = torch.randn(M, K, device=cuda).to(torch.bfloat16); W = torch.randn(N, K, device=cuda).to(torch.bfloat16)
    Y = torch.empty(M, N, device=cuda, dtype=torch.bfloat16)
    def aggr():
        hip.gemm(hip.operand(X, K), hip.operand(W, K), hip.operand(Y, N), M, N, K, hip.BF16)
elif a.aggr == "tn":
    M, N, K = 3072, 768, 8192
    X = torch.randn(K, M, device=cuda).to(torch.bfloat16); W = torch.randn(K, N, device=cuda).to(torch.bfloat16)
    Y = torch.empty(M, N, device=cuda, dtype=torch.float32)
    def aggr():
        hip.gemm(hip.operand(X, M), hip.operand(W, N), hip.operand(Y, N), M, N, K, hip.BF16, flags=hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32)
elif a.aggr in ("attn", "self", "fwd", "qpt", "bwdk", "dqpgemm"):
    other = Attn(8, 499, 1234567, a.abias, a.apdrop) if a.aggr == "self" else Attn(16, 512, 7654321, a.abias, a.apdrop)
    oout = torch.empty_like(other.qkv)
    other.bwdk(oout)
    if a.aggr in ("attn", "self"):
        def aggr():
            other.fwd(); other.bwd(oout)
    elif a.aggr == "fwd":
        aggr = other.fwd
    elif a.aggr == "qpt":
        aggr = other.qpt
    elif a.aggr == "bwdk":
        def aggr():
            other.bwdk(oout)
    else:
        def aggr():
            other.dqpgemm(oout)
elif a.aggr == "none":
    def aggr():
        pass
else:
    raise SystemExit("unknown aggressor")

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
wrong, launches, shown = 0, 0, 0


def check():
    global wrong, shown
    torch.cuda.synchronize()
    for o in pool:
        if not torch.equal(o, ref):
            wrong += 1
            if shown < 6:
                shown += 1
                ne = (o != ref).view(vic.B, vic.T, 3, H, hd)
                msg = []
                for s, nm in enumerate(("dQ", "dK", "dV")):
                    x = ne[:, :, s]
                    if x.any():
                        idx = x.nonzero()
                        msg.append(f"{nm}: {int(x.sum())} elements, b {sorted(set(idx[:, 0].tolist()))} h {sorted(set(idx[:, 2].tolist()))} "
                                   f"t {int(idx[:, 1].min())}..{int(idx[:, 1].max())} d {int(idx[:, 3].min())}..{int(idx[:, 3].max())} "
                                   f"max|diff| {float((o.float() - ref.float()).view(vic.B, vic.T, 3, H, hd)[:, :, s].abs().max()):.3e}")
                print("   wrong launch:", " | ".join(msg), flush=True)
                if shown == 1:
                    nk = ne[:, :, 1].any(-1)          # [B, T, H]: dK rows that differ
                    idx = nk.nonzero()
                    groups = sorted(set((int(b_), int(h_), int(t_) // 32) for b_, t_, h_ in idx.tolist()))
                    rf = ref.float().view(vic.B, vic.T, 3, H, hd)
                    print(f"      dK (b, h, 32-key group): {groups[:24]}; max|dK| {float(rf[:, :, 1].abs().max()):.3e} max|dV| {float(rf[:, :, 2].abs().max()):.3e}", flush=True)
                    b0, h0, g0 = groups[0]
                    dd = (o.float() - ref.float()).view(vic.B, vic.T, 3, H, hd)[b0, 32 * g0:32 * g0 + 32, 1, h0]
                    print(f"      first group: rows with a difference {int((dd != 0).any(-1).sum())} of {dd.shape[0]}, per-row max {[round(float(x), 6) for x in dd.abs().amax(-1)[:8].tolist()]}", flush=True)


def batch_eager():
    for o in pool:
        with torch.cuda.stream(sa):
            vic.bwd(o)
        with torch.cuda.stream(sb):
            for _ in range(a.per):
                aggr()


batch_eager(); torch.cuda.synchronize()   # warm-up on both streams (workspaces, kernel attributes) before any capture
if a.graph:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=sa, capture_error_mode="thread_local"):
        sb.wait_stream(sa)
        batch_eager()
        sa.wait_stream(sb)
    def one():
        with torch.cuda.stream(sa):
            g.replay()
else:
    one = batch_eager

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
while launches < a.reps:
    for o in pool:
        o.zero_()
    torch.cuda.synchronize()
    one()
    launches += len(pool)
    check()
print(f"PAIR aggressor {a.aggr:7s} (bias {a.abias} drop {a.apdrop}) victim bias {a.bias} drop {a.pdrop} graph {a.graph} per {a.per}: {wrong} wrong of {launches} victim launches "
      f"(lib {os.path.basename(os.environ.get('ST5_HIP_LIB', 'default'))})", flush=True)

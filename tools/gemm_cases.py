"""The GEMM launches of one training step, timed one class at a time outside the step (no second stream beside them):
NT form by epilogue class (plain / bias / bias+GELU+pre-activation copy / bias+dropout+residual / act'), TN form (weight
gradient: fp32 output, beta = 1, split-K; immediate and deferred batched reduction) -- with hipBLASLt (torch.matmul) beside
every shape.  Answers "which class is far from the library rate" (tools/gemm_shapes.py answers "which shape costs most")."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speecht5_amd import hip
from tools.bench_kernels import timeit

dev = torch.device("cuda:0")
bf = torch.bfloat16
L = hip.lib()
if os.environ.get("ST5_TN_PHASED"):
    L.st5_gemm_set_tn_phased(int(os.environ["ST5_TN_PHASED"]))
if os.environ.get("ST5_NT_SLOTS"):
    L.st5_gemm_set_nt_slots(int(os.environ["ST5_NT_SLOTS"]))
if os.environ.get("ST5_NT_TILE"):
    L.st5_gemm_set_nt_tile(int(os.environ["ST5_NT_TILE"]))
if os.environ.get("ST5_M64_MAX_TILES"):
    L.st5_gemm_set_m64_max_tiles(int(os.environ["ST5_M64_MAX_TILES"]))


def nt_cases():
    for (M, N, K) in ((8192, 3072, 768), (8192, 768, 3072), (8192, 768, 768), (8192, 2304, 768), (3992, 3072, 768), (3992, 768, 3072),
                      (3992, 768, 768), (3992, 2304, 768), (2504, 768, 768), (2504, 3072, 768)):
        A = torch.randn(M, K, device=dev).to(bf); B = torch.randn(N, K, device=dev).to(bf)
        C = torch.empty(M, N, device=dev, dtype=bf); C2 = torch.empty_like(C); R = torch.randn(M, N, device=dev).to(bf)
        bias = torch.randn(N, device=dev)
        oA, oB, oC, oC2, oR = hip.operand(A, K), hip.operand(B, K), hip.operand(C, N), hip.operand(C2, N), hip.operand(R, N)
        g = lambda **kw: hip.gemm(oA, oB, oC, M, N, K, hip.BF16, **kw)
        cases = (("plain", lambda: g()), ("bias", lambda: g(bias=bias)),
                 ("bias+gelu+pre", lambda: g(bias=bias, act=hip.ACT_GELU, Cpre=oC2)),
                 ("bias+drop+res", lambda: g(bias=bias, R=oR, dropout_p=0.1, seed=7)),
                 ("act' (dgrad)", lambda: g(P=oR, act=hip.ACT_GELU, flags=hip.DACT)))
        line = f"NT M={M:5d} N={N:5d} K={K:5d}"
        for name, f in cases:
            t = timeit(f)
            line += f" | {name} {t*1e6:6.1f}us {2*M*N*K/t/1e12:4.0f}TF"
        t = timeit(lambda: torch.matmul(A, B.t()))
        print(line + f" | hipblaslt {t*1e6:6.1f}us {2*M*N*K/t/1e12:4.0f}TF", flush=True)
        if (M, N, K) in ((8192, 768, 3072), (8192, 3072, 768)):     # the two rows that have to beat the library
            t_plain = timeit(cases[0][1])
            verdict = "MET" if t_plain <= t else "NOT MET"
            print(f"   asserted row {M}x{N}x{K}: plain {2*M*N*K/t_plain/1e12:4.0f} TF against hipBLASLt {2*M*N*K/t/1e12:4.0f} TF -- {verdict}", flush=True)


def tn_cases():
    for (M, N, K) in ((768, 768, 8192), (768, 3072, 8192), (3072, 768, 8192), (2304, 768, 8192), (768, 768, 3992), (768, 3072, 3992),
                      (3072, 768, 3992), (2304, 768, 3992), (768, 768, 2504), (512, 1536, 127992)):
        A = torch.randn(K, M, device=dev).to(bf); B = torch.randn(K, N, device=dev).to(bf)
        Cs = [torch.zeros(M, N, device=dev) for _ in range(6)]
        asum = torch.zeros(M, device=dev)
        oA, oB = hip.operand(A, M), hip.operand(B, N)
        oCs = [hip.operand(c, N) for c in Cs]
        fl = hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32
        t1 = timeit(lambda: hip.gemm(oA, oB, oCs[0], M, N, K, hip.BF16, flags=fl, beta=1.0))
        t1b = timeit(lambda: hip.gemm(oA, oB, oCs[0], M, N, K, hip.BF16, flags=fl, beta=1.0, asum=asum))

        def six():
            for oc in oCs:
                hip.gemm(oA, oB, oc, M, N, K, hip.BF16, flags=fl | hip.DEFERRABLE, beta=1.0, asum=asum)
            hip.check(L.st5_gemm_flush_splitk(hip.stream()), "flush")
        hip.check(L.st5_gemm_defer_splitk(1, hip.stream()), "defer")
        t6 = timeit(six) / 6
        hip.check(L.st5_gemm_defer_splitk(0, hip.stream()), "defer")
        tb = timeit(lambda: torch.matmul(A.t(), B))
        f = 2 * M * N * K
        print(f"TN M={M:5d} N={N:5d} K={K:6d} | split {t1*1e6:6.1f}us {f/t1/1e12:4.0f}TF | +bias col {t1b*1e6:6.1f}us {f/t1b/1e12:4.0f}TF"
              f" | deferred x6 {t6*1e6:6.1f}us {f/t6/1e12:4.0f}TF | hipblaslt(bf16 out) {tb*1e6:6.1f}us {f/tb/1e12:4.0f}TF", flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "nt"):
        nt_cases()
    if which in ("all", "tn"):
        tn_cases()

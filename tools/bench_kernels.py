"""Micro-benchmarks of the individual HIP kernels at SpeechT5-Base cfg-2 shapes (B=8 x 10 s).
Prints one line per kernel: time, achieved TFLOP/s or GB/s.  Run on the GPU box via gpurun."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speecht5_amd import hip

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def gemm_case(name, M, N, K, dtype, aks=False, bks=False, out_f32=False, act=0, batch=1):
    es = torch.tensor([], dtype=dtype).element_size()
    A = torch.randn((K, M) if aks else (M, K), device=dev).to(dtype)
    B = torch.randn((K, N) if bks else (N, K), device=dev).to(dtype)
    C = torch.empty(M, N, device=dev, dtype=torch.float32 if out_f32 else dtype)
    bias = torch.randn(N, device=dev)
    flags = (hip.A_KSTRIDED if aks else 0) | (hip.B_KSTRIDED if bks else 0) | (hip.OUT_F32 if out_f32 else 0)
    opA = hip.operand(A, M if aks else K)
    opB = hip.operand(B, N if bks else K)
    opC = hip.operand(C, N)
    t = timeit(lambda: hip.gemm(opA, opB, opC, M, N, K, hip.dt(dtype), bias=bias, act=act, flags=flags))
    print(f"{name:34s} M={M:6d} N={N:5d} K={K:6d} {str(dtype)[6:]:9s} {t*1e6:9.1f} us  {2*M*N*K/t/1e12:8.1f} TFLOP/s")


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "glds":
        for use in (0, 1):
            hip.lib().st5_gemm_set_glds(use)
            print("---- glds", use)
            for dtype in (torch.bfloat16, torch.float32):
                for (M, N, K) in ((3992, 2304, 768), (3992, 768, 768), (3992, 3072, 768), (3992, 768, 3072), (8192, 3072, 768), (8192, 768, 3072),
                                  (127992, 512, 1536), (8192, 8192, 8192)):
                    gemm_case("NT", M, N, K, dtype, act=1)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "nt256":
        # A/B of the two NT block tiles (st5_gemm_set_nt_tile) on the model's shapes + equality of the results
        dtype = torch.bfloat16
        L = hip.lib()
        shapes = ((3992, 2304, 768), (3992, 768, 768), (3992, 3072, 768), (3992, 768, 3072), (3992, 768, 2304), (8192, 2304, 768), (8192, 768, 768),
                  (8192, 3072, 768), (8192, 768, 3072), (8192, 768, 2304), (5008, 768, 768), (5008, 1536, 768), (127992, 512, 1536), (63992, 512, 1536),
                  (63992, 512, 1024), (8192, 8192, 8192), (301, 520, 768))
        for (M, N, K) in shapes:
            A = torch.randn(M, K, device=dev).to(dtype); B = torch.randn(N, K, device=dev).to(dtype)
            bias = torch.randn(N, device=dev); R = torch.randn(M, N, device=dev).to(dtype)
            outs = []
            line = f"M={M:6d} N={N:5d} K={K:5d}"
            for mode in ((2,) if os.environ.get("NT256_ONLY") else tuple(int(m) for m in os.environ.get("NT_MODES", "1,2,0").split(","))):
                L.st5_gemm_set_nt_tile(mode)
                C = torch.zeros(M, N, device=dev, dtype=dtype)
                if len(sys.argv) > 2 and sys.argv[2] == "plain":
                    f = lambda: hip.gemm(hip.operand(A, K), hip.operand(B, K), hip.operand(C, N), M, N, K, hip.dt(dtype), bias=bias)
                else:
                    f = lambda: hip.gemm(hip.operand(A, K), hip.operand(B, K), hip.operand(C, N), M, N, K, hip.dt(dtype), bias=bias, act=1,
                                         R=hip.operand(R, N), dropout_p=0.1, seed=5)
                t = timeit(f)
                outs.append(C.clone())
                line += f"  | tile{mode}: {t*1e6:7.1f} us {2*M*N*K/t/1e12:6.0f} TF"
            L.st5_gemm_set_nt_tile(0)
            t = timeit(lambda: torch.matmul(A, B.t()))
            line += f"  | hipblaslt {t*1e6:7.1f} us   equal={all(torch.equal(outs[0], o) for o in outs[1:])}"
            print(line)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "blas":
        # reference point only: the library GEMM (hipBLASLt through torch.matmul) on the same shapes
        dtype = torch.bfloat16
        for (M, N, K) in ((3992, 2304, 768), (3992, 768, 768), (3992, 3072, 768), (3992, 768, 3072), (8192, 2304, 768), (8192, 768, 768),
                          (8192, 3072, 768), (8192, 768, 3072), (5008, 768, 768), (5008, 1536, 768), (127992, 512, 1536), (63992, 512, 1536)):
            gemm_case("NT  mine", M, N, K, dtype)
            A = torch.randn(M, K, device=dev).to(dtype); B = torch.randn(N, K, device=dev).to(dtype)
            t = timeit(lambda: torch.matmul(A, B.t()))
            print(f"{'NT  hipblaslt':34s} M={M:6d} N={N:5d} K={K:6d} {'bfloat16':9s} {t*1e6:9.1f} us  {2*M*N*K/t/1e12:8.1f} TFLOP/s")
        for (M, N, K) in ((768, 768, 3992), (2304, 768, 3992), (3072, 768, 3992), (768, 3072, 3992), (768, 768, 8192), (2304, 768, 8192),
                          (3072, 768, 8192), (768, 3072, 8192), (512, 1536, 127992)):
            gemm_case("TN  mine (f32 out)", M, N, K, dtype, aks=True, bks=True, out_f32=True)
            A = torch.randn(K, M, device=dev).to(dtype); B = torch.randn(K, N, device=dev).to(dtype)
            t = timeit(lambda: torch.matmul(A.t(), B))
            print(f"{'TN  hipblaslt (bf16 out)':34s} M={M:6d} N={N:5d} K={K:6d} {'bfloat16':9s} {t*1e6:9.1f} us  {2*M*N*K/t/1e12:8.1f} TFLOP/s")
        sys.exit(0)
    M = 8 * 499
    for dtype in (torch.bfloat16, torch.float32):
        gemm_case("qkv proj (NT)", M, 2304, 768, dtype)
        gemm_case("out proj (NT)", M, 768, 768, dtype)
        gemm_case("fc1+gelu (NT)", M, 3072, 768, dtype, act=1)
        gemm_case("fc2 (NT)", M, 768, 3072, dtype)
        gemm_case("fc1 dgrad (NN)", M, 768, 3072, dtype, bks=True)
        gemm_case("fc1 wgrad (TN) f32 out", 3072, 768, M, dtype, aks=True, bks=True, out_f32=True)
        gemm_case("fc2 wgrad (TN) f32 out", 768, 3072, M, dtype, aks=True, bks=True, out_f32=True)
        gemm_case("conv1 (k3 s2) as gemm", 8 * 15999, 512, 1536, dtype, act=1)
        gemm_case("conv1 wgrad", 512, 1536, 8 * 15999, dtype, aks=True, bks=True, out_f32=True)
        gemm_case("big square", 8192, 8192, 8192, dtype)
    # LayerNorm / softmax / conv0 bandwidth
    L = hip.lib()
    for dtype in (torch.bfloat16,):
        rows, cols = M, 768
        x = torch.randn(rows, cols, device=dev).to(dtype); y = torch.empty_like(x)
        g = torch.ones(cols, device=dev); b = torch.zeros(cols, device=dev)
        mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
        t = timeit(lambda: L.st5_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, cols, 1e-5, hip.dt(dtype), hip.stream()))
        print(f"layernorm fwd {rows}x{cols}: {t*1e6:.1f} us  {2*rows*cols*2/t/1e9:.0f} GB/s")
        for rows2 in (3992, 8192):
            x2 = torch.randn(rows2, cols, device=dev).to(dtype); dy2 = torch.randn_like(x2); dx2 = torch.empty_like(x2)
            mean2 = torch.zeros(rows2, device=dev); rstd2 = torch.ones(rows2, device=dev)
            dg = torch.zeros(cols, device=dev); db = torch.zeros(cols, device=dev)
            ws = hip.workspace(L.st5_layernorm_bwd_ws_bytes(rows2, cols), dev)
            t = timeit(lambda: L.st5_layernorm_bwd(dy2.data_ptr(), x2.data_ptr(), g.data_ptr(), mean2.data_ptr(), rstd2.data_ptr(), dx2.data_ptr(),
                                                   dg.data_ptr(), db.data_ptr(), ws.data_ptr(), rows2, cols, None, 0.0, 0, hip.dt(dtype), hip.stream()))
            print(f"layernorm bwd (dx + dgamma/dbeta) {rows2}x{cols}: {t*1e6:.1f} us  {3*rows2*cols*2/t/1e9:.0f} GB/s")
        BH, T = 96, 499; lds = 504
        sc = torch.randn(BH, T, lds, device=dev).to(dtype); qp = torch.randn(BH, T, 320, device=dev).to(dtype)
        P = torch.empty_like(sc)
        t = timeit(lambda: L.st5_softmax_fwd(sc.data_ptr(), qp.data_ptr(), 0, P.data_ptr(), 0, BH, 12, T, T, lds, 320, 160, 0, 0.0, 0, hip.dt(dtype), hip.stream()))
        print(f"softmax+relpos fwd {BH}x{T}x{T}: {t*1e6:.1f} us  {2*BH*T*lds*2/t/1e9:.0f} GB/s")
        B, S, C = 8, 160000, 512
        wav = torch.randn(B, S, device=dev); w = torch.randn(C, 10, device=dev) * 0.4
        gm = torch.ones(C, device=dev); bt = torch.zeros(C, device=dev)
        Lo = (S - 10) // 5 + 1
        out = torch.empty(B, Lo, C, device=dev, dtype=dtype); stats = torch.empty(B, C, 2, device=dev)
        ws = hip.workspace(L.st5_conv0_ws_bytes(B, S, C, 10, 5), dev)
        t = timeit(lambda: L.st5_conv0_gn_gelu_fwd(wav.data_ptr(), w.data_ptr(), gm.data_ptr(), bt.data_ptr(), out.data_ptr(), stats.data_ptr(), ws.data_ptr(), B, S, C, 10, 5, 1e-5, hip.dt(dtype), hip.stream()))
        print(f"conv0+GN+GELU fwd B={B}: {t*1e6:.1f} us  {(B*S*4+B*Lo*C*2)/t/1e9:.0f} GB/s (algorithmic)")
        dY = torch.randn(B, Lo, C, device=dev).to(dtype)
        dw = torch.zeros(C, 10, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
        t = timeit(lambda: L.st5_conv0_gn_gelu_bwd(wav.data_ptr(), w.data_ptr(), gm.data_ptr(), bt.data_ptr(), stats.data_ptr(), dY.data_ptr(), dw.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), B, S, C, 10, 5, 0.1, hip.dt(dtype), hip.stream()))
        print(f"conv0+GN+GELU bwd B={B}: {t*1e6:.1f} us  {(B*S*4+B*Lo*C*2)/t/1e9:.0f} GB/s (algorithmic, 1 dY read)")

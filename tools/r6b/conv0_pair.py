"""conv layer 0 forward + backward on one stream, launch after launch compared bit for bit with the first result, while an aggressor
(GEMMs / LayerNorm / attention-like torch work) loops on a second stream.  Any difference = a race or a read of unwritten memory."""
import math, sys, torch
sys.path.insert(0, "/root/repo")
from speecht5_amd import hip
cuda = torch.device("cuda:0"); torch.cuda.set_device(0)
Ld = hip.lib()
def victim(B, S, C, k, fold, tab, iters, aggr):
    stride = 5; Lo = (S - k) // stride + 1
    torch.manual_seed(S + C)
    wav = torch.randn(B, S, device=cuda) * 0.7 + 0.05
    w = (torch.randn(C, k) * math.sqrt(2.0 / k)).to(cuda)
    g, b = (torch.rand(C) + 0.5).to(cuda), (torch.randn(C) * 0.1).to(cuda)
    dY = (torch.randn(B, Lo, C, device=cuda) * 0.3).to(torch.bfloat16)
    Ld.st5_conv0_set_fold(fold); Ld.st5_conv0_set_gelu_table(tab)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    A = torch.randn(4096, 4096, device=cuda, dtype=torch.bfloat16); Bm = torch.randn(4096, 4096, device=cuda, dtype=torch.bfloat16)
    X = torch.randn(8192, 768, device=cuda)
    ref = None; bad = {"out": 0, "stats": 0, "mom": 0, "dW": 0, "dG": 0, "dB": 0}
    torch.cuda.synchronize()
    for it in range(iters):
        if aggr:
            with torch.cuda.stream(s2):
                for _ in range(3):
                    Cm = A @ Bm; Y = torch.nn.functional.layer_norm(X, (768,)); Z = torch.softmax(X, -1)
        with torch.cuda.stream(s1):
            ws = hip.workspace(Ld.st5_conv0_ws_bytes(B, S, C, k, stride), cuda)
            out = torch.empty(B, Lo, C, dtype=torch.bfloat16, device=cuda); stats = torch.empty(B, C, 2, device=cuda)
            mom = torch.empty(B, Ld.st5_conv0_mom_count(k), dtype=torch.float64, device=cuda)
            hip.check(Ld.st5_conv0_gn_gelu_fwd_m(wav.data_ptr(), w.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(), stats.data_ptr(), mom.data_ptr(),
                                                 ws.data_ptr(), B, S, C, k, stride, 1e-5, hip.BF16, hip.stream()), "fwd")
            dW = torch.zeros(C, k, device=cuda); dG = torch.zeros(C, device=cuda); dB = torch.zeros(C, device=cuda)
            hip.check(Ld.st5_conv0_gn_gelu_bwd_m(wav.data_ptr(), w.data_ptr(), g.data_ptr(), b.data_ptr(), stats.data_ptr(), mom.data_ptr(), dY.data_ptr(),
                                                 dW.data_ptr(), dG.data_ptr(), dB.data_ptr(), ws.data_ptr(), B, S, C, k, stride, 1.0, hip.BF16, hip.stream()), "bwd")
            cur = {"out": out.view(torch.int16), "stats": stats, "mom": mom, "dW": dW, "dG": dG, "dB": dB}
            if ref is None:
                ref = {kk: v.clone() for kk, v in cur.items()}
            else:
                for kk in cur:
                    if not torch.equal(cur[kk], ref[kk]):
                        bad[kk] += 1
    torch.cuda.synchronize()
    Ld.st5_conv0_set_fold(1); Ld.st5_conv0_set_gelu_table(1)
    return bad
for (B, S, C, k) in ((2, 6000, 64, 10), (8, 160000, 512, 10)):
    for fold, tab in ((1, 1), (0, 0)):
        for aggr in (0, 1):
            print(f"B {B} S {S} C {C} fold {fold} table {tab} aggressor {aggr}:", victim(B, S, C, k, fold, tab, 600 if C == 64 else 150, aggr), flush=True)

"""Does a non_blocking host-to-device copy of PAGEABLE memory return before the source has been read?  (functional.HostStaging.get
in eager mode does `producer().to(device, non_blocking=True)` on a temporary.)  A stream is kept busy, the copy is enqueued behind
the busy work, the host then overwrites the source at once."""
import torch
cuda = torch.device("cuda:0")
s = torch.cuda.Stream()
A = torch.randn(8192, 8192, device=cuda, dtype=torch.bfloat16)
for n in (64, 4096, 1 << 16, 1 << 20, 1 << 24):
    late = 0
    for rep in range(5):
        h = torch.zeros(n, dtype=torch.float32)
        with torch.cuda.stream(s):
            for _ in range(40):
                B = A @ A                      # ~20 ms of work in front of the copy
            d = h.to(cuda, non_blocking=True)
        h.fill_(1.0)                            # the host moves on
        torch.cuda.synchronize()
        late += int(bool((d != 0).any()))
    print(f"pageable source of {n * 4} bytes: device saw the overwritten values in {late} of 5 copies", flush=True)

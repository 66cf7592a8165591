"""Where the several-rank update cuts its backward and how many gradient bytes each phase hands to the process group (DESIGN.md
section 5's table), computed on the real SpeechT5-Base parameter layout on the CPU: the three ranges tile the flat gradient buffer, the
first is complete behind the decoder / heads, the last (exposed) one is a quarter (round 6: the second cut moved from the middle of
the encoder stack to layer L/3, VERDICT r5 item 5: exposed bytes <= 1/3)."""
import torch


def test_phase_byte_ranges_of_the_base_model():
    import bench
    from speecht5_amd.ddp import FlatGradDataParallel
    from speecht5_amd.update import PretrainUpdate
    from speecht5_amd import functional as Fn
    try:
        args, task, model = bench.build(torch.device("cpu"), torch.float32, "base", layerdrop=0.05)[:3]
        ddp = FlatGradDataParallel(model)
        try:
            class U:      # cut_buckets() only reads .ddp and .model
                pass
            u = U()
            u.ddp, u.model = ddp, model
            cuts = PretrainUpdate.cut_buckets(u)
            b = ddp.buckets
            assert len(cuts) == 2 and 0 < cuts[0] < cuts[1] < len(b) - 1
            enc = model.encoder
            assert cuts[0] == ddp.module_bucket[(id(enc), "out")]
            assert cuts[1] == ddp.module_bucket[(id(enc.layers[len(enc.layers) // 3]), None)]
            mb, lo = [], 0
            for c in cuts + [len(b) - 1]:
                mb.append((b[c][1] - b[lo][0]) * 4 / 1e6)
                lo = c + 1
            total = ddp.flat.numel() * 4 / 1e6
            assert abs(sum(mb) - total) < 1.0 and abs(total - 617.6) < 1.0, (mb, total)
            assert abs(mb[0] - 234.3) < 1.0 and abs(mb[1] - 226.8) < 1.0 and abs(mb[2] - 156.4) < 1.0, mb
            assert mb[2] <= total / 3
            # the eight encoder layers of the middle phase: equal buckets
            mid = [b[i][1] - b[i][0] for i in range(cuts[0] + 1, cuts[1] + 1)]
            assert len(mid) == 8 and len(set(mid)) == 1
        finally:
            ddp.close()
    finally:
        Fn.set_layer_boundary_hook(None)
        Fn.set_compute_dtype(torch.float32)

"""Diagnostic (GPU): where do replayed / side-by-side updates stop being bit-identical to the eager in-turn form?
Tiny model (tests/test_graph_gpu._run) over mode x LayerDrop, then the full-size update of bench.py."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_graph_gpu import _run

cuda = torch.device("cuda:0")
bf = torch.bfloat16
def d(a, b):
    return float((a[0] - b[0]).abs().max())
for ld in (0.0, 0.3):
    ref = _run(cuda, bf, "static", 6, layerdrop=ld)
    print(f"tiny ld={ld}: graph vs static            {d(_run(cuda, bf, 'graph', 6, layerdrop=ld), ref):.3e}", flush=True)
    ref2 = _run(cuda, bf, "static", 6, layerdrop=ld)
    print(f"tiny ld={ld}: static vs static           {d(ref2, ref):.3e}", flush=True)
    r = _run(cuda, bf, "static_overlap_turn", 6, layerdrop=ld)
    print(f"tiny ld={ld}: overlap_turn twice         {d(_run(cuda, bf, 'static_overlap_turn', 6, layerdrop=ld), r):.3e}", flush=True)
    print(f"tiny ld={ld}: static_overlap vs turn     {d(_run(cuda, bf, 'static_overlap', 6, layerdrop=ld), r):.3e}", flush=True)
    print(f"tiny ld={ld}: graph_overlap vs turn      {d(_run(cuda, bf, 'graph_overlap', 6, layerdrop=ld), r):.3e}", flush=True)
if len(sys.argv) > 1:
    from tests.test_bench_update_gpu import _run as brun
    for ld in (0.0, 0.05):
        r = brun(cuda, False, "in_turn_2buf", 4, layerdrop=ld)
        print(f"full ld={ld}: eager in_turn_2buf twice   {d(brun(cuda, False, 'in_turn_2buf', 4, layerdrop=ld), r):.3e}", flush=True)
        print(f"full ld={ld}: eager side_by_side vs turn {d(brun(cuda, False, 'side_by_side', 4, layerdrop=ld), r):.3e}", flush=True)
        print(f"full ld={ld}: graph in_turn_2buf vs turn {d(brun(cuda, True, 'in_turn_2buf', 4, layerdrop=ld), r):.3e}", flush=True)
        print(f"full ld={ld}: graph side_by_side vs turn {d(brun(cuda, True, 'side_by_side', 4, layerdrop=ld), r):.3e}", flush=True)

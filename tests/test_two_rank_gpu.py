"""Several RANKS on real hardware: two processes sharing the one MI355X of the test box, gradient exchange
over gloo -- the `shared` branch of bench.py -- running the replayed several-rank update of speecht5_amd/update.py in both
exchange forms (phased: three graphs with the bucket ranges all-reduced between them; one_message: one graph + one all-reduce).
Both ranks hold rank 0's data and seeds, so the exchanged sum is exactly 2 x one rank's gradient and, with
grad_scale = 1 / (2 micro-batches x 2 ranks), every rank's parameters and Adam moments after 4 updates must equal the ONE-rank
replayed update's bit for bit (tests/two_rank_worker.py).  What it proves: the phased capture, the asynchronous hand-over of
bucket ranges between graph launches, wait + eager Adam tail work with a REAL second process on the same GPU; what it cannot
prove on one GPU: RCCL over xGMI (the driver's 8-GPU run)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(nproc, exchange, out, port, extra=()):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "8"
    env.setdefault("GPU_MAX_HW_QUEUES", "8")     # (two processes on ONE device: see bench.respawn)
    worker = os.path.join(ROOT, "tests", "two_rank_worker.py")
    if nproc == 1:
        cmd = [sys.executable, worker, "--exchange", exchange, "--out", out] + list(extra)
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), worker, "--exchange", exchange, "--out", out] + list(extra)
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stdout[-3000:]}\n{r.stderr[-6000:]}"
    return [json.load(open(f"{out}.rank{k}.json")) for k in range(nproc)]


def test_two_ranks_sharing_the_gpu_equal_the_one_rank_update(cuda, tmp_path):
    one = _launch(1, "phased", str(tmp_path / "one"), 0)[0]
    assert one["finite"] and one["t"] == 4 and not one["split"]
    for i, exchange in enumerate(("phased", "one_message")):
        ranks = _launch(2, exchange, str(tmp_path / exchange), 29611 + i)
        assert all(r["split"] for r in ranks)
        assert all(r["phased"] == (exchange == "phased") for r in ranks)
        assert ranks[0]["t"] == ranks[1]["t"] == 4
        assert ranks[0]["digest"] == ranks[1]["digest"], f"{exchange}: the two ranks disagree ({ranks[0]['pnorm']} vs {ranks[1]['pnorm']})"
        assert ranks[0]["digest"] == one["digest"], f"{exchange}: two ranks != one rank ({ranks[0]['pnorm']} vs {one['pnorm']})"


def test_two_ranks_side_by_side_micro_batches_equal_the_one_rank_update(cuda, tmp_path):
    """The DEFAULT form on several ranks (`bench.py --gpus N`): micro-batches side by side on two streams inside the replayed local
    phase (two gradient buffers, summed before the exchange), one all-reduce of the whole buffer behind it, Adam eagerly.  Same data
    on both ranks, so it must equal the one-rank in-turn update bit for bit."""
    one = _launch(1, "phased", str(tmp_path / "one"), 0)[0]
    ranks = _launch(2, "one_message", str(tmp_path / "sbs"), 29651, extra=["--micro", "side_by_side"])
    assert all(r["split"] and not r["phased"] for r in ranks)
    assert ranks[0]["digest"] == ranks[1]["digest"] == one["digest"], (ranks[0]["pnorm"], ranks[1]["pnorm"], one["pnorm"])


def test_two_ranks_side_by_side_phased_exchange_equals_the_one_rank_update(cuda, tmp_path):
    """Round 6, the default several-rank form: both micro-batches inside each of the three phase graphs, every completed bucket range
    summed over the two gradient buffers and all-reduced (gloo here) while the next phase runs.  Same data on both ranks: bit-equal to
    the one-rank update; each rank its OWN data and lr = 0: the buffer the optimizer receives == g_rank0 + g_rank1 of two one-rank
    eager runs bit for bit (a range summed or sent twice, too early or from the wrong buffer would not be)."""
    import torch
    one = _launch(1, "phased", str(tmp_path / "one"), 0)[0]
    ranks = _launch(2, "phased", str(tmp_path / "sbs_ph"), 29681, extra=["--micro", "side_by_side"])
    assert all(r["split"] and r["phased"] for r in ranks)
    assert ranks[0]["digest"] == ranks[1]["digest"] == one["digest"], (ranks[0]["pnorm"], ranks[1]["pnorm"], one["pnorm"])
    g = []
    for r in (0, 1):
        info = _launch(1, "phased", str(tmp_path / f"ref{r}"), 0, extra=["--no-graph", "--data-rank", str(r)])[0]
        g.append(torch.load(str(tmp_path / f"ref{r}") + ".rank0.grad.pt"))
    want = g[0] + g[1]
    out = str(tmp_path / "own_sbs_ph")
    ranks = _launch(2, "phased", out, 29682, extra=["--own-data", "--micro", "side_by_side"])
    assert all(r["split"] and r["phased"] and r["grad_calls"] == 4 for r in ranks)
    for k in (0, 1):
        got = torch.load(f"{out}.rank{k}.grad.pt")
        assert torch.equal(got, want), f"rank {k}: exchanged buffer != g0 + g1 ({int((got != want).sum())} of {got.numel()} sampled elements differ)"


def test_two_ranks_side_by_side_without_a_graph_equal_the_one_rank_update(cuda, tmp_path):
    """`bench.py --gpus N --no-graph` with the default micro-batch mode -- two gradient buffers and NO captured local
    phase -- used to reach accumulate_overlapped() outside local_phase() and trip its assert.  The eager several-rank side-by-side
    update now runs the same split as the replayed one (local phase, one exchange, Adam) and must produce the same bits."""
    one = _launch(1, "phased", str(tmp_path / "one"), 0)[0]
    ranks = _launch(2, "one_message", str(tmp_path / "sbs_eager"), 29671, extra=["--micro", "side_by_side", "--no-graph"])
    assert all(r["split"] and not r["phased"] for r in ranks)
    assert ranks[0]["digest"] == ranks[1]["digest"] == one["digest"], (ranks[0]["pnorm"], ranks[1]["pnorm"], one["pnorm"])


def test_two_ranks_with_their_own_data_exchange_exactly_the_sum_of_their_gradients(cuda, tmp_path):
    """the same-data test above cannot see a reduction that mixes the ranks' buffers.  Here rank r holds its own
    data and seeds, lr = 0 (parameters fixed), and the gradient buffer the optimizer step receives in the 4th update of the two-rank
    REPLAYED run (phased and one-message exchange) must equal g_rank0 + g_rank1 of two one-rank eager runs, bit for bit (a sum of
    two fp32 values does not depend on the order)."""
    import torch
    g = []
    for r in (0, 1):
        info = _launch(1, "phased", str(tmp_path / f"ref{r}"), 0, extra=["--no-graph", "--data-rank", str(r)])[0]
        assert info["grad_calls"] == 4 and info["data_rank"] == r
        g.append(torch.load(str(tmp_path / f"ref{r}") + ".rank0.grad.pt"))
    assert not torch.equal(g[0], g[1]) and torch.isfinite(g[0]).all() and torch.isfinite(g[1]).all()
    want = g[0] + g[1]
    for i, exchange in enumerate(("phased", "one_message")):
        out = str(tmp_path / f"own_{exchange}")
        ranks = _launch(2, exchange, out, 29631 + i, extra=["--own-data"])
        assert all(r["split"] and r["phased"] == (exchange == "phased") and r["grad_calls"] == 4 for r in ranks)
        for k in (0, 1):
            got = torch.load(f"{out}.rank{k}.grad.pt")
            assert torch.equal(got, want), (f"{exchange}, rank {k}: exchanged buffer != g0 + g1 "
                                            f"({int((got != want).sum())} of {got.numel()} sampled elements differ)")


def test_two_ranks_bf16_gradient_payload_stays_close_to_the_exact_exchange(cuda, tmp_path):
    """`exchange_payload="bf16"`: the one-message exchange carries the bf16 rounding of the fp32 gradient sum.  Both
    ranks hold the same data, so the collective's bf16 sum 2 x bf16(g) is exact and the only difference to the fp32 exchange is ONE
    rounding of every gradient element (relative 2^-9): both ranks must agree bit for bit with each other, and after 4 updates the
    sampled parameters must differ from the exact-exchange run's by at most 5 % of the distance the 4 updates moved them."""
    exact = _launch(2, "one_message", str(tmp_path / "exact"), 29661, extra=["--micro", "side_by_side"])
    half = _launch(2, "one_message", str(tmp_path / "half"), 29662, extra=["--micro", "side_by_side", "--payload", "bf16"])
    assert half[0]["digest"] == half[1]["digest"] and all(r["finite"] and r["t"] == 4 for r in half)
    assert half[0]["digest"] != exact[0]["digest"], "the bf16 payload changed nothing: was it used?"
    import torch
    e = torch.load(str(tmp_path / "exact") + ".rank0.params.pt")
    h = torch.load(str(tmp_path / "half") + ".rank0.params.pt")
    assert torch.equal(e["p0"], h["p0"])
    moved = float((e["p"].double() - e["p0"].double()).norm())
    rel = float((e["p"].double() - h["p"].double()).norm()) / moved
    assert 0 < rel <= 0.05, rel       # 4 Adam steps by gradients that differ in their 9th bit (and what that flips downstream)

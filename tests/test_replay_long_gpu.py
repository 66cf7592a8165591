"""the replayed update over a LONG run, in fresh processes.  Round 4's five-update equality tests could not see
events that happen once in a hundred replays; this one replays 200 seeded updates of the benched workload (full size) in three
processes -- in turn on one stream, and twice side by side on two streams (the default, with and without a host synchronisation behind
every update) -- and requires the three trajectories of (parameter, moment) checksums to agree at EVERY update.  Before round 5's fix
of fa2::bwd_dkv_kernel's barrier the side-by-side trajectory left the in-turn one within 2-30 updates in every
process (profiles/r5_replay_hunt.txt)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 200


def _traj(tmp_path, tag, mode, sync):
    out = str(tmp_path / f"{tag}.json")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "replay_worker.py"), "--mode", mode, "--n", str(N), "--sync", str(sync), "--out", out]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stdout[-2000:]}\n{r.stderr[-6000:]}"
    d = json.load(open(out))
    assert d["finite"] and len(d["cs"]) == N
    return d["cs"]


def _first_difference(a, b):
    return next((k + 1 for k in range(N) if a[k] != b[k]), None)


def test_two_hundred_replayed_updates_agree_across_processes_and_modes(cuda, tmp_path):
    turn = _traj(tmp_path, "in_turn", "in_turn", 0)
    side = _traj(tmp_path, "side", "side_by_side", 0)
    side_sync = _traj(tmp_path, "side_sync", "side_by_side", 1)
    assert turn[0] != turn[-1], "the parameters did not move"
    assert _first_difference(turn, side) is None, f"side by side leaves the in-turn trajectory at update {_first_difference(turn, side)}"
    assert _first_difference(side, side_sync) is None, \
        f"two side-by-side processes disagree from update {_first_difference(side, side_sync)} (per-update synchronisation in one of them)"

"""bench.py's launch contract, no GPU: `python bench.py --gpus N` from a bare shell re-executes itself under torch.distributed.run
with one rank per GPU, rendezvous on 127.0.0.1 and the caller's own flags; under the driver (WORLD_SIZE set) it must not."""
import os
import sys
from argparse import Namespace

import pytest


def test_bare_shell_several_ranks_respawns_under_torch_distributed_run(monkeypatch):
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "9", "--warmup", "3", "--exchange", "one_message"])
    monkeypatch.delenv("MASTER_PORT", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.respawn(Namespace(gpus=4))
    assert e.value.code == 7                                   # the ranks' exit status is the script's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 29500 <= int(cmd[cmd.index("--master-port") + 1]) < 29900
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "9", "--warmup", "3", "--exchange", "one_message"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"     # dmabuf IPC: RCCL across processes needs it on this driver
    assert int(seen["env"]["OMP_NUM_THREADS"]) >= 1


def test_flags_of_the_driver_contract_parse(monkeypatch):
    """--gpus / --steps / --warmup (driver), --config 3, --exchange, --micro, --layerdrop: parsed; --config 3 is routed away before
    any GPU call (here: to a stub)."""
    import bench
    got = {}
    monkeypatch.setattr(bench, "config3", lambda a: got.update(vars(a)))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--config", "3", "--steps", "4", "--warmup", "2"])
    bench.main()
    assert got["config"] == 3 and got["steps"] == 4 and got["warmup"] == 2 and got["gpus"] == 1
    assert got["exchange"] == "phased" and got["micro"] == "side_by_side" and got["layerdrop"] == 0.05

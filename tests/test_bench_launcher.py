"""bench.py --gpus N from a bare shell (missing 1): the script starts its own ranks as a fresh child under
torch.distributed.run, hands its arguments on unchanged and leaves with the child's exit code.  CPU only: on this box
the ranks stop at "needs an MI355X", which is exactly the relay being tested."""
import os
import subprocess
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_launcher_command_relays_every_argument():
    import bench
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2", "--backend", "gloo", "--no-full-run", "--config", "c4"]
    cmd = bench.launcher_command(argv, 4, 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == argv                       # the script's own arguments, in order, after the script
    assert 1024 < bench.free_port() < 65536


def test_bare_gpus_flag_starts_the_ranks_and_relays_their_exit_code():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the GPU rehearsal covers this (tools/gloo2_rehearsal.sh)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--no-full-run"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300)
    assert "starting 2 ranks" in p.stderr and "--nproc-per-node 2" in p.stderr
    assert "needs an MI355X" in p.stderr              # the ranks ran bench.py with WORLD_SIZE set and said why they stop
    assert p.returncode != 0                          # ... and their failure is the launcher's exit code

"""bench.py --gpus N: the parent starts N ranks as a child torch.distributed.run (the contract's command), a
WORLD_SIZE that disagrees with --gpus is refused.  Reference path being scaled: core/index_manager.py:371."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_launch_command_is_the_contracts():
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "5", "--warmup", "1"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "1"]


def test_world_size_must_equal_gpus():
    assert bench.world_mismatch(1, {}) is None
    assert bench.world_mismatch(8, {}) is None                      # no launcher: bench starts the ranks itself
    assert bench.world_mismatch(8, {"WORLD_SIZE": "8"}) is None
    assert bench.world_mismatch(1, {"WORLD_SIZE": "1"}) is None     # one-rank rehearsal
    assert "WORLD_SIZE=1" in bench.world_mismatch(8, {"WORLD_SIZE": "1"})
    assert "--gpus 2" in bench.world_mismatch(2, {"WORLD_SIZE": "4"})


def test_mismatch_exits_nonzero_before_any_gpu_call():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr and p.stdout.strip() == ""


def test_parent_relays_child_rc_without_a_gpu():
    # no GPU here: every rank exits with "needs a ROCm GPU"; what is checked is that the parent started ranks
    # (the launcher's log names both) and passed the failure on instead of printing a one-GPU line
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "torch.distributed.run" in p.stderr and "--nproc-per-node 2" in p.stderr
    assert p.stdout.strip() == ""
